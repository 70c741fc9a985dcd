// coral_search.cpp — host side of the amplicon-interval search (no device code).
//
// One step of the breadth-first search of find_interval_i (/root/reference/src/infer_breakpoint_graph.py:343-673) has a part
// that is a pure function of the interval's coordinates (ibg:362-457):
//   * the CN segments reached from the interval through chimeric reads, each with the SET of read names that reach it
//     (ibg:369-384) — sets of str whose iteration order later decides the order of the breakpoints;
//   * segments with fewer reads than min_cluster_cutoff dropped (ibg:385-391), the rest grouped into runs of neighbouring
//     segments whose sets are united with |= (ibg:392-419);
//   * for every run, alignment2bp of every read of the united set against (run, interval) (ibg:428-434, bu:70-96);
//   * cluster_bp_list + the bpc2bp loop over every run's candidates (ibg:436-457; coral_call_breakpoints).
// Here that part is ONE native job on index arrays: the sets are replayed with pyset_emu.h (no Python object is created), the
// candidates are FILTERED out of the pair table the GPU built once per graph build (k_bp_pairs in coral_kernels.hip: which
// pairs of a read's alignments fall into the two intervals is four comparisons per pair), and because the job is pure it is
// computed AHEAD on worker threads as soon as the caller knows an interval will be searched (coral_search_prefetch); the
// order-dependent rest of the search (addbp, interval refinement) then finds the result ready (coral_search_step).
// coral_search_within is the pair filter for alignment2bp_l over all chimeric reads (find_breakpoints, ibg:676-690).
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/coral_hip.h"
#include "pyset_emu.h"

namespace {
using coral_detail::PySetEmu;

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Per-read data packed for the two hot loops (reach sets, pair filter): one contiguous record per read instead of seven
// parallel arrays plus a 32-byte-per-slot table, so a read costs one or two cache lines.  Record = {n rows, first table row}
// then per row {contig, ra, rb, cni0, cni1, bits of pair (k - 1, k + 1), and the whole pair-table slot of pair (k, k + 1):
// c1, p1, c2, p2, gap, bits}.  Candidates of adjacent pairs (almost all of them) are emitted from the record itself; the
// rare skip-one candidates read their slot of the pair table.
struct PackedRow {
    int32_t tid, ra, rb, cni0, cni1, bits_skip;
    int32_t c1, p1, c2, p2, gap, bits_adj;
};
static_assert(sizeof(PackedRow) == 48, "PackedRow layout");

struct Calls {                                            // coral_call_breakpoints on one run's candidates
    int32_t n_clusters = 0, n_calls = 0;
    std::vector<int32_t> cluster_size, flags;
    std::vector<int64_t> head, p1, p2, sup_off, sup_idx;
    std::vector<double> stats;
};

struct StepResult {
    int rc = CORAL_OK;
    char err[200] = "";
    std::vector<int64_t> groups;                          // [n_groups][4]: contig id, first segment, last segment, candidates
    std::vector<int64_t> cand;                            // [K][13]: c1 p1 o1 c2 p2 o2 read(name id) i j gap swapped mqa mqb
    std::vector<int32_t> order;                           // reads (table index) of every run in set-iteration order
    std::vector<int64_t> order_off;                       // [n_groups + 1]
    std::vector<Calls> calls;                             // per run
    // the same, flattened for one-shot retrieval (coral_search_result): see the header for the layout of `meta`
    std::vector<int64_t> meta, sup;
    std::vector<double> stats;
    void clear() {
        rc = CORAL_OK; err[0] = 0;
        groups.clear(); cand.clear(); order.clear(); order_off.assign(1, 0); calls.clear();
        meta.assign(1, 0); sup.clear(); stats.clear();
    }
    void flatten() {
        const size_t ng = groups.size() / 4;
        meta.assign(1, (int64_t)ng);
        sup.clear(); stats.clear();
        for (size_t g = 0; g < ng; ++g) {
            static const Calls none;
            const Calls &c = g < calls.size() ? calls[g] : none;
            meta.insert(meta.end(), groups.begin() + 4 * g, groups.begin() + 4 * g + 4);
            meta.push_back(c.n_clusters);
            meta.push_back(c.n_calls);
            for (int32_t k = 0; k < c.n_clusters; ++k) meta.push_back(c.cluster_size[(size_t)k]);
            for (int32_t k = 0; k < c.n_calls; ++k) {
                const int64_t row[6] = {c.head[(size_t)k], c.p1[(size_t)k], c.p2[(size_t)k], c.flags[(size_t)k],
                                        (int64_t)sup.size(), (int64_t)sup.size() + (c.sup_off[(size_t)k + 1] - c.sup_off[(size_t)k])};
                meta.insert(meta.end(), row, row + 6);
                sup.insert(sup.end(), c.sup_idx.begin() + c.sup_off[(size_t)k], c.sup_idx.begin() + c.sup_off[(size_t)k + 1]);
                stats.insert(stats.end(), c.stats.begin() + 6 * (size_t)k, c.stats.begin() + 6 * (size_t)k + 6);
            }
        }
    }
};

struct Scratch {                                          // per thread
    std::vector<uint32_t> seen;                           // per read: stamp of the step that expanded it
    uint32_t stamp = 0;
    std::vector<char> used;
};

struct Entry {                                            // one (possibly pending) step in the cache
    std::mutex m;
    std::condition_variable cv;
    bool done = false, taken = false;                     // taken: some thread is computing it
    StepResult res;
    int64_t key[5];
};

struct KeyHash {
    size_t operator()(const std::array<int64_t, 5> &k) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int64_t v : k) { h ^= (uint64_t)v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ull; }
        return (size_t)(h ^ (h >> 31));
    }
};

struct Search {
    int64_t n_reads = 0, n_rows = 0, n_ent = 0;
    const int64_t *off = nullptr, *row_read = nullptr, *read_hash = nullptr, *read_name = nullptr, *e_key = nullptr, *e_row = nullptr;
    const int32_t *pairs = nullptr;                       // [2 * n_rows][8]  (k_bp_pairs)
    int32_t n_tid = 0;
    const int64_t *seg_off = nullptr, *seg_start = nullptr, *seg_end = nullptr;      // CN segments per contig, file order; end inclusive
    std::vector<int64_t> pack_off;                        // per read: index of its record in `pack` (in int32 units)
    std::vector<int32_t> pack;
    // parameters of the build (coral_search_params)
    double min_cluster_cutoff = 3.0, accept_floor = 3.0;
    int64_t max_seq_len = 2000000, bp_distance_cutoff = 2000, match_cutoff = 100;
    // synchronous calls (within / between / inline steps) use these
    Scratch main_scratch;
    StepResult main_result;
    const StepResult *current = nullptr;                  // what coral_search_result / coral_search_calls describe
    std::shared_ptr<Entry> current_entry;
    // look-ahead
    std::vector<std::thread> workers;
    std::mutex qm;
    std::condition_variable qcv;
    std::deque<std::shared_ptr<Entry>> queue;
    std::unordered_map<std::array<int64_t, 5>, std::shared_ptr<Entry>, KeyHash> cache;
    bool stop = false;
    char err[256] = "";
    // CORAL_SEARCH_PROFILE=1: seconds per phase and work counters over all steps, printed when the handle is freed
    bool profile = false;
    std::mutex pm;
    double t_reach = 0, t_union = 0, t_cand = 0, t_call = 0, t_wait = 0;
    long long n_steps = 0, n_visit = 0, n_adds = 0, n_keys = 0, n_union_items = 0, n_cand = 0, n_inline = 0;
};

inline const int32_t *rec_of(const Search &S, int64_t r) { return S.pack.data() + S.pack_off[(size_t)r]; }
inline const PackedRow *rows_of(const int32_t *rec) { return reinterpret_cast<const PackedRow *>(rec + 2); }

inline bool row_in(const PackedRow &w, int64_t t, int64_t s, int64_t e) {
    // interval_overlap(rint, [chr, s, e]) with rint = [chr, ra, rb]; for '-' rows ra > rb, i.e. "interval contains the whole
    // alignment" (bu:11-15)
    return w.tid == t && w.ra <= e && s <= w.rb;
}

inline bool emit_fields(const Search &S, std::vector<int64_t> &cand, const int32_t *p, int32_t bits, int64_t ia, int64_t ib, int64_t read) {
    const bool swapped = (bits & 16) != 0;
    const int64_t row[13] = {p[0], p[1], (bits >> 2) & 1, p[2], p[3], (bits >> 3) & 1, S.read_name[read], swapped ? ib : ia,
                             swapped ? ia : ib, p[4], swapped ? 1 : 0, (bits >> 8) & 0xff, (bits >> 16) & 0xff};
    cand.insert(cand.end(), row, row + 13);
    return (bits & 128) == 0;                  // false: a contig outside chr1..22,X,Y,M reaches interval2bp (KeyError, bu:293)
}

// adjacent pair (k, k + 1): everything is in the packed record
inline bool emit_adj(const Search &S, std::vector<int64_t> &cand, const PackedRow &w, int64_t k, int64_t read) {
    return emit_fields(S, cand, &w.c1, w.bits_adj, k, k + 1, read);
}

inline bool emit(const Search &S, std::vector<int64_t> &cand, int64_t slot, int64_t read, int64_t base) {
    const int32_t *p = S.pairs + 8 * slot;
    const int32_t bits = p[5];
    const int64_t ia = p[6] - base, ib = p[7] - base;
    const bool swapped = (bits & 16) != 0;
    const int64_t row[13] = {p[0], p[1], (bits >> 2) & 1, p[2], p[3], (bits >> 3) & 1, S.read_name[read], swapped ? ib : ia,
                             swapped ? ia : ib, p[4], swapped ? 1 : 0, (bits >> 8) & 0xff, (bits >> 16) & 0xff};
    cand.insert(cand.end(), row, row + 13);
    return (bits & 128) == 0;                  // false: a contig outside chr1..22,X,Y,M reaches interval2bp (KeyError, bu:293)
}

// alignment2bp (bu:70-96) of one read between intervals 1 and 2 (either order); appends to `cand`.
inline bool pairs_between(const Search &S, Scratch &T, std::vector<int64_t> &cand, int64_t r, int64_t t1, int64_t s1, int64_t e1,
                          int64_t t2, int64_t s2, int64_t e2) {
    const int32_t *rec = rec_of(S, r);
    const int64_t n = rec[0], base = rec[1];
    if (n < 2) return true;
    const PackedRow *w = rows_of(rec);
    bool ok = true;
    T.used.assign((size_t)n, 0);
    for (int64_t k = 0; k + 1 < n; ++k) {                     // pairs (k, k + 1)
        if (!(w[k].bits_adj & 2)) continue;
        if ((row_in(w[k], t1, s1, e1) && row_in(w[k + 1], t2, s2, e2)) || (row_in(w[k + 1], t1, s1, e1) && row_in(w[k], t2, s2, e2))) {
            T.used[(size_t)k] = 1;
            ok &= emit_adj(S, cand, w[k], k, r);
        }
    }
    for (int64_t k = 1; k + 1 < n; ++k) {                     // pairs (k - 1, k + 1) around a low-MAPQ alignment
        if (T.used[(size_t)k - 1] || T.used[(size_t)k]) continue;
        if (!(w[k].bits_skip & 2)) continue;
        if ((row_in(w[k - 1], t1, s1, e1) && row_in(w[k + 1], t2, s2, e2)) || (row_in(w[k + 1], t1, s1, e1) && row_in(w[k - 1], t2, s2, e2)))
            ok &= emit(S, cand, 2 * (base + k) + 1, r, base);
    }
    return ok;
}

void run_calls(const Search &S, const int64_t *cand, int64_t n, Calls &c) {
    c = Calls();
    if (n == 0) return;
    const int64_t *ptr[13];
    int64_t stride[13];
    for (int f = 0; f < 13; ++f) { ptr[f] = cand + f; stride[f] = 13; }
    c.cluster_size.resize((size_t)n); c.flags.resize((size_t)n);
    c.head.resize((size_t)n); c.p1.resize((size_t)n); c.p2.resize((size_t)n);
    c.sup_off.resize((size_t)n + 1); c.sup_idx.resize((size_t)n); c.stats.resize(6 * (size_t)n);
    coral_call_breakpoints(n, ptr, stride, S.min_cluster_cutoff, S.bp_distance_cutoff, S.match_cutoff, S.accept_floor, 0,
                           &c.n_clusters, c.cluster_size.data(), &c.n_calls, c.head.data(), c.p1.data(), c.p2.data(), c.stats.data(),
                           c.flags.data(), c.sup_off.data(), c.sup_idx.data());
}

void compute_step(Search &S, Scratch &T, const int64_t key[5], StepResult &R) {
    const int64_t tid = key[0], s = key[1], e = key[2], si = key[3], ei = key[4];
    R.clear();
    auto fail = [&](int rc, const char *msg) { R.rc = rc; snprintf(R.err, sizeof(R.err), "%s", msg); };
    if (tid < 0 || tid >= S.n_tid) return fail(CORAL_ERR_ARG, "search_step: contig id out of range");
    // ---- reads hashed to segments si..ei of the contig, in the reference's visiting order (segment, then append order)
    const int64_t *lo = std::lower_bound(S.e_key, S.e_key + S.n_ent, (tid << 32) + si);
    const int64_t *hi = std::lower_bound(S.e_key, S.e_key + S.n_ent, (tid << 32) + ei + 1);
    if (lo == hi) return;
    const double t0 = S.profile ? now_s() : 0.0;
    if (T.seen.size() != (size_t)S.n_reads) { T.seen.assign((size_t)S.n_reads, 0u); T.stamp = 0; }
    if (++T.stamp == 0) { std::fill(T.seen.begin(), T.seen.end(), 0u); T.stamp = 1; }
    std::vector<PySetEmu> sets;
    std::vector<int64_t> codes;                          // contig << 32 | segment, per key, in order of first appearance
    std::unordered_map<int64_t, int32_t> key_of;
    long long n_adds = 0;
    int64_t last_code = -1;
    int32_t last_key = -1;
    auto add = [&](int64_t t, int64_t c, int64_t r) {
        const int64_t code = (t << 32) | c;
        int32_t k;
        if (code == last_code) {
            k = last_key;
        } else {
            auto it = key_of.find(code);
            if (it == key_of.end()) {
                k = (int32_t)sets.size();
                key_of.emplace(code, k);
                sets.emplace_back();
                codes.push_back(code);
            } else {
                k = it->second;
            }
            last_code = code;
            last_key = k;
        }
        sets[(size_t)k].add((int32_t)r, S.read_hash[r]);
        ++n_adds;
    };
    const int64_t *rows_e = S.e_row + (lo - S.e_key);
    const int64_t n_visit = hi - lo;
    for (int64_t v = 0; v < n_visit; ++v) {
        if (v + 12 < n_visit) __builtin_prefetch(&S.row_read[rows_e[v + 12]]);
        if (v + 5 < n_visit) {
            const int64_t rr = rows_e[v + 5];
            if (rr >= 0 && rr < S.n_rows) __builtin_prefetch(S.pack.data() + S.pack_off[(size_t)S.row_read[rr]]);
        }
        const int64_t row = rows_e[v];
        if (row < 0 || row >= S.n_rows) return fail(CORAL_ERR_ARG, "search_step: row out of range");
        const int64_t r = S.row_read[row];
        if (T.seen[(size_t)r] == T.stamp) continue;
        T.seen[(size_t)r] = T.stamp;
        const int32_t *rec = rec_of(S, r);
        const int64_t n = rec[0];
        const PackedRow *w = rows_of(rec);
        for (int64_t k = 0; k < n; ++k) {
            const int64_t t = w[k].tid, c0 = w[k].cni0, c1 = w[k].cni1;
            const bool other = t != tid;
            if (c0 >= 0 && (other || c0 <= si || c0 >= ei)) add(t, c0, r);              // Q9: the boundary segments count as outside
            if (c1 >= 0 && c1 != c0 && (other || c1 <= si || c1 >= ei)) add(t, c1, r);
        }
    }
    const double t1 = S.profile ? now_s() : 0.0;
    // ---- contigs in order of first appearance; per contig the surviving segments ascending, cut into runs (ibg:385-419)
    std::vector<int64_t> contig_order;
    std::unordered_map<int64_t, std::vector<std::pair<int64_t, int32_t>>> bins_of;       // contig -> (segment, key)
    for (size_t k = 0; k < codes.size(); ++k) {
        const int64_t t = codes[k] >> 32;
        auto it = bins_of.find(t);
        if (it == bins_of.end()) {
            contig_order.push_back(t);
            it = bins_of.emplace(t, std::vector<std::pair<int64_t, int32_t>>()).first;
        }
        if (!((double)sets[k].used < S.min_cluster_cutoff)) it->second.emplace_back(codes[k] & 0xFFFFFFFFLL, (int32_t)k);
    }
    struct Run { int64_t t, b0, b1; std::vector<int32_t> keys; };
    std::vector<Run> plan;
    for (int64_t t : contig_order) {
        auto &bins = bins_of[t];
        if (bins.empty()) continue;
        if (t < 0 || t >= S.n_tid) return fail(CORAL_ERR_ARG, "search_step: contig id out of range");
        std::sort(bins.begin(), bins.end());
        const int64_t *st = S.seg_start + S.seg_off[t], *en = S.seg_end + S.seg_off[t];
        const int64_t n_seg = S.seg_off[t + 1] - S.seg_off[t];
        Run cur{t, bins[0].first, bins[0].first, {}};
        for (size_t k = 0; k + 1 < bins.size(); ++k) {
            cur.keys.push_back(bins[k].second);
            const int64_t a = bins[k].first, b = bins[k + 1].first;
            if (a >= n_seg || b >= n_seg) return fail(CORAL_ERR_ARG, "search_step: segment index out of range");
            if (b - a > 2 || st[b] - en[a] > S.max_seq_len) {
                cur.b1 = a;
                plan.push_back(cur);
                cur = Run{t, b, b, {}};
            }
        }
        cur.keys.push_back(bins.back().second);
        cur.b1 = bins.back().first;
        if (cur.b1 >= n_seg || cur.b0 >= n_seg) return fail(CORAL_ERR_ARG, "search_step: segment index out of range");
        plan.push_back(cur);
    }
    // ---- per run: iteration order of  set() | sets[k0] | sets[k1] | ...  then alignment2bp of every read (bu:70-96)
    bool contigs_ok = true;
    double t_u = 0.0, t_c = 0.0;
    long long n_items = 0;
    R.calls.resize(plan.size());
    for (size_t g = 0; g < plan.size(); ++g) {
        const Run &run = plan[g];
        const double tu0 = S.profile ? now_s() : 0.0;
        PySetEmu acc;
        for (int32_t k : run.keys) acc.merge(sets[(size_t)k]);
        const double tu1 = S.profile ? now_s() : 0.0;
        n_items += (long long)acc.used;
        const int64_t t1_ = run.t, s1 = S.seg_start[S.seg_off[run.t] + run.b0], e1 = S.seg_end[S.seg_off[run.t] + run.b1];
        const size_t cand_before = R.cand.size(), order_before = R.order.size();
        R.order.reserve(order_before + acc.used);
        for (size_t slot_e = 0; slot_e <= acc.mask; ++slot_e)
            if (acc.key[slot_e] >= 0) R.order.push_back(acc.key[slot_e]);
        const int32_t *ord = R.order.data() + order_before;
        const size_t n_ord = R.order.size() - order_before;
        for (size_t q = 0; q < n_ord; ++q) {
            if (q + 6 < n_ord) __builtin_prefetch(S.pack.data() + S.pack_off[(size_t)ord[q + 6]]);
            contigs_ok &= pairs_between(S, T, R.cand, ord[q], t1_, s1, e1, tid, s, e);
        }
        R.order_off.push_back((int64_t)R.order.size());
        const int64_t n_c = (int64_t)((R.cand.size() - cand_before) / 13);
        const int64_t gr[4] = {run.t, run.b0, run.b1, n_c};
        R.groups.insert(R.groups.end(), gr, gr + 4);
        const double tu2 = S.profile ? now_s() : 0.0;
        t_u += tu1 - tu0;
        t_c += tu2 - tu1;
    }
    const double t2 = S.profile ? now_s() : 0.0;
    if (!contigs_ok) return fail(CORAL_ERR_FORMAT, "search_step: contig outside chr1..22,X,Y,M");
    // ---- cluster_bp_list + the bpc2bp loop of every run (ibg:436-457: the sub-cluster counter never advances there, Q4)
    {
        int64_t at = 0;
        for (size_t g = 0; g < plan.size(); ++g) {
            const int64_t n_c = R.groups[4 * g + 3];
            run_calls(S, R.cand.data() + 13 * at, n_c, R.calls[g]);
            at += n_c;
        }
    }
    R.flatten();
    if (S.profile) {
        const double t3 = now_s();
        std::lock_guard<std::mutex> lk(S.pm);
        S.t_reach += t1 - t0; S.t_union += t_u; S.t_cand += t_c; S.t_call += t3 - t2;
        ++S.n_steps; S.n_visit += n_visit; S.n_adds += n_adds; S.n_keys += (long long)codes.size(); S.n_union_items += n_items;
        S.n_cand += (long long)(R.cand.size() / 13);
    }
}

// CPUs of the NUMA node the calling thread runs on (empty when it cannot be told): the workers share the chimeric table with the
// caller, whose pages were first touched there; on the other socket every lookup of the search is a remote access.
std::vector<int> cpus_of_my_node() {
    std::vector<int> out;
    const int cpu = sched_getcpu();
    if (cpu < 0) return out;
    for (int node = 0; node < 64; ++node) {
        char path[96];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE *fp = fopen(path, "r");
        if (!fp) break;
        char buf[4096];
        std::vector<int> cpus;
        if (fgets(buf, sizeof(buf), fp)) {
            for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
                int a = 0, b = 0;
                const int n = sscanf(tok, "%d-%d", &a, &b);
                if (n == 1) b = a;
                if (n >= 1) for (int c = a; c <= b; ++c) cpus.push_back(c);
            }
        }
        fclose(fp);
        if (std::find(cpus.begin(), cpus.end(), cpu) != cpus.end()) return cpus;
    }
    return out;
}

void worker_main(Search *S) {
    Scratch T;
    for (;;) {
        std::shared_ptr<Entry> e;
        {
            std::unique_lock<std::mutex> lk(S->qm);
            S->qcv.wait(lk, [&] { return S->stop || !S->queue.empty(); });
            if (S->stop) return;
            e = S->queue.front();
            S->queue.pop_front();
        }
        {
            std::lock_guard<std::mutex> lk(e->m);
            if (e->taken) continue;                      // the caller got there first and computes it itself
            e->taken = true;
        }
        compute_step(*S, T, e->key, e->res);
        {
            std::lock_guard<std::mutex> lk(e->m);
            e->done = true;
        }
        e->cv.notify_all();
    }
}
}  // namespace

extern "C" void *coral_search_create(int64_t n_reads, int64_t n_rows, const int64_t *off, const int64_t *row_read,
                                     const int64_t *row_tid, const int64_t *ra, const int64_t *rb, const int64_t *cni0,
                                     const int64_t *cni1, const int64_t *read_hash, const int64_t *read_name, int64_t n_ent,
                                     const int64_t *e_key, const int64_t *e_row, const int32_t *pairs, int32_t n_tid,
                                     const int64_t *seg_off, const int64_t *seg_start, const int64_t *seg_end) {
    if (n_reads < 0 || n_rows < 0 || n_ent < 0 || n_tid < 0 || !off || !seg_off) return nullptr;
    if (n_rows > 0 && (!row_read || !row_tid || !ra || !rb || !cni0 || !cni1 || !pairs)) return nullptr;
    if (n_reads > 0 && (!read_hash || !read_name)) return nullptr;
    if (n_ent > 0 && (!e_key || !e_row)) return nullptr;
    Search *S = new Search();
    S->n_reads = n_reads; S->n_rows = n_rows; S->n_ent = n_ent;
    S->off = off; S->row_read = row_read; S->read_hash = read_hash; S->read_name = read_name; S->e_key = e_key; S->e_row = e_row;
    S->pairs = pairs; S->n_tid = n_tid; S->seg_off = seg_off; S->seg_start = seg_start; S->seg_end = seg_end;
    S->pack_off.resize((size_t)n_reads);
    S->pack.reserve((size_t)(2 * n_reads + 12 * n_rows));
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t base = off[r], n = off[r + 1] - base;
        if (n < 0 || base < 0 || base + n > n_rows) { delete S; return nullptr; }
        S->pack_off[(size_t)r] = (int64_t)S->pack.size();
        S->pack.push_back((int32_t)n);
        S->pack.push_back((int32_t)base);
        for (int64_t k = base; k < base + n; ++k) {
            const int32_t *adj = pairs + 8 * (2 * k);
            const int32_t w[12] = {(int32_t)row_tid[k], (int32_t)ra[k], (int32_t)rb[k], (int32_t)cni0[k], (int32_t)cni1[k],
                                   pairs[8 * (2 * k + 1) + 5], adj[0], adj[1], adj[2], adj[3], adj[4], adj[5]};
            S->pack.insert(S->pack.end(), w, w + 12);
        }
    }
    const char *pe = getenv("CORAL_SEARCH_PROFILE");
    S->profile = pe && pe[0] == '1';
    S->main_result.clear();
    S->current = &S->main_result;
    return S;
}

// Parameters of the build + the number of look-ahead threads (0 = every step is computed by the caller).
extern "C" int coral_search_params(void *h, double min_cluster_cutoff, int64_t max_seq_len, int64_t bp_distance_cutoff,
                                   int64_t match_cutoff, double accept_floor, int32_t n_threads) {
    if (!h || bp_distance_cutoff <= 0 || n_threads < 0 || n_threads > 64) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    if (!S.workers.empty() || !S.cache.empty()) return CORAL_ERR_ARG;          // set once, before the first step
    S.min_cluster_cutoff = min_cluster_cutoff; S.max_seq_len = max_seq_len; S.bp_distance_cutoff = bp_distance_cutoff;
    S.match_cutoff = match_cutoff; S.accept_floor = accept_floor;
    const std::vector<int> cpus = n_threads > 0 ? cpus_of_my_node() : std::vector<int>();
    for (int32_t k = 0; k < n_threads; ++k) {
        S.workers.emplace_back(worker_main, &S);
        if (!cpus.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
            (void)pthread_setaffinity_np(S.workers.back().native_handle(), sizeof(set), &set);
        }
    }
    return CORAL_OK;
}

extern "C" int coral_search_free(void *h) {
    if (!h) return CORAL_OK;
    Search &S = *(Search *)h;
    {
        std::lock_guard<std::mutex> lk(S.qm);
        S.stop = true;
    }
    S.qcv.notify_all();
    for (std::thread &t : S.workers) t.join();
    if (S.profile)
        fprintf(stderr, "coral_search: %lld steps (%lld computed by the caller, waited %.2f ms)  reach %.2f ms (visit rows %lld, set adds %lld, "
                "keys %lld)  union %.2f ms (%lld items)  candidates %.2f ms (%lld)  calls %.2f ms\n",
                S.n_steps, S.n_inline, S.t_wait * 1e3, S.t_reach * 1e3, S.n_visit, S.n_adds, S.n_keys, S.t_union * 1e3, S.n_union_items,
                S.t_cand * 1e3, S.n_cand, S.t_call * 1e3);
    delete &S;
    return CORAL_OK;
}

extern "C" const char *coral_search_error(void *h) { return h ? ((Search *)h)->err : "null handle"; }

// Ask for the step of interval (tid, s, e) on segments si..ei to be computed ahead (no-op without worker threads).
extern "C" int coral_search_prefetch(void *h, int64_t tid, int64_t s, int64_t e, int64_t si, int64_t ei) {
    if (!h) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    if (S.workers.empty()) return CORAL_OK;
    const std::array<int64_t, 5> key = {tid, s, e, si, ei};
    std::lock_guard<std::mutex> lk(S.qm);
    if (S.cache.find(key) != S.cache.end()) return CORAL_OK;
    auto ent = std::make_shared<Entry>();
    memcpy(ent->key, key.data(), sizeof(ent->key));
    S.cache.emplace(key, ent);
    S.queue.push_back(ent);
    S.qcv.notify_one();
    return CORAL_OK;
}

extern "C" int coral_search_step(void *h, int64_t tid, int64_t s, int64_t e, int64_t si, int64_t ei) {
    if (!h) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    const std::array<int64_t, 5> key = {tid, s, e, si, ei};
    std::shared_ptr<Entry> ent;
    {
        std::lock_guard<std::mutex> lk(S.qm);
        auto it = S.cache.find(key);
        if (it != S.cache.end()) ent = it->second;
    }
    S.current_entry.reset();
    if (ent) {
        bool mine = false;
        {
            std::unique_lock<std::mutex> lk(ent->m);
            if (!ent->taken) { ent->taken = true; mine = true; }           // still queued: do it here, the worker will skip it
        }
        if (mine) {
            compute_step(S, S.main_scratch, ent->key, ent->res);
            std::lock_guard<std::mutex> lk(ent->m);
            ent->done = true;
            ++S.n_inline;
        } else {
            const double w0 = S.profile ? now_s() : 0.0;
            std::unique_lock<std::mutex> lk(ent->m);
            ent->cv.wait(lk, [&] { return ent->done; });
            if (S.profile) S.t_wait += now_s() - w0;
        }
        S.current_entry = ent;
        S.current = &ent->res;
    } else {
        compute_step(S, S.main_scratch, key.data(), S.main_result);
        ++S.n_inline;
        S.current = &S.main_result;
    }
    if (S.current->rc != CORAL_OK) snprintf(S.err, sizeof(S.err), "%s", S.current->err);
    return S.current->rc;
}

// Arrays of the last result (owned by the handle, valid until the next step / within / between call on it).
extern "C" int coral_search_result(void *h, int64_t *n_meta, const int64_t **meta, int64_t *n_cand, const int64_t **cand,
                                   int64_t *n_sup, const int64_t **sup, const double **stats, const int64_t **order_off,
                                   const int32_t **order) {
    if (!h || !n_meta || !meta || !n_cand || !cand || !n_sup || !sup || !stats) return CORAL_ERR_ARG;
    const StepResult &R = *((Search *)h)->current;
    *n_meta = (int64_t)R.meta.size();
    *meta = R.meta.data();
    *n_cand = (int64_t)R.cand.size() / 13;
    *cand = R.cand.data();
    *n_sup = (int64_t)R.sup.size();
    *sup = R.sup.data();
    *stats = R.stats.data();
    if (order_off) *order_off = R.order_off.data();
    if (order) *order = R.order.data();
    return CORAL_OK;
}

// alignment2bp_l (bu:129-186) of every chimeric read, in table (= dict) order, against the interval list: both alignments of
// a pair must have the SAME first overlapping interval (interval_overlap_l, bu:37-44) and either change strand or fail the
// collinearity test.  Result: one group with all candidates (coral_search_result).
extern "C" int coral_search_within(void *h, int32_t n_int, const int64_t *int_tid, const int64_t *int_start, const int64_t *int_end) {
    if (!h || n_int < 0 || (n_int > 0 && (!int_tid || !int_start || !int_end))) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    StepResult &R = S.main_result;
    R.clear();
    S.current = &R;
    S.current_entry.reset();
    std::vector<std::vector<int32_t>> by_tid((size_t)S.n_tid);              // interval indices per contig, list order kept
    for (int32_t k = 0; k < n_int; ++k)
        if (int_tid[k] >= 0 && int_tid[k] < S.n_tid) by_tid[(size_t)int_tid[k]].push_back(k);
    auto first_interval = [&](const PackedRow &w) -> int32_t {
        if (w.tid < 0 || w.tid >= S.n_tid) return -1;
        for (int32_t k : by_tid[(size_t)w.tid])
            if (w.ra <= int_end[k] && int_start[k] <= w.rb) return k;
        return -1;
    };
    bool contigs_ok = true;
    std::vector<int32_t> fi;
    std::vector<char> &used = S.main_scratch.used;
    for (int64_t r = 0; r < S.n_reads; ++r) {
        const int32_t *rec = rec_of(S, r);
        const int64_t n = rec[0], base = rec[1];
        if (n < 2) continue;
        const PackedRow *w = rows_of(rec);
        fi.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) fi[(size_t)k] = first_interval(w[k]);
        used.assign((size_t)n, 0);
        for (int64_t k = 0; k + 1 < n; ++k) {
            const int32_t bits = w[k].bits_adj;
            if (!(bits & 2) || fi[(size_t)k] < 0 || fi[(size_t)k] != fi[(size_t)k + 1]) continue;
            if ((bits & 32) || (bits & 64)) {
                used[(size_t)k] = 1;
                contigs_ok &= emit_adj(S, R.cand, w[k], k, r);
            }
        }
        for (int64_t k = 1; k + 1 < n; ++k) {
            if (used[(size_t)k - 1] || used[(size_t)k]) continue;
            const int32_t bits = w[k].bits_skip;
            if (!(bits & 2) || fi[(size_t)k - 1] < 0 || fi[(size_t)k - 1] != fi[(size_t)k + 1]) continue;
            if ((bits & 32) || (bits & 64)) contigs_ok &= emit(S, R.cand, 2 * (base + k) + 1, r, base);
        }
    }
    const int64_t g[4] = {-1, -1, -1, (int64_t)(R.cand.size() / 13)};
    R.groups.insert(R.groups.end(), g, g + 4);
    R.order_off.push_back(0);
    R.flatten();
    if (!contigs_ok) { snprintf(S.err, sizeof(S.err), "search_within: contig outside chr1..22,X,Y,M"); return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}

// alignment2bp (bu:70-96) of the given reads between two intervals — the single query a search step runs per run;
// exported for the unit tests against the reference's own vectors and for callers that keep the reference's loop.
extern "C" int coral_search_between(void *h, int64_t n_sel, const int32_t *reads, int64_t t1, int64_t s1, int64_t e1, int64_t t2,
                                    int64_t s2, int64_t e2) {
    if (!h || n_sel < 0 || (n_sel > 0 && !reads)) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    StepResult &R = S.main_result;
    R.clear();
    S.current = &R;
    S.current_entry.reset();
    bool contigs_ok = true;
    for (int64_t q = 0; q < n_sel; ++q) {
        const int64_t r = reads[q];
        if (r < 0 || r >= S.n_reads) { snprintf(S.err, sizeof(S.err), "search_between: read index out of range"); return CORAL_ERR_ARG; }
        contigs_ok &= pairs_between(S, S.main_scratch, R.cand, r, t1, s1, e1, t2, s2, e2);
    }
    const int64_t g[4] = {t1, -1, -1, (int64_t)(R.cand.size() / 13)};
    R.groups.insert(R.groups.end(), g, g + 4);
    R.order_off.push_back(0);
    R.flatten();
    if (!contigs_ok) { snprintf(S.err, sizeof(S.err), "search_between: contig outside chr1..22,X,Y,M"); return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}
