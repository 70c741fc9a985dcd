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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
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
    double t_phase[4] = {0, 0, 0, 0};                      // CORAL_SEARCH_PROFILE: reach, union, candidates, calls (seconds, wall)
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
    double t_queued = 0, t_start = 0, t_end = 0;          // CORAL_SEARCH_PROFILE=2: when it was asked for / computed
    int who = -1;                                         // -1 a worker, 0 the caller
};

struct KeyHash {
    size_t operator()(const std::array<int64_t, 5> &k) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int64_t v : k) { h ^= (uint64_t)v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ull; }
        return (size_t)(h ^ (h >> 31));
    }
};

// A small work-sharing pool: a caller hands in a batch of tasks, helps to run tasks (its own or another caller's) while it
// waits, and returns when its batch is done.  Several steps may be computed at once (one per look-ahead thread), each handing
// in its batches; the helpers sleep on a condition variable between batches.
struct TaskPool {
    struct Task { std::function<void()> *fn; std::atomic<int> *left; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Task> q;
    bool stop = false;
    std::vector<std::thread> th;
    explicit TaskPool(int n) {
        for (int k = 0; k < n; ++k)
            th.emplace_back([this]() {
                for (;;) {
                    Task t;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [&] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        t = q.front();
                        q.pop_front();
                    }
                    (*t.fn)();
                    t.left->fetch_sub(1, std::memory_order_acq_rel);
                }
            });
    }
    ~TaskPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &x : th) x.join();
    }
    void run(std::vector<std::function<void()>> &tasks) {
        if (tasks.empty()) return;
        std::atomic<int> left((int)tasks.size());
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t k = 1; k < tasks.size(); ++k) q.push_back(Task{&tasks[k], &left});
        }
        cv.notify_all();
        tasks[0]();
        left.fetch_sub(1, std::memory_order_acq_rel);
        while (left.load(std::memory_order_acquire) > 0) {
            Task t{nullptr, nullptr};
            {
                std::lock_guard<std::mutex> lk(m);
                if (!q.empty()) { t = q.front(); q.pop_front(); }
            }
            if (t.fn) {
                (*t.fn)();
                t.left->fetch_sub(1, std::memory_order_acq_rel);
            } else {
                std::this_thread::yield();
            }
        }
    }
};

struct BfsOut;

struct Search {
    std::unique_ptr<TaskPool> pool;                       // helpers of big steps (coral_search_params: with the look-ahead threads)
    size_t chunk_cap = 9, chunk_reads = 1000;              // stage B of a step: at most chunk_cap chunks per run, of >= chunk_reads reads
    std::shared_ptr<BfsOut> bfs;                          // result of the last coral_search_bfs
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
    bool profile = false, profile_steps = false;
    int64_t par_min_reads = 3000, par_min_cands = 2000;      // a step splits work of at least this size over helper threads (CORAL_SEARCH_PAR_MIN: tests)
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
    // ---- per run: iteration order of  set() | sets[k0] | sets[k1] | ...  then alignment2bp of every read (bu:70-96), then
    // cluster_bp_list + the bpc2bp loop (ibg:436-457: the sub-cluster counter never advances there, Q4).  A step is on the
    // search's critical path (the breadth-first search often needs it the moment it discovers the interval), and the runs of a
    // step are independent of each other, so the three stages go through the handle's task pool when the step is big enough:
    //   A  per run: the union of its sets and the union's iteration order;
    //   B  per (run, chunk of consecutive reads): the pair filter into a list of its own;
    //   C  per run: its chunks' lists appended in chunk order (= the list a single pass gives) and the breakpoint calls.
    const size_t n_runs = plan.size();
    R.calls.resize(n_runs);
    std::vector<std::vector<int32_t>> order_of(n_runs);
    long long n_items = 0;
    const double tA0 = S.profile ? now_s() : 0.0;
    auto stage_a = [&](size_t g) {
        PySetEmu acc;
        for (int32_t k : plan[g].keys) acc.merge(sets[(size_t)k]);
        std::vector<int32_t> &ord = order_of[g];
        ord.reserve(acc.used);
        for (size_t slot_e = 0; slot_e <= acc.mask; ++slot_e)
            if (acc.key[slot_e] >= 0) ord.push_back(acc.key[slot_e]);
    };
    int64_t reads_in_sets = 0;
    for (size_t g = 0; g < n_runs; ++g)
        for (int32_t k : plan[g].keys) reads_in_sets += (int64_t)sets[(size_t)k].used;
    const bool pooled = S.pool && reads_in_sets >= S.par_min_reads;
    if (pooled && n_runs > 1) {
        std::vector<std::function<void()>> tasks;
        for (size_t g = 0; g < n_runs; ++g) tasks.emplace_back([&, g]() { stage_a(g); });
        S.pool->run(tasks);
    } else {
        for (size_t g = 0; g < n_runs; ++g) stage_a(g);
    }
    for (size_t g = 0; g < n_runs; ++g) n_items += (long long)order_of[g].size();
    const double tB0 = S.profile ? now_s() : 0.0;
    // stage B: chunks of ~2 000 reads (at most 8 per run)
    struct Chunk { size_t g, q0, q1; std::vector<int64_t> cand; bool ok = true; };
    std::vector<Chunk> chunks;
    for (size_t g = 0; g < n_runs; ++g) {
        const size_t n_ord = order_of[g].size();
        const size_t pieces = pooled ? std::max<size_t>(1, std::min<size_t>(S.chunk_cap, n_ord / S.chunk_reads)) : 1;
        for (size_t c = 0; c < pieces; ++c) chunks.push_back(Chunk{g, n_ord * c / pieces, n_ord * (c + 1) / pieces, {}, true});
    }
    auto stage_b = [&](Chunk &ch, Scratch &sc) {
        const Run &run = plan[ch.g];
        const int64_t t1_ = run.t, s1 = S.seg_start[S.seg_off[run.t] + run.b0], e1 = S.seg_end[S.seg_off[run.t] + run.b1];
        const int32_t *ord = order_of[ch.g].data();
        for (size_t q = ch.q0; q < ch.q1; ++q) {
            if (q + 6 < ch.q1) __builtin_prefetch(S.pack.data() + S.pack_off[(size_t)ord[q + 6]]);
            ch.ok &= pairs_between(S, sc, ch.cand, ord[q], t1_, s1, e1, tid, s, e);
        }
    };
    if (pooled && chunks.size() > 1) {
        std::vector<std::function<void()>> tasks;
        for (size_t c = 0; c < chunks.size(); ++c)
            tasks.emplace_back([&, c]() {
                Scratch sc;                                  // (pairs_between only uses the `used` marks: a few bytes)
                stage_b(chunks[c], sc);
            });
        S.pool->run(tasks);
    } else {
        for (Chunk &ch : chunks) stage_b(ch, T);
    }
    const double tC0 = S.profile ? now_s() : 0.0;
    bool contigs_ok = true;
    std::vector<int64_t> at(n_runs + 1, 0);
    for (const Chunk &ch : chunks) {
        at[ch.g + 1] += (int64_t)(ch.cand.size() / 13);
        contigs_ok &= ch.ok;
    }
    for (size_t g = 0; g < n_runs; ++g) at[g + 1] += at[g];
    R.cand.resize((size_t)at[n_runs] * 13);
    {
        std::vector<int64_t> w(at.begin(), at.end() - 1);
        for (const Chunk &ch : chunks) {
            if (!ch.cand.empty()) memcpy(R.cand.data() + 13 * w[ch.g], ch.cand.data(), ch.cand.size() * sizeof(int64_t));
            w[ch.g] += (int64_t)(ch.cand.size() / 13);
        }
    }
    for (size_t g = 0; g < n_runs; ++g) {
        R.order.insert(R.order.end(), order_of[g].begin(), order_of[g].end());
        R.order_off.push_back((int64_t)R.order.size());
        const int64_t gr[4] = {plan[g].t, plan[g].b0, plan[g].b1, at[g + 1] - at[g]};
        R.groups.insert(R.groups.end(), gr, gr + 4);
    }
    const double t2 = S.profile ? now_s() : 0.0;
    if (!contigs_ok) return fail(CORAL_ERR_FORMAT, "search_step: contig outside chr1..22,X,Y,M");
    {
        size_t n_big = 0;
        for (size_t g = 0; g < n_runs; ++g) n_big += (at[g + 1] - at[g]) >= S.par_min_cands ? 1 : 0;
        if (pooled && n_big > 1) {
            std::vector<std::function<void()>> tasks;
            for (size_t g = 0; g < n_runs; ++g) tasks.emplace_back([&, g]() { run_calls(S, R.cand.data() + 13 * at[g], at[g + 1] - at[g], R.calls[g]); });
            S.pool->run(tasks);
        } else {
            for (size_t g = 0; g < n_runs; ++g) run_calls(S, R.cand.data() + 13 * at[g], at[g + 1] - at[g], R.calls[g]);
        }
    }
    const double t_u = tB0 - tA0, t_c = t2 - tB0;
    (void)tC0;
    R.flatten();
    if (S.profile) {
        const double t3 = now_s();
        R.t_phase[0] = t1 - t0; R.t_phase[1] = t_u; R.t_phase[2] = t_c; R.t_phase[3] = t3 - t2;
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
        e->t_start = now_s();
        compute_step(*S, T, e->key, e->res);
        e->t_end = now_s();
        {
            std::lock_guard<std::mutex> lk(e->m);
            e->done = true;
        }
        e->cv.notify_all();
    }
}

// ---------------------------------------------------------------------------------------------
// The whole interval search (find_amplicon_intervals' loop over the seeds + find_interval_i, ibg:343-673) on index arrays:
// the order-dependent half — addbp (ibg:326-340), the inside / outside refinement of the target intervals (ibg:459-612),
// interval_exclusive and the connection bookkeeping (ibg:614-673) — next to the pure steps above, so that a build makes ONE
// native call for the search instead of one round trip per interval.  Every quirk of the reference the Python host logic
// reproduced is reproduced here (Q2 the comparison result stored in `l`, Q3 the CN value used as a
// truth value, Q5 the and / or precedence, Q9, Q10, Q13; exceptions swallowed by the reference's try / except leave the
// same partial state).  Containers whose ITERATION ORDER is observable are replayed: the `hit` set of interval_exclusive is
// a CPython set of small ints (PySetEmu with hash(i) = i), dict key order = first insertion.
// ---------------------------------------------------------------------------------------------
struct BfsBp {
    int64_t f[11];                                       // c1 p1 o1 c2 p2 o2 head-name-id i j gap swapped
    double stats[6];
    int32_t flags;
    int64_t ccid;
    std::vector<std::array<int64_t, 2>> chunks;          // [begin, end) into chunk_read / chunk_i / chunk_j
};

struct BfsOut {
    std::vector<int64_t> iv;                             // [n][5]: tid, s, e, ccid, s-is-a-bool (Q2)
    std::vector<BfsBp> bps;
    std::vector<int64_t> chunk_read, chunk_i, chunk_j;
    std::vector<std::array<int64_t, 2>> conn_key;        // insertion order
    std::vector<std::vector<int64_t>> conn_val;          // adds in order (duplicates kept: a set on the Python side)
    std::vector<int64_t> events;                         // [n][6]
    // flattened for retrieval
    std::vector<int64_t> bp_flat, bp_meta, chunk_off, conn_flat, conn_off, conn_vals;
    std::vector<double> bp_stats;
    int64_t err_tid = -1;
};

struct BfsParams {
    const double *seg_cn;
    const int64_t *seg_ix;
    const int32_t *chr_rank;
    const uint8_t *tid_has_rows;
    double cn_gain;
    int64_t D;
    bool log_events, log_debug;
};

enum { BFS_ERR_KEY_CHROM = -10, BFS_ERR_KEY_CHRIDX = -11, BFS_ERR_INDEX = -12 };

struct BfsRun {
    Search &S;
    const BfsParams &P;
    BfsOut &O;
    BfsRun(Search &s, const BfsParams &p, BfsOut &o) : S(s), P(p), O(o) {}
    struct MemoHash {
        size_t operator()(const std::array<int64_t, 3> &k) const {
            uint64_t h = (uint64_t)k[0] * 0x9E3779B97F4A7C15ull;
            h ^= ((uint64_t)k[1] + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
            h ^= ((uint64_t)k[2] + 0x94D049BBull) * 0x94D049BB133111EBull;
            return (size_t)(h ^ (h >> 29));
        }
    };
    std::vector<std::shared_ptr<Entry>> held;
    std::unordered_map<std::array<int64_t, 3>, std::array<int64_t, 2>, MemoHash> chunk_memo;     // (step result, run, call) -> chunk
    double t_steps = 0;                                      // CORAL_SEARCH_PROFILE: seconds inside coral_search_step (waits included)

    int64_t nseg(int64_t t) const { return S.seg_off[t + 1] - S.seg_off[t]; }
    int64_t sstart(int64_t t, int64_t k) const { return S.seg_start[S.seg_off[t] + k]; }
    int64_t send(int64_t t, int64_t k) const { return S.seg_end[S.seg_off[t] + k]; }
    double scn(int64_t t, int64_t k) const { return P.seg_cn[S.seg_off[t] + k]; }
    // pos2cni(chr, pos)[0]: index (the reference's per-chromosome idx, Q8) of the first CN segment, in file order, containing pos;
    // -1 = empty list, -2 = the chromosome has no CN segments at all (KeyError on cns_tree[chr])
    int64_t pos2cni(int64_t t, int64_t pos) const {
        if (t < 0 || t >= S.n_tid || nseg(t) == 0) return -2;
        const int64_t n = nseg(t), *st = S.seg_start + S.seg_off[t], *en = S.seg_end + S.seg_off[t];
        for (int64_t k = 0; k < n; ++k)
            if (st[k] <= pos && pos <= en[k]) return P.seg_ix[S.seg_off[t] + k];
        return -1;
    }
    bool pos2cni_any(int64_t t, int64_t pos) const { return pos2cni(t, pos) >= 0; }
    void event(int64_t ty, int64_t a = 0, int64_t b = 0, int64_t c = 0, int64_t d = 0, int64_t e = 0) {
        if (!P.log_events) return;
        const int64_t row[6] = {ty, a, b, c, d, e};
        O.events.insert(O.events.end(), row, row + 6);
    }
    static bool overlap(int64_t at, int64_t a1, int64_t a2, int64_t bt, int64_t b1, int64_t b2) {
        return at == bt && a1 <= b2 && b1 <= a2;             // interval_overlap (bu:11-15)
    }
    int64_t conn_index(int64_t a, int64_t b, bool create) {
        for (size_t k = 0; k < O.conn_key.size(); ++k)
            if (O.conn_key[k][0] == a && O.conn_key[k][1] == b) return (int64_t)k;
        if (!create) return -1;
        O.conn_key.push_back({a, b});
        O.conn_val.emplace_back();
        return (int64_t)O.conn_key.size() - 1;
    }
    void conn_add(int64_t a, int64_t b, int64_t k) {         // conn.setdefault((min, max), set()).add(k)
        const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
        O.conn_val[(size_t)conn_index(lo, hi, true)].push_back(k);
    }

    int enqueue(int64_t idx) {                               // _prefetch_step
        const int64_t t = O.iv[5 * idx], s = O.iv[5 * idx + 1], e = O.iv[5 * idx + 2];
        const int64_t si = pos2cni(t, s), ei = pos2cni(t, e);
        if (si < 0 || ei < 0) return CORAL_OK;
        if (!P.tid_has_rows[t]) return CORAL_OK;
        return coral_search_prefetch(&S, t, s, e, si, ei);
    }

    // addbp (ibg:326-340): merge into the first breakpoint with the same ends within 200 bp, else append
    int64_t addbp(const int64_t f[11], const double stats[6], int32_t flags, int64_t ccid, const std::array<int64_t, 2> &chunk) {
        for (size_t k = 0; k < O.bps.size(); ++k) {
            const int64_t *b = O.bps[k].f;
            if (b[0] == f[0] && b[3] == f[3] && b[2] == f[2] && b[5] == f[5] && llabs(b[1] - f[1]) < 200 && llabs(b[4] - f[4]) < 200) {
                O.bps[k].chunks.push_back(chunk);
                return (int64_t)k;
            }
        }
        BfsBp nb;
        memcpy(nb.f, f, sizeof(nb.f));
        memcpy(nb.stats, stats, sizeof(nb.stats));
        nb.flags = flags;
        nb.ccid = ccid;
        nb.chunks.push_back(chunk);
        O.bps.push_back(std::move(nb));
        return (int64_t)O.bps.size() - 1;
    }

    int bfs_from(int64_t ai, int64_t ccid) {
        const int64_t half = (int64_t)((double)S.max_seq_len / 2.0);
        const int64_t D = P.D;
        const double gain = P.cn_gain;
        std::deque<int64_t> queue{ai};
        while (!queue.empty()) {
            const int64_t cur = queue.front();
            queue.pop_front();
            const int64_t tid = O.iv[5 * cur], s = O.iv[5 * cur + 1], e = O.iv[5 * cur + 2];
            if (O.iv[5 * cur + 3] == -1) O.iv[5 * cur + 3] = ccid;
            event(0, cur, tid, s, e, O.iv[5 * cur + 3]);
            // ---- _search_step
            const int64_t si = pos2cni(tid, s), ei = pos2cni(tid, e);
            if (si < 0 || ei < 0) continue;
            if (!P.tid_has_rows[tid]) { O.err_tid = tid; return BFS_ERR_KEY_CHROM; }
            const double ts0 = S.profile ? now_s() : 0.0;
            int rc = coral_search_step(&S, tid, s, e, si, ei);
            if (S.profile) t_steps += now_s() - ts0;
            if (rc != CORAL_OK) return rc;
            if (S.current_entry) held.push_back(S.current_entry);      // (step results stay alive: chunk_memo is keyed by their address)
            std::shared_ptr<Entry> hold = S.current_entry;    // the result stays alive while later prefetches touch the cache
            const StepResult &R = *S.current;
            const size_t ng = R.groups.size() / 4;
            const int64_t here_t = tid, here_s = s, here_e = e;      // `here` is not modified before all groups are done
            struct Ref { int64_t t, l, r; bool l_bool; std::vector<int64_t> bps; };
            std::vector<Ref> refined;
            int64_t cand_at = 0;
            for (size_t gi = 0; gi < ng; ++gi) {
                const int64_t c = R.groups[4 * gi], b0 = R.groups[4 * gi + 1], b1 = R.groups[4 * gi + 2], n_c = R.groups[4 * gi + 3];
                const int64_t *cand = R.cand.data() + 13 * cand_at;
                cand_at += n_c;
                if (c < 0 || c >= S.n_tid || b0 < 0 || b1 < 0 || b0 >= nseg(c) || b1 >= nseg(c)) return BFS_ERR_INDEX;
                const int64_t ns = sstart(c, b0), ne = send(c, b1);
                event(1, n_c);
                const Calls &calls = R.calls[gi];
                if (P.log_debug) for (int32_t q = 0; q < calls.n_clusters; ++q) event(2, calls.cluster_size[(size_t)q]);
                std::vector<int64_t> found;
                for (int32_t q = 0; q < calls.n_calls; ++q) {
                    const int64_t head = calls.head[(size_t)q];
                    const int64_t *h = cand + 13 * head;
                    const int64_t f[11] = {h[0], calls.p1[(size_t)q], h[2], h[3], calls.p2[(size_t)q], h[5], h[6], h[7], h[8], h[9], h[10]};
                    // the support triples of (this step, this run, this call): written once — the search visits an interval as often
                    // as it was queued (the reference's queue holds duplicates), and every visit unites the same supports again
                    std::array<int64_t, 2> chunk;
                    // (a step computed without look-ahead threads lives in the handle's one result slot: no stable identity, no memo)
                    const bool memo = (bool)hold;
                    const std::array<int64_t, 3> memo_key = {(int64_t)(intptr_t)hold.get(), (int64_t)gi, (int64_t)q};
                    auto mit = memo ? chunk_memo.find(memo_key) : chunk_memo.end();
                    if (mit != chunk_memo.end()) {
                        chunk = mit->second;
                    } else {
                        const int64_t c0 = (int64_t)O.chunk_read.size();
                        const int64_t u0 = calls.sup_off[(size_t)q], u1 = calls.sup_off[(size_t)q + 1];
                        O.chunk_read.resize((size_t)(c0 + u1 - u0));
                        O.chunk_i.resize((size_t)(c0 + u1 - u0));
                        O.chunk_j.resize((size_t)(c0 + u1 - u0));
                        for (int64_t u = u0; u < u1; ++u) {
                            const int64_t *m = cand + 13 * calls.sup_idx[(size_t)u];
                            O.chunk_read[(size_t)(c0 + u - u0)] = m[6];
                            O.chunk_i[(size_t)(c0 + u - u0)] = m[7];
                            O.chunk_j[(size_t)(c0 + u - u0)] = m[8];
                        }
                        chunk = {c0, c0 + u1 - u0};
                        if (memo) chunk_memo.emplace(memo_key, chunk);
                    }
                    const int64_t k = addbp(f, calls.stats.data() + 6 * (size_t)q, calls.flags[(size_t)q], ccid, chunk);
                    if (std::find(found.begin(), found.end(), k) == found.end()) found.push_back(k);
                }
                struct In { int64_t cni, pos, k; };
                struct Out { int64_t t, cni, pos, k; };
                std::vector<In> inside;
                std::vector<Out> outside;
                for (int64_t k : found) {
                    const int64_t *bp = O.bps[(size_t)k].f;
                    const int64_t t1 = bp[0], p1 = bp[1], t2 = bp[3], p2 = bp[4];
                    // (an IndexError / KeyError of pos2cni(..)[0] ends this breakpoint's turn — what was appended before stays)
                    if (overlap(t1, p1, p1, here_t, here_s, here_e) && overlap(t2, p2, p2, c, ns, ne)) {
                        const int64_t q = pos2cni(t2, p2);
                        if (q >= 0) inside.push_back({q, p2, k});
                    } else if (overlap(t2, p2, p2, here_t, here_s, here_e) && overlap(t1, p1, p1, c, ns, ne)) {
                        const int64_t q = pos2cni(t1, p1);
                        if (q >= 0) inside.push_back({q, p1, k});
                    } else {
                        event(3);
                        const bool o1 = overlap(t1, p1, p1, c, ns, ne), o2 = overlap(t2, p2, p2, c, ns, ne);
                        const int64_t q1 = pos2cni(t1, p1);
                        if (q1 < 0) continue;
                        if (o1) inside.push_back({q1, p1, k}); else outside.push_back({t1, q1, p1, k});
                        const int64_t q2 = pos2cni(t2, p2);
                        if (q2 < 0) continue;
                        if (o2) inside.push_back({q2, p2, k}); else outside.push_back({t2, q2, p2, k});
                    }
                }
                if (found.empty()) continue;
                std::stable_sort(inside.begin(), inside.end(), [](const In &a, const In &b) { return a.cni != b.cni ? a.cni < b.cni : a.pos < b.pos; });
                for (const Out &o : outside)
                    if (o.t < 0 || o.t >= S.n_tid || P.chr_rank[o.t] < 0) { O.err_tid = o.t; return BFS_ERR_KEY_CHRIDX; }
                std::stable_sort(outside.begin(), outside.end(), [&](const Out &a, const Out &b) {
                    const int32_t ra = P.chr_rank[a.t], rb = P.chr_rank[b.t];
                    if (ra != rb) return ra < rb;
                    if (a.cni != b.cni) return a.cni < b.cni;
                    return a.pos < b.pos;
                });
                const int64_t n_c_seg = nseg(c);
                auto seg_ok = [&](int64_t t, int64_t k) { return k >= 0 && k < nseg(t); };
                for (const In &x : inside) if (!seg_ok(c, x.cni)) return BFS_ERR_INDEX;
                for (const Out &x : outside) if (!seg_ok(x.t, x.cni)) return BFS_ERR_INDEX;
                if (n_c_seg == 0) return BFS_ERR_INDEX;
                // ---- runs of `inside` breakpoints -> refined intervals on contig c (ibg:484-546)
                auto split_inside = [&](size_t k) {
                    const int64_t nil = sstart(c, inside[k + 1].cni), lir = send(c, inside[k].cni);
                    const double ncn = scn(c, inside[k + 1].cni), lcn = scn(c, inside[k].cni);
                    const bool amp = ncn >= gain || lcn >= gain;
                    const int64_t dpos = inside[k + 1].pos - inside[k].pos;
                    return inside[k + 1].cni - inside[k].cni > 2 || (double)(nil - lir) > (double)S.max_seq_len / 2.0 || dpos > S.max_seq_len ||
                           (!amp && nil - lir > 2 * D) || (!amp && dpos > 3 * D);
                };
                size_t first = 0;
                for (size_t k = 0; k + 1 < inside.size(); ++k) {
                    if (!split_inside(k)) continue;
                    const In &f = inside[first], &z = inside[k];
                    const int64_t lir = send(c, z.cni);
                    int64_t l = std::max((!(scn(c, f.cni) >= gain) ? f.pos : sstart(c, f.cni)) - D, sstart(c, 0));
                    int64_t r = std::min((!(scn(c, z.cni) >= gain) ? z.pos : lir) + D, send(c, n_c_seg - 1));
                    const double fcn = scn(c, f.cni);
                    if ((fcn != 0.0) && f.pos - half > l) l = f.pos - half;                 // Q3: the CN value as a truth value (NaN is true)
                    if (z.pos + half < r) r = z.pos + half;
                    if (!pos2cni_any(c, l)) l = sstart(c, f.cni);
                    if (!pos2cni_any(c, r)) r = lir;
                    Ref ref{c, l, r, false, {}};
                    for (size_t j = first; j <= k; ++j) ref.bps.push_back(inside[j].k);
                    refined.push_back(std::move(ref));
                    first = k + 1;
                }
                if (!inside.empty()) {
                    const In &f = inside[first], &z = inside.back();
                    int64_t l = std::max((!(scn(c, f.cni) >= gain) ? f.pos : sstart(c, f.cni)) - D, sstart(c, 0));
                    int64_t r = std::min((!(scn(c, z.cni) >= gain) ? z.pos : send(c, z.cni)) + D, send(c, n_c_seg - 1));
                    bool l_bool = false;
                    if (f.pos - half > l) { l = (f.pos - half > l) ? 1 : 0; l_bool = true; }          // Q2: the comparison result is stored
                    if (z.pos + half < r) r = z.pos + half;
                    if (!pos2cni_any(c, l)) { l = sstart(c, f.cni); l_bool = false; }
                    if (!pos2cni_any(c, r)) r = send(c, z.cni);
                    Ref ref{c, l, r, l_bool, {}};
                    for (size_t j = first; j < inside.size(); ++j) ref.bps.push_back(inside[j].k);
                    refined.push_back(std::move(ref));
                }
                // ---- runs of `outside` ends -> refined intervals on their own contigs (ibg:548-612)
                auto split_outside = [&](size_t k) {
                    const Out &a = outside[k], &b = outside[k + 1];
                    const int64_t nil = sstart(b.t, b.cni), lir = send(a.t, a.cni);
                    const double ncn = scn(b.t, b.cni), lcn = scn(a.t, a.cni);
                    const bool amp = ncn >= gain || lcn >= gain;
                    return b.t != a.t || b.cni - a.cni > 2 || (double)(nil - lir) > (double)S.max_seq_len / 2.0 || b.pos - a.pos > S.max_seq_len ||
                           (!amp && nil - lir > 2 * D) || (!amp && b.pos - a.pos > 3 * D);
                };
                first = 0;
                for (size_t k = 0; k + 1 < outside.size(); ++k) {
                    if (!split_outside(k)) continue;
                    const Out &f = outside[first], &z = outside[k];
                    const int64_t lir = send(z.t, z.cni);
                    int64_t l = std::max((!(scn(f.t, f.cni) >= gain) ? f.pos : sstart(f.t, f.cni)) - D, sstart(f.t, 0));
                    int64_t r = std::min((!(scn(z.t, z.cni) >= gain) ? z.pos : lir) + D, send(z.t, nseg(z.t) - 1));
                    if (f.pos - half > l) l = f.pos - half;
                    if (z.pos + half < r) r = z.pos + half;
                    if (!pos2cni_any(f.t, l)) l = sstart(f.t, f.cni);
                    if (!pos2cni_any(z.t, r)) r = lir;
                    refined.push_back(Ref{f.t, l, r, false, {}});
                    first = k + 1;
                }
                if (!outside.empty()) {
                    const Out &f = outside[first], &z = outside.back();
                    int64_t l = std::max((!(scn(f.t, f.cni) >= gain) ? f.pos : sstart(f.t, f.cni)) - D, sstart(f.t, 0));
                    int64_t r = std::min((!(scn(z.t, z.cni) >= gain) ? z.pos : send(z.t, z.cni)) + D, send(z.t, nseg(z.t) - 1));
                    if (f.pos - half > l) l = f.pos - half;
                    if (z.pos + half < r) r = z.pos + half;
                    if (!pos2cni_any(f.t, l)) l = sstart(f.t, f.cni);
                    if (!pos2cni_any(f.t, r)) {
                        if (!seg_ok(f.t, z.cni)) return BFS_ERR_INDEX;
                        r = send(f.t, z.cni);                                            // (by[f[0]][z[1]]: the FIRST end's contig, as written)
                    }
                    refined.push_back(Ref{f.t, l, r, false, {}});
                }
            }
            // ---- the refined intervals against the interval list (ibg:614-673)
            for (const Ref &ref : refined) {
                // interval_exclusive (bu:54-67): parts of the candidate not covered by the list + the (CPython) set of intervals it overlaps
                PySetEmu hit;
                struct Part { int64_t s, e; };
                std::vector<Part> parts{{ref.l, ref.r}};
                const int64_t n_iv = (int64_t)(O.iv.size() / 5);
                for (int64_t k = 0; k < n_iv; ++k) {
                    const int64_t bt = O.iv[5 * k], bs = O.iv[5 * k + 1], be = O.iv[5 * k + 2];
                    for (int64_t j = (int64_t)parts.size() - 1; j >= 0; --j) {
                        const Part p = parts[(size_t)j];
                        if (overlap(ref.t, p.s, p.e, bt, bs, be)) {
                            hit.add((int32_t)k, (int64_t)k);
                            parts.erase(parts.begin() + j);
                            if (p.s < bs) parts.push_back({p.s, bs - 1});
                            if (p.e > be) parts.push_back({be + 1, p.e});
                        }
                    }
                }
                std::vector<int64_t> hits;
                for (size_t slot = 0; slot <= hit.mask; ++slot)
                    if (hit.key[slot] >= 0) hits.push_back(hit.key[slot]);
                auto bp_touches = [&](const int64_t *bp, int64_t o, bool first_end) {
                    const int64_t t = first_end ? bp[0] : bp[3], p = first_end ? bp[1] : bp[4];
                    return overlap(t, p, p, O.iv[5 * o], O.iv[5 * o + 1], O.iv[5 * o + 2]);
                };
                if (parts.empty()) {
                    for (int64_t k : ref.bps) {
                        const int64_t *bp = O.bps[(size_t)k].f;
                        for (int64_t o : hits)
                            if ((o != cur && bp_touches(bp, o, true)) || bp_touches(bp, o, false)) conn_add(cur, o, k);        // Q5
                    }
                    for (int64_t o : hits)
                        if (o != cur && O.iv[5 * o + 3] < 0) {
                            queue.push_back(o);
                            const int rc2 = enqueue(o);
                            if (rc2 != CORAL_OK) return rc2;
                        }
                } else {
                    for (const Part &part : parts) {
                        const int64_t nai = (int64_t)(O.iv.size() / 5);
                        // (a part that still starts at the candidate's own `l` keeps its bool-ness, Q2)
                        const int64_t row[5] = {ref.t, part.s, part.e, -1, (ref.l_bool && part.s == ref.l) ? 1 : 0};
                        O.iv.insert(O.iv.end(), row, row + 5);
                        event(4, ref.t, part.s, part.e, row[4]);
                        const int64_t ck = conn_index(cur, nai, true);
                        O.conn_val[(size_t)ck].clear();                                  // conn[(cur, nai)] = set()
                        for (int64_t k : ref.bps) {
                            if (hits.empty()) {
                                O.conn_val[(size_t)conn_index(cur, nai, false)].push_back(k);
                                continue;
                            }
                            const int64_t *bp = O.bps[(size_t)k].f;
                            for (int64_t o : hits) {
                                if (bp_touches(bp, o, true) || bp_touches(bp, o, false)) conn_add(cur, o, k);
                                else O.conn_val[(size_t)conn_index(cur, nai, false)].push_back(k);
                            }
                        }
                        queue.push_back(nai);
                        const int rc2 = enqueue(nai);
                        if (rc2 != CORAL_OK) return rc2;
                    }
                }
            }
        }
        return CORAL_OK;
    }
};
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
    {   // offsets first (2 + 12 ints per row), then the records are filled in parallel: 16 MB at 2 M reads, on the build's critical path
        int64_t at = 0;
        for (int64_t r = 0; r < n_reads; ++r) {
            const int64_t base = off[r], n = off[r + 1] - base;
            if (n < 0 || base < 0 || base + n > n_rows) { delete S; return nullptr; }
            S->pack_off[(size_t)r] = at;
            at += 2 + 12 * n;
        }
        S->pack.resize((size_t)at);
        auto fill = [&](int64_t r0, int64_t r1) {
            for (int64_t r = r0; r < r1; ++r) {
                const int64_t base = off[r], n = off[r + 1] - base;
                int32_t *w = S->pack.data() + S->pack_off[(size_t)r];
                *w++ = (int32_t)n;
                *w++ = (int32_t)base;
                for (int64_t k = base; k < base + n; ++k) {
                    const int32_t *adj = pairs + 8 * (2 * k);
                    const int32_t v[12] = {(int32_t)row_tid[k], (int32_t)ra[k], (int32_t)rb[k], (int32_t)cni0[k], (int32_t)cni1[k],
                                           pairs[8 * (2 * k + 1) + 5], adj[0], adj[1], adj[2], adj[3], adj[4], adj[5]};
                    memcpy(w, v, sizeof(v));
                    w += 12;
                }
            }
        };
        const int nt = n_reads >= 40000 ? 4 : 1;
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(fill, n_reads * t / nt, n_reads * (t + 1) / nt);
        fill(0, n_reads / nt);
        for (auto &x : th) x.join();
    }
    const char *pe = getenv("CORAL_SEARCH_PROFILE");
    S->profile = pe && (pe[0] == '1' || pe[0] == '2');
    S->profile_steps = pe && pe[0] == '2';
    if (const char *pm = getenv("CORAL_SEARCH_PAR_MIN")) S->par_min_reads = S->par_min_cands = atoll(pm);
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
    if (n_threads > 0) {
        // 8 helpers where the cores are there (with the look-ahead threads and the caller: 15 threads); fewer on small hosts
        const unsigned hw = std::thread::hardware_concurrency();
        int helpers = hw >= 16 ? 8 : hw >= 8 ? 3 : 1;      // (CORAL_SEARCH_HELPERS, CORAL_SEARCH_CHUNK: tuning)
        if (const char *e = getenv("CORAL_SEARCH_HELPERS")) helpers = std::max(0, std::min(15, atoi(e)));
        if (const char *e = getenv("CORAL_SEARCH_CHUNK")) S.chunk_reads = (size_t)std::max(200, atoi(e));
        S.chunk_cap = (size_t)std::max(8, helpers + 1);
        if (helpers > 0) S.pool.reset(new TaskPool(helpers));
    }
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
    ent->t_queued = now_s();
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
        const double t_ask = now_s();
        if (mine) {
            ent->t_start = t_ask;
            ent->who = 0;
            compute_step(S, S.main_scratch, ent->key, ent->res);
            ent->t_end = now_s();
            std::lock_guard<std::mutex> lk(ent->m);
            ent->done = true;
            ++S.n_inline;
        } else {
            const double w0 = S.profile ? now_s() : 0.0;
            std::unique_lock<std::mutex> lk(ent->m);
            ent->cv.wait(lk, [&] { return ent->done; });
            if (S.profile) S.t_wait += now_s() - w0;
        }
        if (S.profile_steps)
            fprintf(stderr, "  step contig %lld [%lld, %lld]: queued %.2f ms before it was asked for, compute %.2f ms (%s), asked -> ready %.2f ms; "
                    "%zu runs, %zu candidates, %zu reads in the runs; reach %.2f union %.2f candidates %.2f calls %.2f ms\n", (long long)key[0], (long long)key[1], (long long)key[2],
                    (t_ask - ent->t_queued) * 1e3, (ent->t_end - ent->t_start) * 1e3, ent->who == 0 ? "caller" : "worker", (now_s() - t_ask) * 1e3,
                    ent->res.groups.size() / 4, ent->res.cand.size() / 13, ent->res.order.size(), ent->res.t_phase[0] * 1e3, ent->res.t_phase[1] * 1e3,
                    ent->res.t_phase[2] * 1e3, ent->res.t_phase[3] * 1e3);
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
    // reads in table (= dict) order; big tables are cut into consecutive ranges of reads filtered on the handle's task pool, each
    // into a list of its own, appended in range order
    auto scan_reads = [&](int64_t r0, int64_t r1, std::vector<int64_t> &out) {
        bool ok = true;
        std::vector<int32_t> fi;
        std::vector<char> used;
        for (int64_t r = r0; r < r1; ++r) {
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
                    ok &= emit_adj(S, out, w[k], k, r);
                }
            }
            for (int64_t k = 1; k + 1 < n; ++k) {
                if (used[(size_t)k - 1] || used[(size_t)k]) continue;
                const int32_t bits = w[k].bits_skip;
                if (!(bits & 2) || fi[(size_t)k - 1] < 0 || fi[(size_t)k - 1] != fi[(size_t)k + 1]) continue;
                if ((bits & 32) || (bits & 64)) ok &= emit(S, out, 2 * (base + k) + 1, r, base);
            }
        }
        return ok;
    };
    bool contigs_ok = true;
    const int pieces = (S.pool && S.n_reads >= 4 * S.par_min_reads) ? 4 : 1;
    if (pieces == 1) {
        contigs_ok = scan_reads(0, S.n_reads, R.cand);
    } else {
        std::vector<std::vector<int64_t>> part((size_t)pieces);
        std::vector<char> okv((size_t)pieces, 1);
        std::vector<std::function<void()>> tasks;
        for (int c = 0; c < pieces; ++c)
            tasks.emplace_back([&, c]() { okv[(size_t)c] = scan_reads(S.n_reads * c / pieces, S.n_reads * (c + 1) / pieces, part[(size_t)c]); });
        S.pool->run(tasks);
        for (int c = 0; c < pieces; ++c) {
            contigs_ok &= okv[(size_t)c] != 0;
            R.cand.insert(R.cand.end(), part[(size_t)c].begin(), part[(size_t)c].end());
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

// The interval search of one build in ONE call: ibg's find_amplicon_intervals loop over the seed intervals + find_interval_i.
// iv: int64 [n_seed][4] = contig id, start, end, ccid (-1) of the seed intervals after the CN-segment snap (ibg:343-360).
extern "C" int coral_search_bfs(void *h, int32_t n_seed, const int64_t *iv, const double *seg_cn, const int64_t *seg_ix,
                                const int32_t *chr_rank, const uint8_t *tid_has_rows, double cn_gain, int64_t interval_delta,
                                int32_t log_level) {
    if (!h || n_seed < 0 || (n_seed > 0 && !iv) || !seg_cn || !seg_ix || !chr_rank || !tid_has_rows) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    S.bfs = std::make_shared<BfsOut>();
    BfsOut &O = *S.bfs;
    for (int32_t k = 0; k < n_seed; ++k) {
        if (iv[4 * k] < 0 || iv[4 * k] >= S.n_tid) { snprintf(S.err, sizeof(S.err), "search_bfs: contig id out of range"); return CORAL_ERR_ARG; }
        const int64_t row[5] = {iv[4 * k], iv[4 * k + 1], iv[4 * k + 2], iv[4 * k + 3], 0};
        O.iv.insert(O.iv.end(), row, row + 5);
    }
    const BfsParams P{seg_cn, seg_ix, chr_rank, tid_has_rows, cn_gain, interval_delta, log_level >= 1, log_level >= 2};
    BfsRun run(S, P, O);
    const double t_bfs0 = now_s();
    int rc = CORAL_OK;
    if (S.n_reads > 0)
        for (int32_t ai = 0; ai < n_seed && rc == CORAL_OK; ++ai) rc = run.enqueue(ai);
    int64_t ccid = 0;
    for (int32_t ai = 0; ai < n_seed && rc == CORAL_OK; ++ai)
        if (O.iv[5 * (size_t)ai + 3] == -1) {
            rc = run.bfs_from(ai, ccid);
            ++ccid;
        }
    if (S.profile) fprintf(stderr, "coral_search_bfs: %.2f ms in all, %.2f ms of it inside coral_search_step (waits for steps included), %zu support triples kept\n",
                           (now_s() - t_bfs0) * 1e3, run.t_steps * 1e3, O.chunk_read.size());
    if (rc == BFS_ERR_KEY_CHROM) snprintf(S.err, sizeof(S.err), "search_bfs: KeyError contig %lld has no hashed alignments", (long long)O.err_tid);
    else if (rc == BFS_ERR_KEY_CHRIDX) snprintf(S.err, sizeof(S.err), "search_bfs: KeyError contig %lld outside chr1..22,X,Y,M", (long long)O.err_tid);
    else if (rc == BFS_ERR_INDEX) snprintf(S.err, sizeof(S.err), "search_bfs: segment index out of range");
    // ---- flatten
    O.bp_flat.clear(); O.bp_meta.clear(); O.bp_stats.clear(); O.chunk_off.clear();
    for (const BfsBp &b : O.bps) {
        O.bp_flat.insert(O.bp_flat.end(), b.f, b.f + 11);
        const int64_t m[4] = {b.flags, b.ccid, (int64_t)(O.chunk_off.size() / 2), (int64_t)(O.chunk_off.size() / 2 + b.chunks.size())};
        O.bp_meta.insert(O.bp_meta.end(), m, m + 4);
        O.bp_stats.insert(O.bp_stats.end(), b.stats, b.stats + 6);
        for (const auto &c : b.chunks) { O.chunk_off.push_back(c[0]); O.chunk_off.push_back(c[1]); }
    }
    O.conn_flat.clear(); O.conn_off.assign(1, 0); O.conn_vals.clear();
    for (size_t k = 0; k < O.conn_key.size(); ++k) {
        O.conn_flat.push_back(O.conn_key[k][0]);
        O.conn_flat.push_back(O.conn_key[k][1]);
        O.conn_vals.insert(O.conn_vals.end(), O.conn_val[k].begin(), O.conn_val[k].end());
        O.conn_off.push_back((int64_t)O.conn_vals.size());
    }
    return rc;
}

// Arrays of the last coral_search_bfs (owned by the handle until the next one / coral_search_free); which =
//   0 intervals int64[n][5] (contig, start, end, ccid, start-is-a-bool)   1 breakpoints int64[n][11] (c1 p1 o1 c2 p2 o2 head name id, i, j, gap, swapped)
//   2 per breakpoint int64[n][4] (flags, ccid, first chunk, end chunk)    3 statistics double[n][6]
//   4 chunks int64[n][2] (begin, end into 5 / 6 / 7)    5 / 6 / 7 support triples: read name id, i, j
//   8 connection keys int64[n][2], insertion order    9 offsets int64[n + 1] into 10    10 breakpoint indices added per key, in order
//   11 events int64[n][6] (for the log)
extern "C" int coral_search_bfs_get(void *h, int32_t which, const void **ptr, int64_t *n) {
    if (!h || !ptr || !n) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    if (!S.bfs) return CORAL_ERR_ARG;
    BfsOut &O = *S.bfs;
    auto give = [&](const auto &v) { *ptr = v.data(); *n = (int64_t)v.size(); return CORAL_OK; };
    switch (which) {
        case 0: return give(O.iv);
        case 1: return give(O.bp_flat);
        case 2: return give(O.bp_meta);
        case 3: return give(O.bp_stats);
        case 4: return give(O.chunk_off);
        case 5: return give(O.chunk_read);
        case 6: return give(O.chunk_i);
        case 7: return give(O.chunk_j);
        case 8: return give(O.conn_flat);
        case 9: return give(O.conn_off);
        case 10: return give(O.conn_vals);
        case 11: return give(O.events);
        default: return CORAL_ERR_ARG;
    }
}
