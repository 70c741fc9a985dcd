// coral_search.cpp — host side of the amplicon-interval search (no device code).
//
// One step of the breadth-first search of find_interval_i (/root/reference/src/infer_breakpoint_graph.py:343-673) has a part
// that is a pure function of the interval's coordinates (ibg:362-434):
//   * the CN segments reached from the interval through chimeric reads, each with the SET of read names that reach it
//     (ibg:369-384) — sets of str whose iteration order later decides the order of the breakpoints;
//   * segments with fewer reads than min_cluster_cutoff dropped (ibg:385-391), the rest grouped into runs of neighbouring
//     segments whose sets are united with |= (ibg:392-419);
//   * for every run, alignment2bp of every read of the united set against (run, interval) (ibg:428-434, bu:70-96).
// coral_search_step does all of it in one call on index arrays: the sets are replayed with pyset_emu.h (no Python object is
// created), and the candidates are FILTERED out of the pair table the GPU built once per graph build (k_bp_pairs in
// coral_kernels.hip) — which pairs of a read's alignments fall into the two intervals is four comparisons per pair.
// coral_search_within is the same filter for alignment2bp_l over all chimeric reads (find_breakpoints, ibg:676-690).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "../../include/coral_hip.h"
#include "pyset_emu.h"

namespace {
using coral_detail::PySetEmu;

struct Search {
    int64_t n_reads = 0, n_rows = 0, n_ent = 0;
    const int64_t *off = nullptr, *row_read = nullptr, *row_tid = nullptr, *ra = nullptr, *rb = nullptr, *cni0 = nullptr,
                  *cni1 = nullptr, *read_hash = nullptr, *read_name = nullptr, *e_key = nullptr, *e_row = nullptr;
    const int32_t *pairs = nullptr;                       // [2 * n_rows][8]  (k_bp_pairs)
    int32_t n_tid = 0;
    const int64_t *seg_off = nullptr, *seg_start = nullptr, *seg_end = nullptr;      // CN segments per contig, file order; end inclusive
    std::vector<uint32_t> seen;                           // per read: stamp of the step that expanded it
    uint32_t stamp = 0;
    // results of the last call
    std::vector<int64_t> groups;                          // [n_groups][4]: contig id, first segment, last segment, candidates
    std::vector<int64_t> cand;                            // [K][13]: c1 p1 o1 c2 p2 o2 read(name id) i j gap swapped mqa mqb
    std::vector<int32_t> order;                           // reads (table index) of every run in set-iteration order
    std::vector<int64_t> order_off;                       // [n_groups + 1]
    std::vector<char> used;                               // scratch: per-read "pair k gave a candidate" flags
    char err[256] = "";
    // CORAL_SEARCH_PROFILE=1: seconds per phase of coral_search_step and work counters, printed when the handle is freed
    bool profile = false;
    double t_reach = 0, t_plan = 0, t_union = 0, t_cand = 0;
    long long n_steps = 0, n_visit = 0, n_adds = 0, n_keys = 0, n_union_items = 0;
};
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline bool row_in(const Search &S, int64_t row, int64_t t, int64_t s, int64_t e) {
    // interval_overlap(rint, [chr, s, e]) with rint = [chr, ra, rb]; for '-' rows ra > rb, i.e. "interval contains the whole
    // alignment" (bu:11-15)
    return S.row_tid[row] == t && S.ra[row] <= e && s <= S.rb[row];
}

inline bool emit(Search &S, int64_t slot, int64_t read) {
    const int32_t *p = S.pairs + 8 * slot;
    const int32_t bits = p[5];
    const int64_t base = S.off[read];
    const int64_t ia = p[6] - base, ib = p[7] - base;
    const bool swapped = (bits & 16) != 0;
    const int64_t row[13] = {p[0], p[1], (bits >> 2) & 1, p[2], p[3], (bits >> 3) & 1, S.read_name[read], swapped ? ib : ia,
                             swapped ? ia : ib, p[4], swapped ? 1 : 0, (bits >> 8) & 0xff, (bits >> 16) & 0xff};
    S.cand.insert(S.cand.end(), row, row + 13);
    return (bits & 128) == 0;                  // false: a contig outside chr1..22,X,Y,M reaches interval2bp (KeyError, bu:293)
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
    S->off = off; S->row_read = row_read; S->row_tid = row_tid; S->ra = ra; S->rb = rb; S->cni0 = cni0; S->cni1 = cni1;
    S->read_hash = read_hash; S->read_name = read_name; S->e_key = e_key; S->e_row = e_row; S->pairs = pairs;
    S->n_tid = n_tid; S->seg_off = seg_off; S->seg_start = seg_start; S->seg_end = seg_end;
    S->seen.assign((size_t)n_reads, 0u);
    const char *pe = getenv("CORAL_SEARCH_PROFILE");
    S->profile = pe && pe[0] == '1';
    return S;
}

extern "C" int coral_search_free(void *h) {
    if (h && ((Search *)h)->profile) {
        Search &S = *(Search *)h;
        fprintf(stderr, "coral_search: %lld steps  reach %.2f ms (visit rows %lld, set adds %lld, keys %lld)  plan %.2f ms  union %.2f ms (%lld items)  candidates %.2f ms\n",
                S.n_steps, S.t_reach * 1e3, S.n_visit, S.n_adds, S.n_keys, S.t_plan * 1e3, S.t_union * 1e3, S.n_union_items, S.t_cand * 1e3);
    }
    delete (Search *)h;
    return CORAL_OK;
}

extern "C" const char *coral_search_error(void *h) { return h ? ((Search *)h)->err : "null handle"; }

// Arrays of the last result (owned by the handle, valid until the next call on it).
extern "C" int coral_search_result(void *h, int64_t *n_groups, const int64_t **groups, int64_t *n_cand, const int64_t **cand,
                                   const int64_t **order_off, const int32_t **order) {
    if (!h || !n_groups || !groups || !n_cand || !cand) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    *n_groups = (int64_t)S.groups.size() / 4;
    *groups = S.groups.data();
    *n_cand = (int64_t)S.cand.size() / 13;
    *cand = S.cand.data();
    if (order_off) *order_off = S.order_off.data();
    if (order) *order = S.order.data();
    return CORAL_OK;
}

extern "C" int coral_search_step(void *h, int64_t tid, int64_t s, int64_t e, int64_t si, int64_t ei, double min_cluster_cutoff,
                                 int64_t max_seq_len) {
    if (!h) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    S.groups.clear(); S.cand.clear(); S.order.clear(); S.order_off.assign(1, 0);
    if (tid < 0 || tid >= S.n_tid) { snprintf(S.err, sizeof(S.err), "search_step: contig id out of range"); return CORAL_ERR_ARG; }
    // ---- reads hashed to segments si..ei of the contig, in the reference's visiting order (segment, then append order)
    const int64_t *lo = std::lower_bound(S.e_key, S.e_key + S.n_ent, (tid << 32) + si);
    const int64_t *hi = std::lower_bound(S.e_key, S.e_key + S.n_ent, (tid << 32) + ei + 1);
    if (lo == hi) return CORAL_OK;
    const double t0 = S.profile ? now_s() : 0.0;
    if (++S.stamp == 0) { std::fill(S.seen.begin(), S.seen.end(), 0u); S.stamp = 1; }
    std::vector<PySetEmu> sets;
    std::vector<int64_t> codes;                          // contig << 32 | segment, per key, in order of first appearance
    std::unordered_map<int64_t, int32_t> key_of;
    auto add = [&](int64_t t, int64_t c, int64_t r) {
        const int64_t code = (t << 32) | c;
        auto it = key_of.find(code);
        int32_t k;
        if (it == key_of.end()) {
            k = (int32_t)sets.size();
            key_of.emplace(code, k);
            sets.emplace_back();
            codes.push_back(code);
        } else {
            k = it->second;
        }
        sets[(size_t)k].add((int32_t)r, S.read_hash[r]);
        ++S.n_adds;
    };
    for (const int64_t *v = lo; v < hi; ++v) {
        const int64_t row = S.e_row[v - S.e_key];
        if (row < 0 || row >= S.n_rows) { snprintf(S.err, sizeof(S.err), "search_step: row out of range"); return CORAL_ERR_ARG; }
        const int64_t r = S.row_read[row];
        if (S.seen[(size_t)r] == S.stamp) continue;
        S.seen[(size_t)r] = S.stamp;
        for (int64_t k = S.off[r]; k < S.off[r + 1]; ++k) {
            const int64_t t = S.row_tid[k], c0 = S.cni0[k], c1 = S.cni1[k];
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
        if (!((double)sets[k].used < min_cluster_cutoff)) it->second.emplace_back(codes[k] & 0xFFFFFFFFLL, (int32_t)k);
    }
    struct Run { int64_t t, b0, b1; std::vector<int32_t> keys; };
    std::vector<Run> plan;
    for (int64_t t : contig_order) {
        auto &bins = bins_of[t];
        if (bins.empty()) continue;
        if (t < 0 || t >= S.n_tid) { snprintf(S.err, sizeof(S.err), "search_step: contig id out of range"); return CORAL_ERR_ARG; }
        std::sort(bins.begin(), bins.end());
        const int64_t *st = S.seg_start + S.seg_off[t], *en = S.seg_end + S.seg_off[t];
        const int64_t n_seg = S.seg_off[t + 1] - S.seg_off[t];
        Run cur{t, bins[0].first, bins[0].first, {}};
        for (size_t k = 0; k + 1 < bins.size(); ++k) {
            cur.keys.push_back(bins[k].second);
            const int64_t a = bins[k].first, b = bins[k + 1].first;
            if (a >= n_seg || b >= n_seg) { snprintf(S.err, sizeof(S.err), "search_step: segment index out of range"); return CORAL_ERR_ARG; }
            if (b - a > 2 || st[b] - en[a] > max_seq_len) {
                cur.b1 = a;
                plan.push_back(cur);
                cur = Run{t, b, b, {}};
            }
        }
        cur.keys.push_back(bins.back().second);
        cur.b1 = bins.back().first;
        if (cur.b1 >= n_seg) { snprintf(S.err, sizeof(S.err), "search_step: segment index out of range"); return CORAL_ERR_ARG; }
        plan.push_back(cur);
    }
    // ---- per run: iteration order of  set() | sets[k0] | sets[k1] | ...  then alignment2bp of every read (bu:70-96)
    bool contigs_ok = true;
    const double t2 = S.profile ? now_s() : 0.0;
    double t_u = 0.0;
    for (const Run &run : plan) {
        const double tu0 = S.profile ? now_s() : 0.0;
        PySetEmu acc;
        for (int32_t k : run.keys) acc.merge(sets[(size_t)k]);
        if (S.profile) { t_u += now_s() - tu0; S.n_union_items += (long long)acc.used; }
        const int64_t t1 = run.t, s1 = S.seg_start[S.seg_off[run.t] + run.b0], e1 = S.seg_end[S.seg_off[run.t] + run.b1];
        const size_t cand_before = S.cand.size();
        for (size_t slot_e = 0; slot_e <= acc.mask; ++slot_e) {
            const int32_t r = acc.key[slot_e];
            if (r < 0) continue;
            S.order.push_back(r);
            const int64_t base = S.off[r], n = S.off[r + 1] - base;
            if (n < 2) continue;
            S.used.assign((size_t)n, 0);
            for (int64_t k = 0; k + 1 < n; ++k) {                     // pairs (k, k + 1)
                const int64_t a = base + k, b = a + 1;
                if (!(S.pairs[8 * (2 * a) + 5] & 2)) continue;
                if ((row_in(S, a, t1, s1, e1) && row_in(S, b, tid, s, e)) || (row_in(S, b, t1, s1, e1) && row_in(S, a, tid, s, e))) {
                    S.used[(size_t)k] = 1;
                    contigs_ok &= emit(S, 2 * a, r);
                }
            }
            for (int64_t k = 1; k + 1 < n; ++k) {                     // pairs (k - 1, k + 1) around a low-MAPQ alignment
                if (S.used[(size_t)k - 1] || S.used[(size_t)k]) continue;
                const int64_t m = base + k, a = m - 1, b = m + 1;
                if (!(S.pairs[8 * (2 * m + 1) + 5] & 2)) continue;
                if ((row_in(S, a, t1, s1, e1) && row_in(S, b, tid, s, e)) || (row_in(S, b, t1, s1, e1) && row_in(S, a, tid, s, e)))
                    contigs_ok &= emit(S, 2 * m + 1, r);
            }
        }
        S.order_off.push_back((int64_t)S.order.size());
        const int64_t g[4] = {run.t, run.b0, run.b1, (int64_t)((S.cand.size() - cand_before) / 13)};
        S.groups.insert(S.groups.end(), g, g + 4);
    }
    if (S.profile) {
        const double t3 = now_s();
        S.t_reach += t1 - t0; S.t_plan += t2 - t1; S.t_union += t_u; S.t_cand += (t3 - t2) - t_u;
        ++S.n_steps; S.n_visit += hi - lo; S.n_keys += (long long)codes.size();
    }
    if (!contigs_ok) { snprintf(S.err, sizeof(S.err), "search_step: contig outside chr1..22,X,Y,M"); return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}

// alignment2bp_l (bu:129-186) of every chimeric read, in table (= dict) order, against the interval list: both alignments of
// a pair must have the SAME first overlapping interval (interval_overlap_l, bu:37-44) and either change strand or fail the
// collinearity test.  Result: one group with all candidates (coral_search_result).
extern "C" int coral_search_within(void *h, int32_t n_int, const int64_t *int_tid, const int64_t *int_start, const int64_t *int_end) {
    if (!h || n_int < 0 || (n_int > 0 && (!int_tid || !int_start || !int_end))) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    S.groups.clear(); S.cand.clear(); S.order.clear(); S.order_off.assign(1, 0);
    std::vector<std::vector<int32_t>> by_tid((size_t)S.n_tid);              // interval indices per contig, list order kept
    for (int32_t k = 0; k < n_int; ++k)
        if (int_tid[k] >= 0 && int_tid[k] < S.n_tid) by_tid[(size_t)int_tid[k]].push_back(k);
    auto first_interval = [&](int64_t row) -> int32_t {
        const int64_t t = S.row_tid[row];
        if (t < 0 || t >= S.n_tid) return -1;
        for (int32_t k : by_tid[(size_t)t])
            if (S.ra[row] <= int_end[k] && int_start[k] <= S.rb[row]) return k;
        return -1;
    };
    bool contigs_ok = true;
    std::vector<int32_t> fi;
    for (int64_t r = 0; r < S.n_reads; ++r) {
        const int64_t base = S.off[r], n = S.off[r + 1] - base;
        if (n < 2) continue;
        fi.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) fi[(size_t)k] = first_interval(base + k);
        S.used.assign((size_t)n, 0);
        for (int64_t k = 0; k + 1 < n; ++k) {
            const int32_t bits = S.pairs[8 * (2 * (base + k)) + 5];
            if (!(bits & 2) || fi[(size_t)k] < 0 || fi[(size_t)k] != fi[(size_t)k + 1]) continue;
            if ((bits & 32) || (bits & 64)) {
                S.used[(size_t)k] = 1;
                contigs_ok &= emit(S, 2 * (base + k), r);
            }
        }
        for (int64_t k = 1; k + 1 < n; ++k) {
            if (S.used[(size_t)k - 1] || S.used[(size_t)k]) continue;
            const int32_t bits = S.pairs[8 * (2 * (base + k) + 1) + 5];
            if (!(bits & 2) || fi[(size_t)k - 1] < 0 || fi[(size_t)k - 1] != fi[(size_t)k + 1]) continue;
            if ((bits & 32) || (bits & 64)) contigs_ok &= emit(S, 2 * (base + k) + 1, r);
        }
    }
    const int64_t g[4] = {-1, -1, -1, (int64_t)(S.cand.size() / 13)};
    S.groups.insert(S.groups.end(), g, g + 4);
    S.order_off.push_back(0);
    if (!contigs_ok) { snprintf(S.err, sizeof(S.err), "search_within: contig outside chr1..22,X,Y,M"); return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}

// alignment2bp (bu:70-96) of the given reads between two intervals — the single query coral_search_step runs per run;
// exported for the unit tests against the reference's own vectors and for callers that keep the reference's loop.
extern "C" int coral_search_between(void *h, int64_t n_sel, const int32_t *reads, int64_t t1, int64_t s1, int64_t e1, int64_t t2,
                                    int64_t s2, int64_t e2) {
    if (!h || n_sel < 0 || (n_sel > 0 && !reads)) return CORAL_ERR_ARG;
    Search &S = *(Search *)h;
    S.groups.clear(); S.cand.clear(); S.order.clear(); S.order_off.assign(1, 0);
    bool contigs_ok = true;
    for (int64_t q = 0; q < n_sel; ++q) {
        const int64_t r = reads[q];
        if (r < 0 || r >= S.n_reads) { snprintf(S.err, sizeof(S.err), "search_between: read index out of range"); return CORAL_ERR_ARG; }
        const int64_t base = S.off[r], n = S.off[r + 1] - base;
        if (n < 2) continue;
        S.used.assign((size_t)n, 0);
        for (int64_t k = 0; k + 1 < n; ++k) {
            const int64_t a = base + k, b = a + 1;
            if (!(S.pairs[8 * (2 * a) + 5] & 2)) continue;
            if ((row_in(S, a, t1, s1, e1) && row_in(S, b, t2, s2, e2)) || (row_in(S, b, t1, s1, e1) && row_in(S, a, t2, s2, e2))) {
                S.used[(size_t)k] = 1;
                contigs_ok &= emit(S, 2 * a, r);
            }
        }
        for (int64_t k = 1; k + 1 < n; ++k) {
            if (S.used[(size_t)k - 1] || S.used[(size_t)k]) continue;
            const int64_t m = base + k, a = m - 1, b = m + 1;
            if (!(S.pairs[8 * (2 * m + 1) + 5] & 2)) continue;
            if ((row_in(S, a, t1, s1, e1) && row_in(S, b, t2, s2, e2)) || (row_in(S, b, t1, s1, e1) && row_in(S, a, t2, s2, e2)))
                contigs_ok &= emit(S, 2 * m + 1, r);
        }
    }
    const int64_t g[4] = {t1, -1, -1, (int64_t)(S.cand.size() / 13)};
    S.groups.insert(S.groups.end(), g, g + 4);
    if (!contigs_ok) { snprintf(S.err, sizeof(S.err), "search_between: contig outside chr1..22,X,Y,M"); return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}
