// coral_host.cpp — host-side native pieces of libcoral_hip.so (no device code).
//
//   coral_cluster_first_fit   exact restatement of the greedy first-fit breakpoint clustering of
//                             /root/reference/src/breakpoint_utilities.py:268-282 with a per-cluster
//                             set of *distinct* coordinates, so a cluster of N near-identical candidates
//                             costs O(N * distinct) instead of O(N^2).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <exception>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/coral_hip.h"
#include "coral_names.h"
#include "pyset_emu.h"

namespace {
struct Member {
    int64_t a, b;
    int32_t cluster;      // lowest-numbered cluster that contains a candidate with exactly these coordinates
};
inline uint64_t cell_key(int64_t cx, int64_t cy) {
    return (uint64_t)cx * 0x9E3779B97F4A7C15ull ^ ((uint64_t)cy + 0x7F4A7C15F39CC060ull + ((uint64_t)cx << 6) + ((uint64_t)cx >> 2));
}
struct Cell {
    int64_t cx, cy;
    std::vector<Member> m;
};
}  // namespace

// "Join the FIRST existing cluster that has any member within `cutoff` at both ends" == the lowest cluster ordinal
// among all earlier candidates within range (clusters are numbered in creation order).  A uniform grid of cell size
// `cutoff` over (p1, p2) bounds the search to the 3 x 3 neighbouring cells, and only distinct coordinates are kept per
// cell (with the lowest cluster they belong to), so N near-identical candidates cost O(N * distinct) and N scattered
// singletons cost O(N), instead of the O(N^2) scan of the reference.  Exact: every comparison is on the coordinates.
extern "C" int coral_cluster_first_fit(int64_t n, const int64_t *p1, const int64_t *p2, int64_t cutoff,
                                       int32_t *cluster_of, int32_t *n_clusters) {
    if (n < 0 || cutoff <= 0 || (n > 0 && (!p1 || !p2 || !cluster_of)) || !n_clusters) return CORAL_ERR_ARG;
    std::unordered_map<uint64_t, std::vector<Cell>> grid;       // hash -> cells (collisions resolved by exact cx, cy)
    grid.reserve((size_t)n * 2 + 16);
    auto floor_div = [](int64_t v, int64_t d) { return v >= 0 ? v / d : -((-v + d - 1) / d); };
    int32_t ncl = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t x = p1[i], y = p2[i];
        const int64_t cx = floor_div(x, cutoff), cy = floor_div(y, cutoff);
        int32_t best = INT32_MAX;
        // own cell first: most candidates repeat coordinates of cluster 0, and nothing can beat cluster 0
        static const int order9[9][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
        for (int q = 0; q < 9 && best != 0; ++q) {
            const int64_t dx = order9[q][0], dy = order9[q][1];
            auto it = grid.find(cell_key(cx + dx, cy + dy));
            if (it == grid.end()) continue;
            for (const Cell &c : it->second) {
                if (c.cx != cx + dx || c.cy != cy + dy) continue;
                for (const Member &m : c.m)
                    if (m.cluster < best && llabs(x - m.a) < cutoff && llabs(y - m.b) < cutoff) {
                        best = m.cluster;
                        if (best == 0) break;
                    }
            }
        }
        if (best == INT32_MAX) best = ncl++;
        cluster_of[i] = best;
        std::vector<Cell> &bucket = grid[cell_key(cx, cy)];
        Cell *cell = nullptr;
        for (Cell &c : bucket)
            if (c.cx == cx && c.cy == cy) {
                cell = &c;
                break;
            }
        if (!cell) {
            bucket.push_back(Cell{cx, cy, {}});
            cell = &bucket.back();
        }
        bool found = false;
        for (Member &m : cell->m)
            if (m.a == x && m.b == y) {
                if (best < m.cluster) m.cluster = best;
                found = true;
                break;
            }
        if (!found) cell->m.push_back(Member{x, y, best});
    }
    *n_clusters = ncl;
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// coral_first_seen_rows — mark the first occurrence of every distinct row of an int64 matrix (row-major),
// in input order.  Exact: rows are compared column by column on every hash hit (open addressing, linear probing).
// Restates the per-read "if sa not in list: append" de-duplication of
// /root/reference/src/infer_breakpoint_graph.py:146-151 for all reads at once (the read name id is column 0).
// ---------------------------------------------------------------------------------------------
extern "C" int coral_first_seen_rows(int64_t n, int32_t ncols, const int64_t *rows, uint8_t *is_first) {
    if (n < 0 || ncols <= 0 || (n > 0 && (!rows || !is_first))) return CORAL_ERR_ARG;
    size_t cap = 16;
    while (cap < (size_t)n * 2) cap <<= 1;
    std::vector<int64_t> table(cap, -1);
    const size_t mask = cap - 1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t *r = rows + (size_t)i * (size_t)ncols;
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int c = 0; c < ncols; ++c) {
            h ^= (uint64_t)r[c] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
        }
        size_t slot = (size_t)(h ^ (h >> 31)) & mask;
        bool first = true;
        for (;;) {
            const int64_t j = table[slot];
            if (j < 0) {
                table[slot] = i;
                break;
            }
            const int64_t *q = rows + (size_t)j * (size_t)ncols;
            bool same = true;
            for (int c = 0; c < ncols && same; ++c) same = q[c] == r[c];
            if (same) {
                first = false;
                break;
            }
            slot = (slot + 1) & mask;
        }
        is_first[i] = first ? 1 : 0;
    }
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// coral_pyset_* — iteration order of CPython `set` objects, computed without creating them.
//
// The reference iterates Python sets of read-name strings inside its interval search
// (/root/reference/src/infer_breakpoint_graph.py:379-384 builds one set per reached CN segment with .add(),
// :405-419 unions them with |= into one set per candidate interval, :428/432 iterates that set), and the order of that
// iteration decides the order of the emitted breakpoints.  To produce the same order at array speed, these functions
// replay CPython 3.10's set algorithm (Objects/setobject.c: set_add_entry / set_insert_clean / set_table_resize /
// set_merge; LINEAR_PROBES 9, PERTURB_SHIFT 5, resize at fill*5 >= mask*3 to used*4 (used*2 above 50000), the |=
// pre-resize to (used + other.used)*2 and its empty-target fast paths) on (item id, hash) pairs.  Item identity stands
// for string equality.  tests/test_pyset_order.py checks the replay against real sets of str on random inputs.
// ---------------------------------------------------------------------------------------------
namespace {
using coral_detail::PySetEmu;
struct PySetBatch {
    std::vector<PySetEmu> sets;
    virtual ~PySetBatch() {}
};
}  // namespace

// One emulated set per key, filled with .add() in entry order (== set([first]) then .add(...), ibg:379-384).
extern "C" void *coral_pyset_batch_create(int64_t n_entries, const int32_t *key_of_entry, const int32_t *item,
                                          const int64_t *item_hash, int32_t n_keys, int32_t *out_count) {
    if (n_entries < 0 || n_keys < 0 || (n_entries > 0 && (!key_of_entry || !item || !item_hash))) return nullptr;
    PySetBatch *b = new PySetBatch();
    b->sets.resize((size_t)n_keys);
    for (int64_t i = 0; i < n_entries; ++i) {
        const int32_t k = key_of_entry[i];
        if (k < 0 || k >= n_keys) { delete b; return nullptr; }
        b->sets[(size_t)k].add(item[i], item_hash[item[i]]);
    }
    if (out_count)
        for (int32_t k = 0; k < n_keys; ++k) out_count[k] = (int32_t)b->sets[(size_t)k].used;
    return b;
}

// result = set(); for k in keys: result |= sets[k]; return list(result)      (ibg:405-419 then the iteration at :432)
extern "C" int coral_pyset_union_order(void *handle, int32_t n_union, const int32_t *keys, int32_t *out_items, int32_t *out_n) {
    if (!handle || n_union < 0 || (n_union > 0 && !keys) || !out_n) return CORAL_ERR_ARG;
    PySetBatch *b = (PySetBatch *)handle;
    PySetEmu acc;
    for (int32_t j = 0; j < n_union; ++j) {
        if (keys[j] < 0 || (size_t)keys[j] >= b->sets.size()) return CORAL_ERR_ARG;
        acc.merge(b->sets[(size_t)keys[j]]);
    }
    int32_t n = 0;
    for (size_t e = 0; e <= acc.mask; ++e)
        if (acc.key[e] >= 0) {
            if (out_items) out_items[n] = acc.key[e];
            ++n;
        }
    *out_n = n;
    return CORAL_OK;
}

extern "C" int coral_pyset_batch_free(void *handle) {
    delete (PySetBatch *)handle;
    return CORAL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Reachable CN segments of one amplicon interval (ibg:369-384): walk the reads hashed to segments [si, ei] of `tid` in
// visiting order, and add every read to the emulated set of each (chromosome, segment) it touches outside the interval.
// Keys are numbered in order of first appearance, which is the insertion order of the reference's nested dicts.
namespace {
struct ReachSets : PySetBatch {
    std::vector<int64_t> codes;      // tid << 32 | cni per key
};
}  // namespace

extern "C" void *coral_reach_create(int64_t n_visit, const int64_t *visit_rows, const int64_t *row_read, const int64_t *off,
                                    const int64_t *row_tid, const int64_t *cni0, const int64_t *cni1, int64_t n_reads,
                                    int64_t tid, int64_t si, int64_t ei, const int64_t *read_hash, int32_t *n_keys_out) {
    if (n_visit < 0 || n_reads < 0 || !n_keys_out || (n_visit > 0 && (!visit_rows || !row_read || !off || !row_tid || !cni0 ||
                                                                       !cni1 || !read_hash)))
        return nullptr;
    ReachSets *b = new ReachSets();
    std::vector<uint8_t> seen((size_t)n_reads, 0);
    std::unordered_map<int64_t, int32_t> key_of;
    auto emit = [&](int64_t t, int64_t c, int64_t r) {
        const int64_t code = (t << 32) | c;
        auto it = key_of.find(code);
        int32_t k;
        if (it == key_of.end()) {
            k = (int32_t)b->sets.size();
            key_of.emplace(code, k);
            b->sets.emplace_back();
            b->codes.push_back(code);
        } else {
            k = it->second;
        }
        b->sets[(size_t)k].add((int32_t)r, read_hash[r]);
    };
    for (int64_t v = 0; v < n_visit; ++v) {
        const int64_t r = row_read[visit_rows[v]];
        if (r < 0 || r >= n_reads) { delete b; return nullptr; }
        if (seen[(size_t)r]) continue;
        seen[(size_t)r] = 1;
        for (int64_t k = off[r]; k < off[r + 1]; ++k) {
            const int64_t t = row_tid[k], c0 = cni0[k], c1 = cni1[k];
            const bool other = t != tid;
            if (c0 >= 0 && (other || c0 <= si || c0 >= ei)) emit(t, c0, r);               // Q9: boundary segments count as outside
            if (c1 >= 0 && c1 != c0 && (other || c1 <= si || c1 >= ei)) emit(t, c1, r);
        }
    }
    *n_keys_out = (int32_t)b->sets.size();
    return static_cast<PySetBatch *>(b);
}

extern "C" int coral_reach_keys(void *handle, int64_t *codes, int32_t *counts) {
    if (!handle || !codes || !counts) return CORAL_ERR_ARG;
    ReachSets *b = static_cast<ReachSets *>((PySetBatch *)handle);
    for (size_t k = 0; k < b->sets.size(); ++k) {
        codes[k] = b->codes[k];
        counts[k] = (int32_t)b->sets[k].used;
    }
    return CORAL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// NM statistics of the mapped, non-chimeric MAPQ-60 records (ibg:153-157): count, sum of e and sum of e*e with
// e = NM / query_length, accumulated in file order with one rounding per addition — the float results the reference gets
// from its sequential `+=` (build with -ffp-contract=off).  One pass over the host mirrors of the records.
extern "C" int coral_nm_stats(int64_t n, const int32_t *tid, const int64_t *sa_off, const int32_t *mapq, const int32_t *nm,
                              const int32_t *qlen, int64_t *count, double *sum_e, double *sum_e2) {
    if (n < 0 || !count || !sum_e || !sum_e2 || (n > 0 && (!tid || !sa_off || !mapq || !nm || !qlen))) return CORAL_ERR_ARG;
    int64_t c = 0;
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (tid[i] < 0 || sa_off[i + 1] > sa_off[i] || mapq[i] != 60) continue;
        if (qlen[i] == 0) return CORAL_ERR_ZERODIV;      // record without SEQ: NM / query_length raises in the reference (ibg:154)
        const double e = (double)nm[i] / (double)qlen[i];
        const double e2 = e * e;
        s0 += e;
        s1 += e2;
        ++c;
    }
    *count = c;
    *sum_e = s0;
    *sum_e2 = s1;
    return CORAL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Read support of the concordant edges (ibg:1043-1055).  For edge q the reference fetches the reads covering four positions
// (p, p + 1, p - 101, p + 101), intersects the four NAME sets and removes the names that support a discordant edge at either
// node.  Here the four fetches arrive as record ordinals (coral_point_cover), names are ids, and a per-name stamp replaces
// the sets: bit d of mark[name] = "covers position d of the current edge".
extern "C" int coral_concordant_counts(int32_t n_edges, const int64_t *pt_begin, const int64_t *pt_end, const int32_t *pt_rec,
                                       int64_t n_pt_rec, const int32_t *rec_name, int64_t n_rec, int64_t n_names, const int64_t *sup_off,
                                       const int64_t *sup_name, int64_t *count) {
    if (n_edges < 0 || n_names < 0 || !count || (n_edges > 0 && (!pt_begin || !pt_end || !sup_off))) return CORAL_ERR_ARG;
    if (n_edges == 0) return CORAL_OK;
    for (int64_t p = 0; p < 4 * (int64_t)n_edges; ++p)
        if (pt_begin[p] < 0 || pt_end[p] < pt_begin[p] || pt_end[p] > n_pt_rec) return CORAL_ERR_ARG;
    if ((n_pt_rec > 0 && (!pt_rec || !rec_name)) || (sup_off[n_edges] > 0 && !sup_name)) return CORAL_ERR_ARG;
    // one word per read name: epoch << 6 | bits.  The array lives on between calls (per thread) and the epoch keeps counting, so a
    // build does not pay for clearing 8 MB at 2 M reads; it is cleared when it grows or when the epoch would wrap
    static thread_local std::vector<uint32_t> mark;
    static thread_local uint32_t epoch = 0;
    if (mark.size() < (size_t)n_names || epoch + (uint32_t)n_edges + 1u >= (1u << 26)) {
        mark.assign((size_t)n_names, 0u);
        epoch = 0;
    }
    for (int32_t q = 0; q < n_edges; ++q) {
        const uint32_t tag = (++epoch) << 6;
        for (int d = 0; d < 4; ++d)
            for (int64_t k = pt_begin[4 * q + d]; k < pt_end[4 * q + d]; ++k) {
                const int64_t rec = pt_rec[k];
                if (rec < 0 || rec >= n_rec) return CORAL_ERR_ARG;
                const int32_t nm = rec_name[rec];
                if (nm < 0 || nm >= n_names) return CORAL_ERR_ARG;
                uint32_t &m = mark[(size_t)nm];
                if ((m & ~63u) != tag) m = tag;
                m |= 1u << d;
            }
        for (int64_t k = sup_off[q]; k < sup_off[q + 1]; ++k) {
            const int64_t nm = sup_name[k];
            if (nm < 0 || nm >= n_names) return CORAL_ERR_ARG;
            uint32_t &m = mark[(size_t)nm];
            if ((m & ~63u) != tag) m = tag;
            m |= 16u;                                          // supports a discordant edge at one of the two nodes
        }
        int64_t c = 0;
        for (int64_t k = pt_begin[4 * q]; k < pt_end[4 * q]; ++k) {
            uint32_t &m = mark[(size_t)rec_name[pt_rec[k]]];
            if ((m & 63u) == 15u) {                            // all four positions, no discordant support, not counted yet
                ++c;
                m |= 32u;
            }
        }
        count[q] = c;
    }
    return CORAL_OK;
}


// Read-name tables of consecutive byte ranges of one file -> one table numbered by first appearance over the file
// (coral_names.h: unify).  What rank 0 does with the gathered per-rank decodes; replaces a Python dict over 2 M str.
extern "C" int coral_names_unify(int32_t n_pieces, const int64_t *n_names, const uint8_t *const *blob, const int64_t *const *off,
                                 int32_t *const *lut, uint8_t *out_blob, int64_t *out_off, int64_t *n_global, int32_t n_threads) {
    if (n_pieces < 0 || !n_global || !out_off || (n_pieces > 0 && (!n_names || !blob || !off || !lut))) return CORAL_ERR_ARG;
    int64_t total = 0;
    for (int32_t p = 0; p < n_pieces; ++p) {
        if (n_names[p] < 0 || (n_names[p] > 0 && (!blob[p] || !off[p] || !lut[p]))) return CORAL_ERR_ARG;
        total += n_names[p];
    }
    if (total > 0x7FFFFFFF) return CORAL_ERR_ARG;             // name ids are int32 throughout
    if (total > 0 && !out_blob) return CORAL_ERR_ARG;
    try {
        *n_global = coral_names::unify(n_pieces, n_names, blob, off, lut, out_blob, out_off, n_threads);
    } catch (const std::exception &) {
        return CORAL_ERR_ARG;
    }
    return CORAL_OK;
}


// ---------------------------------------------------------------------------------------------
// coral_cn_solve — the CN assignment of one amplicon graph: argmin Σ w_inv/x + w_lin·x − w_log·log x  s.t.  A x = 0, x > 0,
// started at x = 1 (the convex program compute_cn_lr hands to cvxopt.solvers.cp, /root/reference/src/breakpoint_graph.py:495-606).
// Infeasible-start Newton on the KKT system with a backtracking search on the KKT residual — the iteration of
// coral_amd/breakpoint_graph.py:solve_cn_lr statement by statement (same formulas, same stopping rules), on the sparse balance
// matrix (an edge touches at most two nodes: a column of A has at most two non-zeros), so that an iteration costs tens of
// microseconds instead of a dozen numpy round trips.  A must have linearly independent rows (the caller drops the others).
// Returns 0 = done (x filled), 1 = the reduced Newton system was singular at some iteration: the caller runs its general path.
// ---------------------------------------------------------------------------------------------
extern "C" int coral_cn_solve(int32_t n, int32_t p, const double *w_inv, const double *w_lin, const double *w_log, const double *A,
                              int32_t max_iter, double *x_out, double *nu_out, int32_t *n_iter) {
    if (n <= 0 || p < 0 || !w_inv || !w_lin || !w_log || (p > 0 && (!A || !nu_out)) || !x_out) return CORAL_ERR_ARG;
    // columns of A as (row, value) lists
    std::vector<int32_t> col_off((size_t)n + 1, 0);
    std::vector<int32_t> col_row;
    std::vector<double> col_val;
    for (int32_t j = 0; j < n; ++j) {
        for (int32_t i = 0; i < p; ++i)
            if (A[(size_t)i * n + j] != 0.0) { col_row.push_back(i); col_val.push_back(A[(size_t)i * n + j]); }
        col_off[(size_t)j + 1] = (int32_t)col_row.size();
    }
    std::vector<double> x((size_t)n, 1.0), nu((size_t)p, 0.0), r((size_t)n + p), r_new((size_t)n + p), h((size_t)n), dx((size_t)n), dnu((size_t)p),
        xt((size_t)n), nut((size_t)p), hinv((size_t)n);
    auto residual = [&](const std::vector<double> &xv, const std::vector<double> &nv, std::vector<double> &out) {
        for (int32_t i = 0; i < p; ++i) out[(size_t)n + i] = 0.0;
        for (int32_t j = 0; j < n; ++j) {
            double atnu = 0.0;
            for (int32_t q = col_off[(size_t)j]; q < col_off[(size_t)j + 1]; ++q) {
                atnu += col_val[(size_t)q] * nv[(size_t)col_row[(size_t)q]];
                out[(size_t)n + col_row[(size_t)q]] += col_val[(size_t)q] * xv[(size_t)j];
            }
            out[(size_t)j] = w_lin[j] - w_log[j] / xv[(size_t)j] - w_inv[j] / (xv[(size_t)j] * xv[(size_t)j]) + atnu;
        }
    };
    auto norm = [&](const std::vector<double> &v) {
        double s = 0.0;
        for (double e : v) s += e * e;
        return sqrt(s);
    };
    std::vector<int32_t> zero;
    std::vector<double> M, rhs;
    residual(x, nu, r);
    int32_t it = 0;
    for (; it < max_iter; ++it) {
        zero.clear();
        for (int32_t j = 0; j < n; ++j) {
            h[(size_t)j] = w_log[j] / (x[(size_t)j] * x[(size_t)j]) + 2.0 * w_inv[j] / (x[(size_t)j] * x[(size_t)j] * x[(size_t)j]);
            if (h[(size_t)j] > 0) hinv[(size_t)j] = 1.0 / h[(size_t)j];
            else { hinv[(size_t)j] = 0.0; zero.push_back(j); }
        }
        const int32_t nz = (int32_t)zero.size(), m = p + nz;
        // M = [[S, -Az], [Azᵀ, 0]],  S = Σ_j hinv_j a_j a_jᵀ over the columns with h > 0;  rhs = [rp − Ap (hinv ∘ rd_p); −rd_z]
        M.assign((size_t)m * m, 0.0);
        rhs.assign((size_t)m, 0.0);
        for (int32_t i = 0; i < p; ++i) rhs[(size_t)i] = r[(size_t)n + i];
        for (int32_t j = 0; j < n; ++j) {
            if (!(h[(size_t)j] > 0)) continue;
            const double w = hinv[(size_t)j];
            for (int32_t q = col_off[(size_t)j]; q < col_off[(size_t)j + 1]; ++q) {
                const int32_t a = col_row[(size_t)q];
                rhs[(size_t)a] -= col_val[(size_t)q] * (w * r[(size_t)j]);
                for (int32_t q2 = col_off[(size_t)j]; q2 < col_off[(size_t)j + 1]; ++q2)
                    M[(size_t)a * m + col_row[(size_t)q2]] += (col_val[(size_t)q] * w) * col_val[(size_t)q2];
            }
        }
        for (int32_t k = 0; k < nz; ++k) {
            const int32_t j = zero[(size_t)k];
            for (int32_t q = col_off[(size_t)j]; q < col_off[(size_t)j + 1]; ++q) {
                M[(size_t)col_row[(size_t)q] * m + p + k] = -col_val[(size_t)q];
                M[(size_t)(p + k) * m + col_row[(size_t)q]] = col_val[(size_t)q];
            }
            rhs[(size_t)p + k] = -r[(size_t)j];
        }
        // Gaussian elimination with partial pivoting
        bool singular = false;
        for (int32_t c = 0; c < m && !singular; ++c) {
            int32_t piv = c;
            double best = fabs(M[(size_t)c * m + c]);
            for (int32_t i = c + 1; i < m; ++i)
                if (fabs(M[(size_t)i * m + c]) > best) { best = fabs(M[(size_t)i * m + c]); piv = i; }
            if (!(best > 0.0) || !std::isfinite(best)) { singular = true; break; }
            if (piv != c) {
                for (int32_t k = c; k < m; ++k) std::swap(M[(size_t)c * m + k], M[(size_t)piv * m + k]);
                std::swap(rhs[(size_t)c], rhs[(size_t)piv]);
            }
            const double d = M[(size_t)c * m + c];
            for (int32_t i = c + 1; i < m; ++i) {
                const double f = M[(size_t)i * m + c] / d;
                if (f == 0.0) continue;
                double *ri = &M[(size_t)i * m], *rc = &M[(size_t)c * m];
                for (int32_t k = c + 1; k < m; ++k) ri[k] -= f * rc[k];
                rhs[(size_t)i] -= f * rhs[(size_t)c];
            }
        }
        if (singular) return 1;
        for (int32_t c = m - 1; c >= 0; --c) {
            double s = rhs[(size_t)c];
            for (int32_t k = c + 1; k < m; ++k) s -= M[(size_t)c * m + k] * rhs[(size_t)k];
            rhs[(size_t)c] = s / M[(size_t)c * m + c];
            if (!std::isfinite(rhs[(size_t)c])) return 1;
        }
        for (int32_t i = 0; i < p; ++i) dnu[(size_t)i] = rhs[(size_t)i];
        for (int32_t j = 0; j < n; ++j) {
            if (!(h[(size_t)j] > 0)) continue;
            double atdnu = 0.0;
            for (int32_t q = col_off[(size_t)j]; q < col_off[(size_t)j + 1]; ++q) atdnu += col_val[(size_t)q] * dnu[(size_t)col_row[(size_t)q]];
            dx[(size_t)j] = -(r[(size_t)j] + atdnu) * hinv[(size_t)j];
        }
        for (int32_t k = 0; k < nz; ++k) dx[(size_t)zero[(size_t)k]] = rhs[(size_t)p + k];
        double t = 1.0;
        bool any = false;
        double lim = 0.0;
        for (int32_t j = 0; j < n; ++j)
            if (dx[(size_t)j] < 0) {
                const double v = -x[(size_t)j] / dx[(size_t)j];
                if (!any || v < lim) lim = v;
                any = true;
            }
        if (any) t = std::min(1.0, 0.99 * lim);
        const double r0 = norm(r);
        bool improved = false;
        while (t > 1e-10) {
            for (int32_t j = 0; j < n; ++j) xt[(size_t)j] = x[(size_t)j] + t * dx[(size_t)j];
            for (int32_t i = 0; i < p; ++i) nut[(size_t)i] = nu[(size_t)i] + t * dnu[(size_t)i];
            residual(xt, nut, r_new);
            if (norm(r_new) <= (1.0 - 0.01 * t) * r0) { improved = true; break; }
            t *= 0.5;
        }
        double rel = 0.0;
        for (int32_t j = 0; j < n; ++j) rel = std::max(rel, fabs(dx[(size_t)j]) / x[(size_t)j]);
        const bool small = rel < 1e-13;
        if (!improved) {
            if (small) for (int32_t j = 0; j < n; ++j) x[(size_t)j] += dx[(size_t)j];      // at the float64 floor: the (tiny) full step, then stop
            break;
        }
        x.swap(xt);
        nu.swap(nut);
        r.swap(r_new);
        if (small) break;
    }
    for (int32_t j = 0; j < n; ++j) x_out[j] = x[(size_t)j];
    for (int32_t i = 0; i < p; ++i) nu_out[i] = nu[(size_t)i];
    if (n_iter) *n_iter = it;
    return CORAL_OK;
}


// coral_independent_rows — a maximal linearly independent subset of the rows of A (double[m][n], entries 0 / ±1: the balance rows of
// the CN program, bg:526-541): rows are taken in order and kept when they are not in the span of the rows kept so far (Gram-Schmidt
// against an orthonormal basis of that span, re-orthogonalised once).  keep[m] receives 1 / 0; returns the number kept.
extern "C" int coral_independent_rows(int32_t m, int32_t n, const double *A, uint8_t *keep, double tol) {
    if (m < 0 || n <= 0 || (m > 0 && (!A || !keep))) return CORAL_ERR_ARG;
    std::vector<double> Q;                       // kept rows, orthonormalised, row-major
    std::vector<double> v((size_t)n), c;
    int32_t k = 0;
    const int32_t cap = m < n ? m : n;
    for (int32_t i = 0; i < m; ++i) {
        keep[i] = 0;
        if (k == cap) continue;
        double norm2 = 0.0;
        for (int32_t j = 0; j < n; ++j) { v[(size_t)j] = A[(size_t)i * n + j]; norm2 += v[(size_t)j] * v[(size_t)j]; }
        const double norm = sqrt(norm2);
        if (norm == 0.0) continue;
        for (int pass = 0; pass < 2 && k > 0; ++pass) {
            c.assign((size_t)k, 0.0);
            for (int32_t q = 0; q < k; ++q) {
                double d = 0.0;
                const double *qr = &Q[(size_t)q * n];
                for (int32_t j = 0; j < n; ++j) d += qr[j] * v[(size_t)j];
                c[(size_t)q] = d;
            }
            for (int32_t q = 0; q < k; ++q) {
                const double *qr = &Q[(size_t)q * n];
                const double d = c[(size_t)q];
                for (int32_t j = 0; j < n; ++j) v[(size_t)j] -= d * qr[j];
            }
        }
        double res2 = 0.0;
        for (int32_t j = 0; j < n; ++j) res2 += v[(size_t)j] * v[(size_t)j];
        const double res = sqrt(res2);
        if (res > tol * (norm > 1.0 ? norm : 1.0) && res > 1e-7 * norm) {
            Q.resize((size_t)(k + 1) * n);
            for (int32_t j = 0; j < n; ++j) Q[(size_t)k * n + j] = v[(size_t)j] / res;
            keep[i] = 1;
            ++k;
        }
    }
    return k;
}
