// coral_host.cpp — host-side native pieces of libcoral_hip.so (no device code).
//
//   coral_cluster_first_fit   exact restatement of the greedy first-fit breakpoint clustering of
//                             /root/reference/src/breakpoint_utilities.py:268-282 with a per-cluster
//                             set of *distinct* coordinates, so a cluster of N near-identical candidates
//                             costs O(N * distinct) instead of O(N^2).
#include <stdint.h>
#include <stdlib.h>

#include <unordered_set>
#include <vector>

#include "../../include/coral_hip.h"

namespace {
struct Cluster {
    std::vector<int64_t> a, b;                  // distinct (p1, p2) members
    std::unordered_set<uint64_t> seen;
    int64_t amin, amax, bmin, bmax;
};
inline uint64_t mix(int64_t a, int64_t b) {
    uint64_t x = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b + 0x7F4A7C15ull + ((uint64_t)a << 6));
    return x;
}
}  // namespace

extern "C" int coral_cluster_first_fit(int64_t n, const int64_t *p1, const int64_t *p2, int64_t cutoff,
                                       int32_t *cluster_of, int32_t *n_clusters) {
    if (n < 0 || (n > 0 && (!p1 || !p2 || !cluster_of)) || !n_clusters) return CORAL_ERR_ARG;
    std::vector<Cluster> cl;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t x = p1[i], y = p2[i];
        int home = -1;
        for (size_t c = 0; c < cl.size() && home < 0; ++c) {
            Cluster &k = cl[c];
            if (x - k.amax >= cutoff || k.amin - x >= cutoff || y - k.bmax >= cutoff || k.bmin - y >= cutoff) continue;
            const size_t m = k.a.size();
            for (size_t j = 0; j < m; ++j) {
                if (llabs(x - k.a[j]) < cutoff && llabs(y - k.b[j]) < cutoff) {
                    home = (int)c;
                    break;
                }
            }
        }
        if (home < 0) {
            cl.emplace_back();
            home = (int)cl.size() - 1;
            Cluster &k = cl[home];
            k.amin = k.amax = x;
            k.bmin = k.bmax = y;
        }
        Cluster &k = cl[home];
        // identical coordinates add no information to later membership tests: keep each pair once.
        // (exact 128-bit identity is checked on hash hits, so a hash collision can never drop a pair)
        const uint64_t h = mix(x, y);
        bool dup = false;
        if (k.seen.count(h)) {
            for (size_t j = 0; j < k.a.size(); ++j)
                if (k.a[j] == x && k.b[j] == y) {
                    dup = true;
                    break;
                }
        }
        if (!dup) {
            k.seen.insert(h);
            k.a.push_back(x);
            k.b.push_back(y);
            if (x < k.amin) k.amin = x;
            if (x > k.amax) k.amax = x;
            if (y < k.bmin) k.bmin = y;
            if (y > k.bmax) k.bmax = y;
        }
        cluster_of[i] = home;
    }
    *n_clusters = (int32_t)cl.size();
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
