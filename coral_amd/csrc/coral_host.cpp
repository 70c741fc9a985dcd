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
