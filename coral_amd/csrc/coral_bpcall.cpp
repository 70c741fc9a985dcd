// coral_bpcall.cpp — breakpoint candidates -> clusters -> exact breakpoints, in one native call (host code, no device code).
//
// Restates, for arrays of candidates, the chain the reference runs per pair of intervals:
//   cluster_bp_list   /root/reference/src/breakpoint_utilities.py:252-286   (groups by chromosome pair + orientations in
//                                                                             first-seen order, greedy first-fit inside a group)
//   bpc2bp            /root/reference/src/breakpoint_utilities.py:299-388   (3-sigma trim, mode / median consensus, support)
//   bp_match          /root/reference/src/breakpoint_utilities.py:391-416
//   the sub-cluster loop of /root/reference/src/infer_breakpoint_graph.py:436-457, :693-718, :777-802
// Every float operation is performed in the reference's order on IEEE doubles (build with -ffp-contract=off); the sums the
// reference keeps in Python integers are kept in 128-bit integers and converted once, as int -> float conversion does.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <array>
#include <unordered_map>
#include <vector>

#include "../../include/coral_hip.h"

namespace {
enum Field { C1, P1, O1, C2, P2, O2, READ, SI, SJ, GAP, SWAPPED, MQA, MQB, N_FIELDS };

struct View {
    const int64_t *p[N_FIELDS];
    int64_t s[N_FIELDS];
    int64_t at(int f, int64_t i) const { return p[f][i * s[f]]; }
};

typedef __int128 wide_t;

// math.sqrt(sq_mean - mean * mean); a negative argument is the reference's ValueError branch.
inline bool sigma(double sq_mean, double mean, double *out) {
    const double x = sq_mean - mean * mean;
    if (x < 0 || x != x) return false;
    *out = sqrt(x);
    return true;
}

// Unique mode, else the median rounded toward the junction side of the last cluster member (bu:336-357, Appendix A Q6).
// Values of one cluster lie within a few hundred bp of each other, so they are counted in a small table; a wide spread
// falls back to sorting.  Both give the same mode / tie / median.
int64_t consensus(std::vector<int64_t> &v, bool last_is_plus, std::vector<int32_t> &count) {
    const size_t n = v.size();
    int64_t lo = v[0], hi = v[0];
    for (int64_t x : v) {
        lo = std::min(lo, x);
        hi = std::max(hi, x);
    }
    auto round_median = [&](int64_t a, int64_t b) {
        const double med = ((double)a + (double)b) / 2.0;
        return (int64_t)(last_is_plus ? ceil(med) : floor(med));
    };
    if (hi - lo < 8192) {
        count.assign((size_t)(hi - lo + 1), 0);
        for (int64_t x : v) ++count[(size_t)(x - lo)];
        int32_t top = 0;
        size_t n_top = 0, n_unique = 0, mode = 0;
        for (size_t k = 0; k < count.size(); ++k) {
            if (!count[k]) continue;
            ++n_unique;
            if (count[k] > top) {
                top = count[k];
                n_top = 1;
                mode = k;
            } else if (count[k] == top) {
                ++n_top;
            }
        }
        if (n_unique == 1 || n_top == 1) return lo + (int64_t)mode;
        // order statistics n/2 - 1 and n/2 (0-based) from the counts
        const size_t want_b = n / 2, want_a = n % 2 == 1 ? n / 2 : n / 2 - 1;
        int64_t a = 0, b = 0;
        size_t seen = 0;
        bool have_a = false;
        for (size_t k = 0; k < count.size(); ++k) {
            seen += (size_t)count[k];
            if (!have_a && seen > want_a) {
                a = lo + (int64_t)k;
                have_a = true;
            }
            if (seen > want_b) {
                b = lo + (int64_t)k;
                break;
            }
        }
        return n % 2 == 1 ? b : round_median(a, b);
    }
    std::sort(v.begin(), v.end());
    size_t top = 0, n_top = 0, n_unique = 0;
    int64_t mode = v[0];
    for (size_t a = 0; a < n;) {
        size_t b = a;
        while (b < n && v[b] == v[a]) ++b;
        ++n_unique;
        if (b - a > top) {
            top = b - a;
            n_top = 1;
            mode = v[a];
        } else if (b - a == top) {
            ++n_top;
        }
        a = b;
    }
    if (n_unique == 1 || n_top == 1) return mode;
    if (n % 2 == 1) return v[n / 2];
    return round_median(v[n / 2 - 1], v[n / 2]);
}

// Number of distinct (read, i, j) triples among the supporting candidates (len(set(tuples)), ibg:443 / :699 / :783).
struct TripleSet {
    std::vector<std::array<int64_t, 3>> slot;
    std::vector<uint8_t> used;
    size_t distinct(const View &c, const std::vector<int64_t> &sup) {
        size_t cap = 16;
        while (cap < sup.size() * 2) cap <<= 1;
        slot.resize(cap);
        used.assign(cap, 0);
        size_t n = 0;
        for (int64_t i : sup) {
            const std::array<int64_t, 3> t = {c.at(READ, i), c.at(SI, i), c.at(SJ, i)};
            uint64_t h = (uint64_t)t[0] * 0x9E3779B97F4A7C15ull;
            h ^= ((uint64_t)t[1] + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
            h ^= ((uint64_t)t[2] + 0x94D049BBull) * 0x94D049BB133111EBull;
            size_t k = (size_t)(h ^ (h >> 29)) & (cap - 1);
            for (;;) {
                if (!used[k]) {
                    used[k] = 1;
                    slot[k] = t;
                    ++n;
                    break;
                }
                if (slot[k] == t) break;
                k = (k + 1) & (cap - 1);
            }
        }
        return n;
    }
};
}  // namespace

extern "C" int coral_call_breakpoints(int64_t n, const int64_t *const *field_ptr, const int64_t *field_stride,
                                      double min_cluster_cutoff, int64_t bp_distance_cutoff, int64_t match_cutoff,
                                      double accept_floor, int32_t advance_subcluster, int32_t *n_clusters,
                                      int32_t *cluster_size, int32_t *n_calls, int64_t *call_head, int64_t *call_p1,
                                      int64_t *call_p2, double *call_stats, int32_t *call_flags, int64_t *call_sup_off,
                                      int64_t *sup_idx) {
    if (n < 0 || !field_ptr || !field_stride || !n_clusters || !n_calls || bp_distance_cutoff <= 0) return CORAL_ERR_ARG;
    *n_clusters = 0;
    *n_calls = 0;
    if (n == 0) return CORAL_OK;
    if (!cluster_size || !call_head || !call_p1 || !call_p2 || !call_stats || !call_flags || !call_sup_off || !sup_idx)
        return CORAL_ERR_ARG;
    View c;
    for (int f = 0; f < N_FIELDS; ++f) {
        if (!field_ptr[f] || field_stride[f] <= 0) return CORAL_ERR_ARG;
        c.p[f] = field_ptr[f];
        c.s[f] = field_stride[f];
    }
    // ---- cluster_bp_list: groups in first-seen order, members in input order
    std::unordered_map<int64_t, int32_t> group_of;
    std::vector<std::vector<int64_t>> groups;
    for (int64_t i = 0; i < n; ++i) {
        // (chr1, chr2, o1, o2) as one key: 31 bits per contig id (BAM refIDs are int32), no assumption about the header's size
        const int64_t key = (((c.at(C1, i) & 0x7fffffffLL) << 33) | ((c.at(C2, i) & 0x7fffffffLL) << 2) | ((c.at(O1, i) & 1) << 1) |
                             (c.at(O2, i) & 1));
        auto it = group_of.find(key);
        if (it == group_of.end()) {
            group_of.emplace(key, (int32_t)groups.size());
            groups.emplace_back();
            groups.back().push_back(i);
        } else {
            groups[(size_t)it->second].push_back(i);
        }
    }
    std::vector<std::vector<int64_t>> clusters;
    std::vector<int64_t> a, b;
    std::vector<int32_t> cl;
    for (auto &g : groups) {
        if ((double)g.size() < min_cluster_cutoff) {           // small groups pass through as one cluster (bu:285)
            clusters.push_back(g);
            continue;
        }
        a.resize(g.size());
        b.resize(g.size());
        cl.resize(g.size());
        for (size_t k = 0; k < g.size(); ++k) {
            a[k] = c.at(P1, g[k]);
            b[k] = c.at(P2, g[k]);
        }
        int32_t ncl = 0;
        const int rc = coral_cluster_first_fit((int64_t)g.size(), a.data(), b.data(), bp_distance_cutoff, cl.data(), &ncl);
        if (rc != CORAL_OK) return rc;
        const size_t base = clusters.size();
        clusters.resize(base + (size_t)ncl);
        for (size_t k = 0; k < g.size(); ++k) clusters[base + (size_t)cl[k]].push_back(g[k]);
    }
    // ---- exact breakpoints per cluster
    const double cut = (double)match_cutoff;
    const double sd_floor = cut / 2.99;
    int32_t calls = 0;
    int64_t n_sup_total = 0;
    call_sup_off[0] = 0;
    std::vector<int64_t> rest, next, vals1, vals2, sup;
    std::vector<uint8_t> keep, ok;
    std::vector<int32_t> count;
    TripleSet triples;
    for (auto &cluster : clusters) {
        cluster_size[(*n_clusters)++] = (int32_t)cluster.size();
        if ((double)cluster.size() < min_cluster_cutoff) continue;
        int64_t sub = 0;
        rest = cluster;
        while ((double)rest.size() >= min_cluster_cutoff && !rest.empty()) {
            const int64_t head = rest[0];
            const int64_t o1 = c.at(O1, head), o2 = c.at(O2, head);
            const double nn = (double)rest.size();
            wide_t s1 = 0, s11 = 0, s2 = 0, s22 = 0;
            for (int64_t i : rest) {
                const wide_t x = c.at(P1, i), y = c.at(P2, i);
                s1 += x; s11 += x * x; s2 += y; s22 += y * y;
            }
            const double mu1 = (double)s1 / nn, mu2 = (double)s2 / nn;
            double sd1, sd2;
            if (!sigma((double)s11 / nn, mu1, &sd1)) sd1 = sd_floor; else if (!(sd1 > sd_floor)) sd1 = sd_floor;
            if (!sigma((double)s22 / nn, mu2, &sd2)) sd2 = sd_floor; else if (!(sd2 > sd_floor)) sd2 = sd_floor;
            const double hi1 = mu1 + 3.0 * sd1, lo1 = mu1 - 3.0 * sd1, hi2 = mu2 + 3.0 * sd2, lo2 = mu2 - 3.0 * sd2;
            vals1.clear();
            vals2.clear();
            for (int64_t i : rest) {
                const double x = (double)c.at(P1, i), y = (double)c.at(P2, i);
                if (x <= hi1 && x >= lo1 && y <= hi2 && y >= lo2) {
                    vals1.push_back(c.at(P1, i));
                    vals2.push_back(c.at(P2, i));
                }
            }
            int64_t bp1 = o1 == 0 ? 0 : 1000000000, bp2 = o2 == 0 ? 0 : 1000000000;
            if (!vals1.empty()) {
                const int64_t last = rest.back();
                bp1 = consensus(vals1, c.at(O1, last) == 0, count);
                bp2 = consensus(vals2, c.at(O2, last) == 0, count);
            }
            // bp_match of every member against (bp1, bp2)
            sup.clear();
            next.clear();
            for (int64_t i : rest) {
                const int64_t x = c.at(P1, i), y = c.at(P2, i);
                const bool d1 = llabs(x - bp1) < match_cutoff, d2 = llabs(y - bp2) < match_cutoff;
                const double rgap = (double)c.at(GAP, i) * 1.2;
                bool match;
                if (rgap <= 0) {
                    match = d1 && d2;
                } else {
                    double left = rgap;
                    bool u1, u2;
                    if (o1 == 0) {
                        u1 = x <= bp1 - match_cutoff;
                        if (u1) left = left - (double)(bp1 - match_cutoff - x + 1);
                    } else {
                        u1 = x >= bp1 + match_cutoff;
                        if (u1) left = left - (double)(x - bp1 - match_cutoff + 1);
                    }
                    if (o2 == 0) {
                        u2 = y <= bp2 - match_cutoff;
                        if (u2) left = left - (double)(bp2 - match_cutoff - y + 1);
                    } else {
                        u2 = y >= bp2 + match_cutoff;
                        if (u2) left = left - (double)(y - bp2 - match_cutoff + 1);
                    }
                    match = ((u1 && left >= 0) || d1) && ((u2 && left >= 0) || d2);
                }
                (match ? sup : next).push_back(i);
            }
            if (sup.empty()) break;                            // no support: nothing to report, nothing left (bu:374-376)
            const int64_t n_distinct = (int64_t)triples.distinct(c, sup);
            if ((sub == 0 && (double)n_distinct >= min_cluster_cutoff) || (double)n_distinct >= accept_floor) {
                const double k = (double)sup.size();
                wide_t a1 = 0, a11 = 0, a2 = 0, a22 = 0;
                int64_t m4 = 0, m5 = 0;
                for (int64_t i : sup) {
                    const wide_t x = c.at(P1, i), y = c.at(P2, i);
                    a1 += x; a11 += x * x; a2 += y; a22 += y * y;
                    const bool sw = c.at(SWAPPED, i) != 0;
                    m4 += sw ? c.at(MQB, i) : c.at(MQA, i);
                    m5 += sw ? c.at(MQA, i) : c.at(MQB, i);
                }
                double *st = call_stats + 6 * (size_t)calls;
                st[0] = (double)a1 / k;
                st[1] = (double)a2 / k;
                st[2] = (double)a11 / k;
                st[3] = (double)a22 / k;
                st[4] = (double)m4 / k;
                st[5] = (double)m5 / k;
                int32_t flags = 0;
                double sg;
                if (sigma(st[2], st[0], &sg)) st[2] = sg; else { st[2] = 0; flags |= 1; }      // flag: the reference stores int 0
                if (sigma(st[3], st[1], &sg)) st[3] = sg; else { st[3] = 0; flags |= 2; }
                call_flags[calls] = flags;
                call_head[calls] = head;
                call_p1[calls] = bp1;
                call_p2[calls] = bp2;
                for (int64_t i : sup) sup_idx[n_sup_total++] = i;
                call_sup_off[++calls] = n_sup_total;
            }
            rest.swap(next);
            if (advance_subcluster) ++sub;
        }
    }
    *n_calls = calls;
    return CORAL_OK;
}
