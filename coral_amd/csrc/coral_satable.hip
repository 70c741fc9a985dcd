// coral_satable.hip — K3: the chimeric-alignment table of ALL reads from the tokenised SA tags, on the GPU.
//
// Replaces the per-record loop of fetch() (/root/reference/src/infer_breakpoint_graph.py:139-174: first-seen
// de-duplication of SA entries per read name, read_length from the first record with flag < 256, reads without a
// primary dropped) and alignment_from_satags + the nine cigar2pos* functions
// (/root/reference/src/cigar_parsing.py:17-269: query interval, reference interval, stable (qs, qe) sort).
//
// Pipeline (rocPRIM/hipCUB primitives for the radix sorts and prefix sums, hand-written kernels for the rest):
//   rows keyed by read-name id -> stable radix sort -> one thread per read: de-duplicate, parse, insertion-sort its
//   (few) rows -> reads ordered by their first SA-bearing record (the reference's dict order) -> compaction.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>

#include "../../include/coral_hip.h"

#define INVALID_KEY 0x7fffffff

static thread_local char g_sa_err[256] = "";
extern "C" const char *coral_sa_last_error(void) { return g_sa_err; }
static int sa_err(int code, const char *msg) {
    snprintf(g_sa_err, sizeof(g_sa_err), "%s", msg);
    return code;
}

__global__ void k_first_primary(int n_rec, const int32_t *__restrict__ tid, const int32_t *__restrict__ flagmq,
                                const int32_t *__restrict__ name, int32_t *__restrict__ first_primary) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rec && tid[r] >= 0 && (flagmq[r] & 0xFFFF) < 256) atomicMin(&first_primary[name[r]], r);
}

__global__ void k_row_keys(int n_sa, const int32_t *__restrict__ sa_rec, const int32_t *__restrict__ tid,
                           const int32_t *__restrict__ name, int32_t *__restrict__ keys, int32_t *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sa) return;
    const int r = sa_rec[i];
    keys[i] = tid[r] >= 0 ? name[r] : INVALID_KEY;        // whole-file fetch() skips unplaced records
    vals[i] = i;
}

__global__ void k_heads(int n, const int32_t *__restrict__ keys, int32_t *__restrict__ head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void k_group_starts(int n, const int32_t *__restrict__ head, const int32_t *__restrict__ gid_incl,
                               int32_t *__restrict__ gstart) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) gstart[gid_incl[i] - 1] = i;
    if (i == n - 1) gstart[gid_incl[i]] = n;
}

struct ParsedRow {
    int qs, qe, tid, ra, rb, strand, mapq, nm;
};

// cigar2pos* for the nine shapes (cp:17-215), from the tokenised [c5 S] m M [x I | -x D] [c3 S]
__device__ __forceinline__ void parse_row(const int32_t *__restrict__ f, int nm, int rl, ParsedRow &o) {
    const int tid = f[0], pos1 = f[1], strand = f[2], c5 = f[3], m = f[4], x = f[5], c3 = f[6];
    const bool fwd = strand == 0, has5 = c5 > 0, has3 = c3 > 0, ins = x > 0, del = x < 0;
    const int al = m + (del ? -x : 0);
    int qs, qe;
    if (has5 && has3) {
        if (!ins && !del) { qs = fwd ? c5 : c3; qe = qs + al - 1; }
        else if (fwd) { qs = c5; qe = rl - c3 - 1; }
        else { qs = c3; qe = rl - c5 - 1; }
    } else if (has5) {
        if (fwd) { qs = c5; qe = rl - 1; }
        else { qs = 0; qe = ins ? rl - c5 - 1 : (del ? m - 1 : al - 1); }
    } else {
        if (!fwd) { qs = c3; qe = rl - 1; }
        else { qs = 0; qe = ins ? rl - c3 - 1 : (del ? m - 1 : al - 1); }
    }
    o.qs = qs; o.qe = qe; o.tid = tid; o.strand = strand; o.mapq = f[7]; o.nm = nm;
    o.ra = fwd ? pos1 - 1 : pos1 + al - 2;
    o.rb = fwd ? pos1 + al - 2 : pos1 - 1;
}

__global__ void k_group_process(int n_groups, const int32_t *__restrict__ gstart, const int32_t *__restrict__ keys,
                                const int32_t *__restrict__ vals, const int32_t *__restrict__ sa,
                                const int32_t *__restrict__ sa_nm, const int32_t *__restrict__ first_primary,
                                const int32_t *__restrict__ rec_qlen, int32_t *__restrict__ kept_ws /* [n_sa] */,
                                int32_t *__restrict__ tmp_rows /* [n_sa][8] */, int32_t *__restrict__ n_kept,
                                int32_t *__restrict__ g_first, int32_t *__restrict__ g_failed, int32_t *__restrict__ err) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int a = gstart[g], b = gstart[g + 1];
    const int name = keys[a];
    n_kept[g] = 0;
    g_failed[g] = 0;
    g_first[g] = INVALID_KEY;
    if (name == INVALID_KEY) return;
    const int fp = first_primary[name];
    if (fp == INVALID_KEY) return;                       // chimeric read without a primary alignment: dropped (ibg:163-173)
    g_first[g] = vals[a];                                // stable sort: the read's first SA row in file order
    const int rl = rec_qlen[fp];
    // The group owns slots a..b of kept_ws and tmp_rows (it has b - a SA rows), so a read may have any number of
    // alignments: nothing is kept in per-thread arrays.
    // first-seen de-duplication (ibg:146-151), then cp:246-255: the read fails as a whole at its first offending entry
    int32_t *__restrict__ kept_idx = kept_ws + a;
    int nk = 0;
    for (int i = a; i < b; ++i) {
        const int32_t *fi = sa + 8ll * vals[i];
        const int nmi = sa_nm[vals[i]];
        bool dup = false;
        for (int j = 0; j < nk && !dup; ++j) {
            const int32_t *fj = sa + 8ll * kept_idx[j];
            bool same = sa_nm[kept_idx[j]] == nmi;
            for (int c = 0; c < 8 && same; ++c) same = fi[c] == fj[c];
            dup = same;
        }
        if (dup) continue;
        kept_idx[nk++] = vals[i];
    }
    for (int j = 0; j < nk; ++j) {
        const int32_t *f = sa + 8ll * kept_idx[j];
        if (f[3] == -2) { atomicMax(err, 3); return; }                       // unknown shape: KeyError in the reference
        if ((f[3] <= 0 && f[6] <= 0) || f[4] <= 0) { g_failed[g] = 1; return; }   // no S or no M: ([], [], [])
    }
    // parse + stable insertion sort by (qs, qe) (cp:263), in place in the group's slots of tmp_rows
    ParsedRow *__restrict__ rows = reinterpret_cast<ParsedRow *>(tmp_rows + 8ll * a);
    for (int j = 0; j < nk; ++j) {
        ParsedRow p;
        parse_row(sa + 8ll * kept_idx[j], sa_nm[kept_idx[j]], rl, p);
        if (p.qe == p.qs) atomicMax(err, 4);                                  // ZeroDivisionError at cp:268
        int k = j;
        while (k > 0) {
            const ParsedRow q = rows[k - 1];
            if (!(q.qs > p.qs || (q.qs == p.qs && q.qe > p.qe))) break;
            rows[k] = q;
            --k;
        }
        rows[k] = p;
    }
    n_kept[g] = nk;
}

__global__ void k_order_counts(int n_groups, const int32_t *__restrict__ gf_sorted, const int32_t *__restrict__ g_sorted,
                               const int32_t *__restrict__ n_kept, int32_t *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups) cnt[i] = gf_sorted[i] != INVALID_KEY ? n_kept[g_sorted[i]] : 0;
}

__global__ void k_scatter(int n_reads, const int32_t *__restrict__ g_sorted, const int32_t *__restrict__ off,
                          const int32_t *__restrict__ gstart, const int32_t *__restrict__ keys,
                          const int32_t *__restrict__ n_kept, const int32_t *__restrict__ g_failed,
                          const int32_t *__restrict__ tmp_rows, int32_t *__restrict__ out_rows,
                          int32_t *__restrict__ out_name, int32_t *__restrict__ out_failed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const int g = g_sorted[i];
    out_name[i] = keys[gstart[g]];
    out_failed[i] = g_failed[g];
    const int nk = n_kept[g];
    const int32_t *src = tmp_rows + 8ll * gstart[g];
    int32_t *dst = out_rows + 8ll * off[i];
    for (int j = 0; j < 8 * nk; ++j) dst[j] = src[j];
}

__global__ void k_read_length(int n_names, int32_t *__restrict__ fp, const int32_t *__restrict__ qlen, int32_t *__restrict__ rl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_names) return;
    const int v = fp[i];
    if (v == 0x7f7f7f7f) { fp[i] = INVALID_KEY; rl[i] = -1; } else rl[i] = qlen[v];
}

__global__ void k_iota(int n, int32_t *__restrict__ v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

__global__ void k_count_valid(int n, const int32_t *__restrict__ gf, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && gf[i] != INVALID_KEY && (i == n - 1 || gf[i + 1] == INVALID_KEY)) out[0] = i + 1;
}

template <typename T>
static T *carve(char *&p, size_t n) {
    T *r = reinterpret_cast<T *>(p);
    p += (n * sizeof(T) + 255) / 256 * 256;
    return r;
}

extern "C" int coral_sa_table(int32_t n_rec, const int32_t *rec_tid, const int32_t *rec_flagmq, const int32_t *rec_qlen,
                              const int32_t *rec_name, int32_t n_names, int32_t n_sa, const int32_t *sa,
                              const int32_t *sa_nm, const int32_t *sa_rec, void *workspace, int64_t workspace_bytes,
                              int32_t *out_rows, int32_t *out_off, int32_t *out_name, int32_t *out_failed,
                              int32_t *out_read_length, int32_t *counts /* host [2]: n_reads, n_rows */, void *stream) {
    if (!counts) return sa_err(CORAL_ERR_ARG, "sa_table: counts is null");
    counts[0] = counts[1] = 0;
    if (n_rec < 0 || n_sa < 0 || n_names < 0) return sa_err(CORAL_ERR_ARG, "sa_table: negative size");
    hipStream_t s = (hipStream_t)stream;
    // ---- workspace layout (and its size)
    size_t cub_bytes = 0, b1 = 0, b2 = 0, b3 = 0;
    {
        int32_t *n32 = nullptr;
        hipcub::DeviceRadixSort::SortPairs(nullptr, b1, n32, n32, n32, n32, n_sa > 0 ? n_sa : 1, 0, 32, s);
        hipcub::DeviceScan::InclusiveSum(nullptr, b2, n32, n32, n_sa > 0 ? n_sa : 1, s);
        hipcub::DeviceScan::ExclusiveSum(nullptr, b3, n32, n32, n_sa > 0 ? n_sa + 1 : 1, s);
        cub_bytes = b1 > b2 ? b1 : b2;
        cub_bytes = cub_bytes > b3 ? cub_bytes : b3;
    }
    const size_t ns = (size_t)(n_sa > 0 ? n_sa : 1);
    size_t need = 0;
    {
        char *p = nullptr;
        carve<char>(p, cub_bytes);
        for (int k = 0; k < 13; ++k) carve<int32_t>(p, ns + 2);
        carve<int32_t>(p, 8 * ns);
        carve<int32_t>(p, (size_t)n_names + 1);
        need = (size_t)p;
    }
    if (!workspace || (size_t)workspace_bytes < need) {
        counts[0] = (int32_t)(need >> 20) + 1;          // MiB needed
        return sa_err(CORAL_ERR_CAPACITY, "sa_table: workspace too small");
    }
    if (out_read_length == nullptr) return sa_err(CORAL_ERR_ARG, "sa_table: null output");
    char *p = (char *)workspace;
    void *cub_tmp = carve<char>(p, cub_bytes);
    int32_t *keys = carve<int32_t>(p, ns + 2), *vals = carve<int32_t>(p, ns + 2), *keys_s = carve<int32_t>(p, ns + 2),
            *vals_s = carve<int32_t>(p, ns + 2), *head = carve<int32_t>(p, ns + 2), *gid = carve<int32_t>(p, ns + 2),
            *gstart = carve<int32_t>(p, ns + 2), *n_kept = carve<int32_t>(p, ns + 2), *g_first = carve<int32_t>(p, ns + 2),
            *g_failed = carve<int32_t>(p, ns + 2), *g_ids = carve<int32_t>(p, ns + 2), *scratch = carve<int32_t>(p, ns + 2),
            *kept_ws = carve<int32_t>(p, ns + 2);
    int32_t *tmp_rows = carve<int32_t>(p, 8 * ns);
    int32_t *first_primary = carve<int32_t>(p, (size_t)n_names + 1);
    const int B = 256;
    // ---- read_length = query_length of the first record with flag < 256 per name (ibg:142-143)
    (void)hipMemsetAsync(first_primary, 0x7f, sizeof(int32_t) * ((size_t)n_names + 1), s);      // 0x7f7f7f7f > any ordinal
    if (n_rec > 0) hipLaunchKernelGGL(k_first_primary, dim3((n_rec + B - 1) / B), dim3(B), 0, s, n_rec, rec_tid, rec_flagmq, rec_name, first_primary);
    // first_primary values of 0x7f7f7f7f mean "none": normalise to INVALID_KEY and produce read_length
    hipError_t e = hipSuccess;
    if (n_names > 0) hipLaunchKernelGGL(k_read_length, dim3((n_names + B - 1) / B), dim3(B), 0, s, n_names, first_primary, rec_qlen, out_read_length);
    if (n_sa == 0) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return sa_err(CORAL_ERR_HIP, hipGetErrorString(e));
        return CORAL_OK;
    }
    if (!out_rows || !out_off || !out_name || !out_failed) return sa_err(CORAL_ERR_ARG, "sa_table: null output");
    const int G = (n_sa + B - 1) / B;
    int32_t *err = scratch;            // scratch[0] = error code, scratch[1..] reused below
    (void)hipMemsetAsync(err, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(k_row_keys, dim3(G), dim3(B), 0, s, n_sa, sa_rec, rec_tid, rec_name, keys, vals);
    size_t tb = cub_bytes;
    hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, keys, keys_s, vals, vals_s, n_sa, 0, 32, s);
    hipLaunchKernelGGL(k_heads, dim3(G), dim3(B), 0, s, n_sa, keys_s, head);
    tb = cub_bytes;
    hipcub::DeviceScan::InclusiveSum(cub_tmp, tb, head, gid, n_sa, s);
    hipLaunchKernelGGL(k_group_starts, dim3(G), dim3(B), 0, s, n_sa, head, gid, gstart);
    int32_t n_groups = 0;
    e = hipMemcpyAsync(&n_groups, gid + (n_sa - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return sa_err(CORAL_ERR_HIP, hipGetErrorString(e));
    const int GG = (n_groups + B - 1) / B;
    hipLaunchKernelGGL(k_group_process, dim3(GG), dim3(B), 0, s, n_groups, gstart, keys_s, vals_s, sa, sa_nm, first_primary,
                       rec_qlen, kept_ws, tmp_rows, n_kept, g_first, g_failed, err);
    // ---- reads in the reference's dict order = by first SA-bearing record = by their first SA row
    hipLaunchKernelGGL(k_iota, dim3(GG), dim3(B), 0, s, n_groups, g_ids);
    int32_t *gf_sorted = keys, *g_sorted = vals;          // the unsorted row keys are no longer needed
    tb = cub_bytes;
    hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, g_first, gf_sorted, g_ids, g_sorted, n_groups, 0, 32, s);
    int32_t *cnt = head, *off = gid;                      // reuse
    hipLaunchKernelGGL(k_order_counts, dim3(GG), dim3(B), 0, s, n_groups, gf_sorted, g_sorted, n_kept, cnt);
    (void)hipMemsetAsync(cnt + n_groups, 0, sizeof(int32_t), s);
    tb = cub_bytes;
    hipcub::DeviceScan::ExclusiveSum(cub_tmp, tb, cnt, off, n_groups + 1, s);
    // number of reads = groups with a valid key (they sort first)
    (void)hipMemsetAsync(scratch + 1, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(k_count_valid, dim3(GG), dim3(B), 0, s, n_groups, gf_sorted, scratch + 1);
    int32_t h[2] = {0, 0};
    e = hipMemcpyAsync(h, scratch, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    int32_t n_rows = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_rows, off + n_groups, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return sa_err(CORAL_ERR_HIP, hipGetErrorString(e));
    if (h[0] == 3) return sa_err(CORAL_ERR_FORMAT, "sa_table: SA CIGAR shape outside SM/MS/SMS/SMD/MDS/SMDS/SMI/MIS/SMIS");
    if (h[0] == 4) return sa_err(CORAL_ERR_ZERODIV, "sa_table: zero-length query interval (ZeroDivisionError in the reference)");
    const int n_reads = h[1];
    if (n_reads > 0) {
        hipLaunchKernelGGL(k_scatter, dim3((n_reads + B - 1) / B), dim3(B), 0, s, n_reads, g_sorted, off, gstart, keys_s, n_kept,
                           g_failed, tmp_rows, out_rows, out_name, out_failed);
        e = hipMemcpyAsync(out_off, off, sizeof(int32_t) * ((size_t)n_reads + 1), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return sa_err(CORAL_ERR_HIP, hipGetErrorString(e));
    }
    counts[0] = n_reads;
    counts[1] = n_rows;
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// coral_hash_rows — hash_alignment_to_seg (/root/reference/src/infer_breakpoint_graph.py:181-210) on the SA table's
// device rows: the CN segment holding each end of every local alignment (the reference's two IntervalTree point queries
// per alignment) and the inverted index (contig, segment) -> alignments in the reference's append order.
//   k_hash_rows: one thread per table row, two binary searches over the (contig, start)-sorted disjoint segment table;
//   the up to two (segment, row) entries of a row get the key contig << 32 | segment, and ONE stable radix sort of the
//   keys (values = 2 * row + end, i.e. already in append order) yields every per-segment list in order.
// ---------------------------------------------------------------------------------------------
#define HASH_INVALID 0x7fffffffffffffffLL

__device__ __forceinline__ int seg_lookup(const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_start,
                                          const int32_t *__restrict__ seg_end, const int32_t *__restrict__ seg_idx, int n_seg,
                                          int t, int p) {
    int lo = 0, hi = n_seg;                 // last segment with (tid, start) <= (t, p)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int mt = seg_tid[mid];
        if (mt < t || (mt == t && seg_start[mid] <= p)) lo = mid + 1; else hi = mid;
    }
    const int k = lo - 1;
    return (k >= 0 && seg_tid[k] == t && p < seg_end[k]) ? seg_idx[k] : -1;
}

__global__ __launch_bounds__(256) void k_hash_rows(int n_rows, const int32_t *__restrict__ rows, int n_seg,
                                                   const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_start,
                                                   const int32_t *__restrict__ seg_end, const int32_t *__restrict__ seg_idx,
                                                   const int32_t *__restrict__ tid_has_segs, int n_tid,
                                                   int32_t *__restrict__ cni0, int32_t *__restrict__ cni1,
                                                   long long *__restrict__ keys, int32_t *__restrict__ vals,
                                                   uint32_t *__restrict__ n_valid) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    int nv = 0;
    if (r < n_rows) {
        const int32_t *f = rows + 8ll * r;
        const int t = f[2], a = f[3], b = f[4];
        int c0 = -3, c1 = -3;                                  // -3: the contig has no CN segments (ibg:210: set([-1]))
        if (t >= 0 && t < n_tid && tid_has_segs[t]) {
            c0 = seg_lookup(seg_tid, seg_start, seg_end, seg_idx, n_seg, t, a < b ? a : b);
            c1 = seg_lookup(seg_tid, seg_start, seg_end, seg_idx, n_seg, t, a < b ? b : a);
        }
        cni0[r] = c0;
        cni1[r] = c1;
        const bool v0 = c0 >= 0, v1 = c1 >= 0 && c1 != c0;
        keys[2ll * r] = v0 ? (((long long)t << 32) | (long long)c0) : HASH_INVALID;
        keys[2ll * r + 1] = v1 ? (((long long)t << 32) | (long long)c1) : HASH_INVALID;
        vals[2ll * r] = 2 * r;
        vals[2ll * r + 1] = 2 * r + 1;
        nv = (v0 ? 1 : 0) + (v1 ? 1 : 0);
    }
    // one atomic per wave
    for (int d = 32; d > 0; d >>= 1) nv += __shfl_xor(nv, d);
    if ((threadIdx.x & 63) == 0 && nv) atomicAdd(n_valid, (uint32_t)nv);
}

__global__ void k_hash_finish(int n, const int32_t *__restrict__ vals_sorted, int32_t *__restrict__ e_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) e_row[i] = vals_sorted[i] >> 1;
}

extern "C" int coral_hash_rows(int32_t n_rows, const int32_t *rows, int32_t n_seg, const int32_t *seg_tid,
                               const int32_t *seg_start, const int32_t *seg_end, const int32_t *seg_idx,
                               const int32_t *tid_has_segs, int32_t n_tid, void *workspace, int64_t workspace_bytes,
                               int32_t *cni0, int32_t *cni1, int64_t *e_key, int32_t *e_row, int32_t *n_ent, void *stream) {
    if (!n_ent) return sa_err(CORAL_ERR_ARG, "hash_rows: n_ent is null");
    *n_ent = 0;
    if (n_rows < 0 || n_seg < 0 || n_tid < 0) return sa_err(CORAL_ERR_ARG, "hash_rows: negative size");
    if (n_rows == 0) return CORAL_OK;
    if (!rows || !tid_has_segs || !cni0 || !cni1 || !e_key || !e_row || (n_seg > 0 && (!seg_tid || !seg_start || !seg_end || !seg_idx)))
        return sa_err(CORAL_ERR_ARG, "hash_rows: null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t ne = 2 * (size_t)n_rows;
    size_t cub_bytes = 0;
    {
        long long *k64 = nullptr;
        int32_t *v32 = nullptr;
        hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, k64, k64, v32, v32, (int)ne, 0, 63, s);
    }
    size_t need;
    {
        char *p = nullptr;
        carve<char>(p, cub_bytes);
        carve<long long>(p, ne);
        carve<int32_t>(p, ne);
        carve<int32_t>(p, ne);
        carve<uint32_t>(p, 4);
        need = (size_t)p;
    }
    if (!workspace || (size_t)workspace_bytes < need) {
        *n_ent = (int32_t)(need >> 20) + 1;               // MiB needed
        return sa_err(CORAL_ERR_CAPACITY, "hash_rows: workspace too small");
    }
    char *p = (char *)workspace;
    void *cub_tmp = carve<char>(p, cub_bytes);
    long long *keys = carve<long long>(p, ne);
    int32_t *vals = carve<int32_t>(p, ne), *vals_s = carve<int32_t>(p, ne);
    uint32_t *counter = carve<uint32_t>(p, 4);
    (void)hipMemsetAsync(counter, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_hash_rows, dim3((n_rows + 255) / 256), dim3(256), 0, s, (int)n_rows, rows, (int)n_seg, seg_tid, seg_start,
                       seg_end, seg_idx, tid_has_segs, (int)n_tid, cni0, cni1, keys, vals, counter);
    size_t tb = cub_bytes;
    // sorted keys go straight to e_key; the invalid entries (key 2^63 - 1) sort behind every valid one
    hipcub::DeviceRadixSort::SortPairs(cub_tmp, tb, keys, reinterpret_cast<long long *>(e_key), vals, vals_s, (int)ne, 0, 63, s);
    hipLaunchKernelGGL(k_hash_finish, dim3((int)((ne + 255) / 256)), dim3(256), 0, s, (int)ne, vals_s, e_row);
    uint32_t h = 0;
    hipError_t e = hipMemcpyAsync(&h, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return sa_err(CORAL_ERR_HIP, hipGetErrorString(e));
    *n_ent = (int32_t)h;
    return CORAL_OK;
}
