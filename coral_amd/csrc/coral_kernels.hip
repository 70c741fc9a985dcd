// coral_kernels.hip — gfx950 (MI355X, CDNA4) kernels behind include/coral_hip.h.
//
// All of this is integer / indexing work bounded by HBM bandwidth (no MFMA): 64-lane waves stream
// 16-byte CIGAR quads (1 KiB per wave-instruction), wave-level scans give every op its reference
// offset, ballot + popcount compaction emits the rare candidates, and integer atomics (order-free,
// hence bit-exact) reduce per-segment sums.
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/coral_hip.h"

#define WAVE 64
typedef uint32_t cquad_t __attribute__((ext_vector_type(4)));   // one 16-byte CIGAR quad = one dwordx4 load
#define SCAN_BLOCK 256          // 4 waves per workgroup
#define OP_PAD_QUAD 0x0000000Fu // op 15, length 0: consumes nothing

// op classes as bit masks over the BAM op code (MIDNSHP=X are 0..8; 15 = layout padding)
#define MASK_REF 0x18Du   // M D N = X advance the reference
#define MASK_ALN 0x181u   // M = X are aligned blocks (pysam get_blocks / count_coverage)
#define MASK_QRY 0x1B3u   // M I S H = X count towards infer_read_length()
#define MASK_DN 0x00Cu    // D N: advance the reference without being aligned (what a gap is made of)
#define MASK_ISH 0x032u   // I S H: count towards infer_read_length() without being aligned

static thread_local char g_err[512] = "";

static int set_err(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

static int hip_err(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return CORAL_ERR_HIP;
}

// CORAL_KERNELS_SHA: first 16 hex digits of sha256(coral_kernels.hip), passed by the build (__graft_entry__.build): bench.py serves a
// committed PMC traffic figure as `roofline.traffic` only when it was measured on a library built from this very source.
#ifndef CORAL_KERNELS_SHA
#define CORAL_KERNELS_SHA "unknown"
#endif
extern "C" const char *coral_version(void) { return "coral_hip 0.3 (gfx950) kernels:" CORAL_KERNELS_SHA; }
extern "C" const char *coral_last_error(void) { return g_err; }
// name of the kernel coral_cigar_scan launches, as rocprofv3 prints it (bench.py puts it next to the roofline figures)
extern "C" const char *coral_scan_kernel_name(void);

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_add(int x, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    return x;
}

__device__ __forceinline__ long long wave_reduce_add64(long long x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_zero(int src) {
    return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, 0xf, true);
}

// inclusive add-scan over the 64 lanes: 4 row shifts + 2 row broadcasts (no LDS crossbar)
__device__ __forceinline__ int wave_incl_scan_add_dpp(int x) {
    x += dpp_zero<0x111, 0xf>(x);   // row_shr:1
    x += dpp_zero<0x112, 0xf>(x);   // row_shr:2
    x += dpp_zero<0x114, 0xf>(x);   // row_shr:4
    x += dpp_zero<0x118, 0xf>(x);   // row_shr:8
    x += dpp_zero<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
    x += dpp_zero<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ int wave_sum_dpp(int x) {
    return __builtin_amdgcn_readlane(wave_incl_scan_add_dpp(x), 63);
}

// ---------------------------------------------------------------------------------------------
// K1  k_cigar_scan_v4 — the fused CIGAR pass (production kernel of coral_cigar_scan).
//
// Per alignment record: Σ aligned bases, infer_read_length, first / last aligned block, and every gap > min_gap between
// consecutive aligned blocks (the get_blocks() walk of /root/reference/src/infer_breakpoint_graph.py:750-762 plus the
// per-record sums count_coverage / infer_read_length need, ibg:131, :1031-1034).  HBM-bound: 4 bytes per CIGAR op, read once.
//
// Structure (what the measurements asked for: profiles/r01_scan_variants.md, profiles/r02_scan.md):
//   * ONE MOVING WINDOW.  The op array of all records is one contiguous stream of 16-byte quads (records are padded to whole
//     quads).  Work is handed out in groups of consecutive records from a cursor, to as many waves as are resident at once
//     (grid = what fits the chip): all waves read close to each other and the window moves through the array front to back.
//   * STATIC REGISTER RING.  RING quads per lane, loop unrolled by RING so that no quad is ever moved: while a chunk (64 lanes x
//     16 bytes = 1 KiB) is processed, the next RING - 1 are in flight as full, 128-byte-aligned global_load_dwordx4 that ignore
//     record boundaries (scalar base + per-lane offset, non-temporal), retired by counted waits.  An LDS ring fed by LDS-DMA
//     was measured slower (lab/coral_scan_v3_ldsdma.hip.inc).
//   * LANE-LOCAL FAST PATH.  A chunk in which every lane holds an aligned op and at most min_gap / 2 of D / N cannot end a
//     reportable gap, and the three per-record sums are order-free: such a chunk costs ~45 VALU instructions and no cross-lane
//     step.  Record boundaries inside such a chunk are lane masks; the wave-level sums run once per RECORD, not per chunk.
//     Only flagged chunks (and the one after a flagged last lane) take the exact scan (`scan_piece`: DPP add-scan for the
//     reference offsets, max-scan + distance tests for the gaps), once per record piece, outside the pipelined loop.
//   * SUMMARIES LEAVE AS FULL LINES.  The four per-record results are one 16-byte row; rows are parked in LDS and written per
//     group as one coalesced store (round 1 wrote four scattered 4-byte words per record: 7.3 x write amplification).
// ---------------------------------------------------------------------------------------------
struct ScanState {
    int carry_ref, carry_end, msum, qsum, first;
    bool tail_flagged;
};

// One piece = the lanes of one chunk that belong to the current record (the others hold OP_PAD_QUAD).  `quad0` = index within
// the record of lane 0's quad (may be negative for lanes in front of the record; those lanes are padding).
//
// Conservative gap filter: a gap is the sum of the reference-advancing NON-aligned ops (D, N) between two consecutive aligned
// ops.  With G(l) = that sum over the four ops of lane l, a lane is flagged when it holds real ops but no aligned op, or when
// G(l) > min_gap / 2.  If no lane of the piece is flagged and the last lane of the record's previous piece was not flagged
// either, two consecutive aligned ops are at most one lane apart, so every gap that ends in this piece is
// <= G(l) + G(l + 1) <= min_gap: nothing to report, and the running "end of the last aligned block" is the one of the highest
// lane holding an aligned op.  Exact for every input (tests/test_gpu_kernels.py: adversarial CIGAR set).
__device__ __forceinline__ void scan_piece(const cquad_t b0, const int lane, ScanState &st, const bool gaps_on, const int min_gap,
                                           const int half_gap, const int rec, const int quad0, const int p0,
                                           int32_t *__restrict__ gaps, uint32_t *__restrict__ gap_count, const uint32_t gap_cap) {
    int len[4], adv[4], aend[4], ref[4];
    int fal[4];                          // 0 / -1 per op, "is an aligned block"
    int tot = 0, asum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = b0[k];
        len[k] = (int)(w >> 4);
        // v_bfe_i32 takes its bit offset from the low 5 bits of the operand: op code + 16 * (length & 1); the class masks are
        // replicated into both halves, so the op word itself is the offset
        const int fr_ = __builtin_amdgcn_sbfe((int)(MASK_REF * 0x10001u), w, 1u);
        const int fa_ = __builtin_amdgcn_sbfe((int)(MASK_ALN * 0x10001u), w, 1u);
        const int fq_ = __builtin_amdgcn_sbfe((int)(MASK_QRY * 0x10001u), w, 1u);
        adv[k] = len[k] & fr_;
        aend[k] = len[k] & fa_;
        fal[k] = fa_;
        st.qsum += len[k] & fq_;
        asum += aend[k];
        tot += adv[k];
    }
    st.msum += asum;
    const int incl = wave_incl_scan_add_dpp(tot);
    ref[0] = st.carry_ref + incl - tot;
    const int m3 = fal[3], m2 = m3 | fal[2], m1 = m2 | fal[1], m0 = m1 | fal[0];
    const unsigned long long flagged = (__ballot(m0 == 0) & __ballot(b0[0] != OP_PAD_QUAD)) | __ballot(tot - asum > half_gap);
    const bool exact = flagged != 0ull || st.tail_flagged;
    st.tail_flagged = (flagged >> 63) != 0ull;
    if (!exact) {
        const unsigned long long has = __ballot(m0 != 0);   // 0 only for a piece without any aligned op (all padding)
        if (has != 0ull) {
            if (st.carry_end == 0) {        // (wave-uniform) the record's first block starts in this piece:
                // reference-advancing ops in front of the lane's first aligned op (masks only, no selects)
                const int n0 = ~fal[0], n1 = n0 & ~fal[1], n2 = n1 & ~fal[2];
                const int lf = ref[0] + (adv[0] & n0) + (adv[1] & n1) + (adv[2] & n2);
                st.first = __builtin_amdgcn_readlane(lf, (int)__builtin_ctzll(has));
            }
            // end of the last aligned block of the lane, relative to the lane's first op
            const int off_end = adv[0] + (adv[1] & m1) + (adv[2] & m2) + (adv[3] & m3);
            st.carry_end = __builtin_amdgcn_readlane(ref[0] + off_end, 63 - (int)__builtin_clzll(has));
        }
    } else {
        const bool aln[4] = {fal[0] != 0, fal[1] != 0, fal[2] != 0, fal[3] != 0};
        ref[1] = ref[0] + adv[0];
        ref[2] = ref[1] + adv[1];
        ref[3] = ref[2] + adv[2];
        // running "end of the last aligned block" inside the lane (0 = none yet)
        int run[4];
        run[0] = aln[0] ? ref[0] + aend[0] : 0;
        run[1] = aln[1] ? ref[1] + aend[1] : run[0];
        run[2] = aln[2] ? ref[2] + aend[2] : run[1];
        run[3] = aln[3] ? ref[3] + aend[3] : run[2];
        // previous block end seen by this lane = max over earlier lanes (ends never decrease), else the carry
        int mx = run[3];
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x111, 0xf, 0xf, true));
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x112, 0xf, 0xf, true));
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x114, 0xf, 0xf, true));
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x118, 0xf, 0xf, true));
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x142, 0xa, 0xf, true));
        mx = max(mx, __builtin_amdgcn_update_dpp(0, mx, 0x143, 0xc, 0xf, true));
        const int shifted = __builtin_amdgcn_update_dpp(st.carry_end, mx, 0x138, 0xf, 0xf, false);   // wave_shr:1
        const int prev_in = max(shifted, st.carry_end);
        if (st.carry_end == 0) {                    // (wave-uniform) still looking for the record's first block
            const unsigned long long has = __ballot(run[3] != 0);
            if (has != 0ull) {
                int lf = ref[3];
                lf = aln[2] ? ref[2] : lf;
                lf = aln[1] ? ref[1] : lf;
                lf = aln[0] ? ref[0] : lf;
                st.first = __builtin_amdgcn_readlane(lf, (int)__builtin_ctzll(has));
            }
        }
        // gap test, branch-free: distance from the previous block end (a huge "previous" when there is none)
        const int none = 0x3fffffff;
        const int pin = prev_in == 0 ? none : prev_in;
        const int pv1 = run[0] ? run[0] : pin, pv2 = run[1] ? run[1] : pin, pv3 = run[2] ? run[2] : pin;
        const bool h0 = aln[0] && (ref[0] - pin > min_gap);
        const bool h1 = aln[1] && (ref[1] - pv1 > min_gap);
        const bool h2 = aln[2] && (ref[2] - pv2 > min_gap);
        const bool h3 = aln[3] && (ref[3] - pv3 > min_gap);
        if (gaps_on && __ballot(h0 | h1 | h2 | h3) != 0ull) {          // rare: some lane holds a large gap
            const int pv[4] = {pin, pv1, pv2, pv3};
            const bool hit[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (hit[k]) {
                    const uint32_t slot = atomicAdd(gap_count, 1u);
                    if (slot < gap_cap) {
                        int4 row = make_int4(rec, (quad0 + lane) * 4 + k, p0 + pv[k], p0 + ref[k]);
                        reinterpret_cast<int4 *>(gaps)[slot] = row;
                    }
                }
            }
        }
        st.carry_end = max(st.carry_end, __builtin_amdgcn_readlane(mx, 63));
    }
    st.carry_ref += __builtin_amdgcn_readlane(incl, 63);
}

// First record r in [0, n_rec] with cigar_off[r] >= target (cigar_off ascending, n_rec + 1 entries): 64 probes per step.
__device__ __forceinline__ long long first_record_at(const int64_t *__restrict__ cigar_off, long long n_rec, long long target, int lane) {
    long long lo = 0, hi = n_rec;                 // answer in [lo, hi]
    while (hi - lo > 64) {
        const long long idx = lo + ((hi - lo) * (long long)(lane + 1)) / 65;        // lo < idx_0 < ... < idx_63 < hi
        const bool ge = cigar_off[idx] >= target;
        const unsigned long long m = __ballot(ge);                                  // 0..0 1..1 from some lane on
        const int f = m == 0ull ? 64 : (int)__builtin_ctzll(m);                     // first lane whose probe is >= target
        const long long lo_new = f == 0 ? lo : (lo + ((hi - lo) * (long long)f) / 65) + 1;
        const long long hi_new = f == 64 ? hi : lo + ((hi - lo) * (long long)(f + 1)) / 65;
        lo = lo_new;
        hi = hi_new;
    }
    const long long idx = lo + lane;
    const bool ge = idx <= hi && cigar_off[idx <= n_rec ? idx : n_rec] >= target;
    const unsigned long long m = __ballot(ge);
    return m == 0ull ? hi : lo + (long long)__builtin_ctzll(m);
}

// One 1 KiB wave load (64 lanes x 16 bytes, non-temporal) that the compiler does NOT track: the scan kernel keeps a fixed number
// of them in flight and retires them with its own counted waits (`ring_wait`), because the compiler's wait insertion falls back
// to vmcnt(0) at the loop's exit latch.  Every such load must be retired (`ring_wait<0>`) before its register can be reused.
__device__ __forceinline__ void ring_load(cquad_t &q, const char *wave_uniform_base, uint32_t lane_off) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(q) : "v"(lane_off), "s"(wave_uniform_base) : "memory");
}
template <int OUTSTANDING>
__device__ __forceinline__ void ring_wait(cquad_t &q) {          // q is valid once at most OUTSTANDING younger loads are in flight
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "n"(OUTSTANDING) : "memory");
}

// What the fast path keeps per lane between two exact positions: the sums are lane-local (no wave scan per chunk).
//   position of the next op of the record = st.carry_ref + wave_sum(ref_lane)
template <int RING>
__global__ __launch_bounds__(SCAN_BLOCK) void k_cigar_scan_v4(
    long long n_rec, const int32_t *__restrict__ pos, const int32_t *__restrict__ flagmq,
    const int64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar, int min_gap, int min_mapq,
    int4 *__restrict__ summary, int32_t *__restrict__ gaps, uint32_t *__restrict__ counters, uint32_t gap_cap, int group) {
    uint32_t *__restrict__ gap_count = counters;
    __shared__ int4 park_all[SCAN_BLOCK / WAVE][WAVE];           // per wave: 1 KiB of parked summaries
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long wave = (long long)blockIdx.x * (SCAN_BLOCK / WAVE) + wib;
    const long long nwaves = (long long)gridDim.x * (SCAN_BLOCK / WAVE);
    const cquad_t pad = {OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD};

    // ---- work: groups of `group` consecutive records.  The first two groups of a wave are static (wave, wave + nwaves), later
    // ones come from the work cursor (counters[1], zero on entry).  All resident waves therefore stream one compact window of
    // the op array that moves through it front to back (measured at cfg3: 2.7 ms, against 3.5 ms when every wave walks its own
    // 1 / nwaves slice of the array), and the dynamic hand-out levels the unequal record lengths.  One cursor serves ~80 M
    // requests / s (measured), which is why a group is 24 records and not 4.  Group ids are known two rounds ahead and the
    // bounds of the next group are loaded while the current one is processed: a group change costs no dependent round trip.
    const long long total_q = cigar_off[n_rec] >> 2;
    const int last_rec = (int)n_rec - 1;                         // n_rec < 2^31
    if (group <= 0) {
        // a group is sized by WORK, ~11 000 quads (44 000 ops: 24 records of config 3, the measured optimum there; 6 records of
        // config 5's 100 kb reads, where 24-record groups left waves with one group more than others: 1.67 -> 1.46 ms)
        const long long avg_q = total_q / n_rec > 0 ? total_q / n_rec : 1;
        const long long gq = 11000 / avg_q;
        group = (int)(gq < 1 ? 1 : gq > 24 ? 24 : gq);
    }
    const int n_groups = (int)((n_rec + group - 1) / group);
    const int dyn_base = 2 * (int)nwaves;                        // first group id the cursor hands out
    int g = (int)wave, g_next = (int)(wave + nwaves);
    long long nx_s0, nx_s1, nx_e0;                               // cigar_off[first], [end], [first + 1] of the group about to start
    int nx_pos, nx_fm;
    auto request_group = [&](int gg) {                           // (clamped: valid loads for any gg)
        const long long a = (long long)gg * group, e = a + group;
        const int first = a < n_rec ? (int)a : (int)n_rec, end = e < n_rec ? (int)e : (int)n_rec;
        nx_s0 = cigar_off[first];
        nx_s1 = cigar_off[end];
        nx_e0 = cigar_off[first < (int)n_rec ? first + 1 : first];
        nx_pos = pos[first < last_rec ? first : last_rec];
        nx_fm = flagmq[first < last_rec ? first : last_rec];
    };
    request_group(g);
    while (g < n_groups) {
    unsigned ticket = 0;
    if (lane == 0) ticket = atomicAdd(counters + 1, 1u);
    const int ra = g * group;
    const int rb = ra + group < (int)n_rec ? ra + group : (int)n_rec;
    const long long s0 = nx_s0 >> 2, s1 = nx_s1 >> 2;            // quad range of the group
    const long long e0 = nx_e0 >> 2;
    const int first_pos = nx_pos, first_fm = nx_fm;
    request_group(g_next);
    const long long base = s0 & ~7ll;                            // loads start on a 128-byte line
    const int n_chunks = (int)((s1 - base + WAVE - 1) / WAVE);   // may be 0 (only empty records)

    // ---- loads: full 1 KiB chunks that ignore record boundaries, wave-uniform base + per-lane byte offset; only the lanes of
    // the stream's very last chunk are clamped (they re-read the last quad and are masked out as "not in any record")
    const char *wave_src = reinterpret_cast<const char *>(cigar) + base * 16;
    const long long room_bytes = (total_q - base) * 16 - 16;     // largest valid byte offset of a quad load from wave_src
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto lane_offset = [&](int c) -> uint32_t {                  // the lane's byte offset inside chunk c, clamped to the stream's end
        const long long room = room_bytes - (long long)c * (WAVE * 16);          // >= 0 for every chunk of the wave
        const uint32_t lim = (room >> 31) != 0 ? 0x7fffffffu : (uint32_t)room;
        return lane_off < lim ? lane_off : lim;
    };
    auto load = [&](int c) -> cquad_t {                          // compiler-tracked (exact path: 0.1 % of the chunks).  A plain load, NOT
        // non-temporal: only the ring's loads carry `nt`, which is how tools/check_scan_ring.py tells them apart in the disassembly
        return *reinterpret_cast<const cquad_t *>(wave_src + (long long)c * (WAVE * 16) + lane_offset(c));
    };

    // ---- record cursor (metadata is wave-uniform: scalar loads, requested one record ahead); positions relative to `base`
    int r = ra;
    int rec_start = (int)(s0 - base), rec_end = (int)(e0 - base);
    int p0 = first_pos;
    bool gaps_on = ((first_fm >> 16) & 0xff) >= min_mapq;
    long long n_end = 0;
    int n_pos = 0, n_fm = 0;
    auto request = [&](int rr) {                                 // unconditional (clamped) loads: no select on the loaded value
        const int c = rr <= last_rec ? rr : last_rec;
        n_end = cigar_off[c + 1];
        n_pos = pos[c];
        n_fm = flagmq[c];
    };
    request(ra + 1);
    ScanState st = {0, 0, 0, 0, 0, false};
    int ref_lane = 0;                                            // (per lane) reference advance of the fast chunks since the last sync
    bool dirty = false;                                          // ref_lane != 0 somewhere: st.carry_ref / st.carry_end are behind
    bool have_first = false;                                     // the record's first aligned block is known (st.first)
    const int half_gap = min_gap >> 1;
    int parked = 0;                                              // summaries waiting in LDS (wave-uniform)
    int4 *park = &park_all[wib][0];
    // make st.carry_ref / st.carry_end exact again at the start of chunk c.  While dirty, the previous chunk belongs to the
    // current record and its last lane holds an aligned op: the D / N ops behind that op (normally none) are read back with
    // one scalar load instead of being tracked per chunk.
    auto sync_at = [&](int c) {
        if (dirty) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(wave_src) + ((long long)c * WAVE - 1) * 4;
            int trail = 0;
            bool open = true;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const uint32_t x = w[k];
                open = open && ((MASK_ALN >> (x & 15u)) & 1u) == 0u;
                if (open && ((MASK_REF >> (x & 15u)) & 1u)) trail += (int)(x >> 4);
            }
            st.carry_ref += wave_sum_dpp(ref_lane);
            st.carry_end = st.carry_ref - trail;
            ref_lane = 0;
            dirty = false;
        }
    };
    auto flush = [&]() {                                         // `parked` rows -> summary[r - parked .. r)
        if (lane < parked) summary[r - parked + lane] = park[lane];
        parked = 0;
    };
    auto emit = [&](int first_out, int end_out) {                // the current record is complete
        const int ms = wave_sum_dpp(st.msum);
        const int qs = wave_sum_dpp(st.qsum);
        if (lane == 0) park[parked] = make_int4(ms, qs, first_out, end_out);
        ++parked;
        st = {0, 0, 0, 0, 0, false};
        ref_lane = 0;
        dirty = false;
        have_first = false;
        ++r;
        if (parked == WAVE) flush();
        rec_start = rec_end;
        rec_end = (int)((n_end >> 2) - base);
        p0 = n_pos;
        gaps_on = ((n_fm >> 16) & 0xff) >= min_mapq;
        request(r + 1);
    };
    auto finish_record = [&]() {                                 // exact state (not dirty)
        emit(st.carry_end > 0 ? p0 + st.first : -1, st.carry_end > 0 ? p0 + st.carry_end : -1);
    };

    // ---- fast chunk: returns false (state untouched) when the chunk needs the exact path.
    // Per-lane sums of the four ops; flags as in scan_piece: a lane without aligned op, or with more than min_gap / 2 of
    // D / N — then (and after a flagged last lane) the chunk is not fast.
    auto fast = [&](const cquad_t quad, const int c) __attribute__((always_inline)) -> bool {
        const int cq0 = c * WAVE, cq1 = cq0 + WAVE;
        int fal[4], radv[4];
        int asum = 0, rsum = 0, qsum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = quad[k];
            const int len = (int)(w >> 4);
            fal[k] = __builtin_amdgcn_sbfe((int)(MASK_ALN * 0x10001u), w, 1u);
            radv[k] = len & __builtin_amdgcn_sbfe((int)(MASK_REF * 0x10001u), w, 1u);
            asum += len & fal[k];
            rsum += radv[k];
            qsum += len & __builtin_amdgcn_sbfe((int)(MASK_QRY * 0x10001u), w, 1u);
        }
        const int m3 = fal[3], m2 = m3 | fal[2], m1 = m2 | fal[1], m0 = m1 | fal[0];
        const unsigned long long flagged = __ballot(m0 == 0) | __ballot(rsum - asum > half_gap);
        if (flagged != 0ull || st.tail_flagged || !(have_first || rec_start >= cq0)) return false;      // (wave-uniform)
        if (rec_start <= cq0 && rec_end > cq1) {                 // inside one record: lane-local sums only
            st.msum += asum;
            st.qsum += qsum;
            ref_lane += rsum;
            dirty = true;
            return true;
        }
        // ---- a chunk with record boundaries and nothing flagged: every lane holds an aligned op, no gap can be reported;
        // a record starting here has its first block in its first lane, one ending here has its last block in its last lane
        const int n0 = ~fal[0], n1 = n0 & ~fal[1], n2 = n1 & ~fal[2];
        const int lf = (radv[0] & n0) + (radv[1] & n1) + (radv[2] & n2);          // reference advance before the lane's first aligned op
        const int lt = (radv[3] & ~m3) + (radv[2] & ~m2) + (radv[1] & ~m1);       // ... behind the lane's last aligned op
        while (r < rb) {                                         // once per record present in this chunk
            const int a = (rec_start > cq0 ? rec_start : cq0) - cq0, b = (rec_end < cq1 ? rec_end : cq1) - cq0;
            if (b > a) {
                const bool in = (unsigned)(lane - a) < (unsigned)(b - a);
                if (rec_start >= cq0) {                          // the record starts in this chunk (st.carry_ref == 0)
                    st.first = __builtin_amdgcn_readlane(lf, a);
                    have_first = true;
                }
                st.msum += in ? asum : 0;
                st.qsum += in ? qsum : 0;
                ref_lane += in ? rsum : 0;
                dirty = true;
            }
            if (rec_end > cq1) break;                            // the record continues in the next chunk
            if (have_first) {
                const int end = st.carry_ref + wave_sum_dpp(ref_lane) - (b > a ? __builtin_amdgcn_readlane(lt, b - 1) : 0);
                emit(p0 + st.first, p0 + end);
            } else {
                emit(-1, -1);                                    // a record without ops
            }
            if (rec_start >= cq1) break;                         // the next record starts in a later chunk
        }
        return true;
    };
    // ---- exact chunk: wave scans per record piece (scan_piece)
    auto exact = [&](const cquad_t quad, const int c) {
        const int cq0 = c * WAVE, cq1 = cq0 + WAVE;
        sync_at(c);
        if (rec_start <= cq0 && rec_end > cq1) {                 // inside one record
            scan_piece(quad, lane, st, gaps_on, min_gap, half_gap, r, cq0 - rec_start, p0, gaps, gap_count, gap_cap);
        } else {
            const int lq = cq0 + lane;
            while (r < rb) {                                     // once per record present in this chunk
                const int a = rec_start > cq0 ? rec_start : cq0, b = rec_end < cq1 ? rec_end : cq1;
                if (b > a) {
                    cquad_t piece = pad;
                    if (lq >= a && lq < b) piece = quad;
                    scan_piece(piece, lane, st, gaps_on, min_gap, half_gap, r, cq0 - rec_start, p0, gaps, gap_count, gap_cap);
                }
                if (rec_end > cq1) break;                        // the record continues in the next chunk
                finish_record();
                if (rec_start >= cq1) break;                     // the next record starts in a later chunk
            }
        }
        have_first = st.carry_end > 0;
    };

    // ---- static ring of RING register quads, loop unrolled by RING so that no quad is ever moved: while chunk c is processed
    // the loads of c + 1 .. c + RING - 1 are in flight (the compiler's counted s_waitcnt vmcnt retires them in order).  Only the
    // fast chunk code is replicated; a chunk that needs the exact path leaves the pipeline (the loads in flight are dropped),
    // is handled by the one copy of the exact code below, and the pipeline restarts behind it — rare by construction.
    int c = 0;
    const int last_chunk = n_chunks - 1;
    cquad_t q[RING];
    auto issue = [&](cquad_t &dst, int cc) {                     // every ring load is issued unconditionally (chunk index clamped to
        const int k = cc < last_chunk ? cc : last_chunk;         // the group's last chunk): the number in flight is the same on every
        ring_load(dst, wave_src + (long long)k * (WAVE * 16), lane_offset(k));       // path, and no register is ever copied while its load is in flight
    };
    auto drain = [&]() {
#pragma unroll
        for (int k = 0; k < RING; ++k) ring_wait<0>(q[k]);
    };
    while (c < n_chunks) {
#pragma unroll
        for (int k = 0; k < RING - 1; ++k) issue(q[k], c + k);
        for (;;) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                issue(q[(k + RING - 1) % RING], c + k + RING - 1);
                ring_wait<RING - 1>(q[k]);
                if (c + k >= n_chunks) { c = n_chunks; goto leave; }
                if (!fast(q[k], c + k)) { c += k; goto leave; }
            }
            c += RING;
        }
    leave:
        drain();                                                 // loads still in flight are dropped
        if (c >= n_chunks) break;
        do {                                                     // chunk c, and the chunks a flagged last lane drags along
            exact(load(c), c);
            ++c;
        } while (c < n_chunks && st.tail_flagged);
    }
    while (r < rb) finish_record();                              // records without any op at the end of the range
    if (parked) flush();
    g = g_next;
    g_next = dyn_base + (int)__builtin_amdgcn_readfirstlane(ticket);
    }
}

extern "C" const char *coral_scan_kernel_name(void) { return "k_cigar_scan_v4<6>"; }

static int check_records(const coral_records_t *rec) {
    if (!rec) return set_err(CORAL_ERR_ARG, "records: null");
    if (rec->n_rec < 0 || rec->n_rec > 0x7fffffffLL) return set_err(CORAL_ERR_ARG, "records: n_rec out of range");
    if (rec->n_rec > 0 && (!rec->tid || !rec->pos || !rec->end || !rec->flagmq || !rec->n_cigar || !rec->cigar_off || !rec->cigar))
        return set_err(CORAL_ERR_ARG, "records: null array");
    if (((uintptr_t)rec->cigar) & 15u) return set_err(CORAL_ERR_ARG, "records: cigar base must be 16-byte aligned");
    return CORAL_OK;
}

// workgroups of the scan: as many as are resident at once (register-bound), never more waves than records.
// CORAL_SCAN_RING (4 / 6 / 8 / 12), CORAL_SCAN_GROUP (1 .. 64) and CORAL_SCAN_WG_PER_CU: tuning overrides (tools/scan_sweep.sh).
#define SCAN_RING_DEFAULT 6
static int g_ring = 0, g_wg_per_cu = 0, g_group = 0;          // g_group 0: the kernel sizes its groups by CIGAR ops (see there)
template <int RING>
static int scan_grid(long long n_rec) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cigar_scan_v4<RING>, SCAN_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (g_wg_per_cu > 0 && g_wg_per_cu < per_cu) per_cu = g_wg_per_cu;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    long long blocks = (n_rec + (SCAN_BLOCK / WAVE) - 1) / (SCAN_BLOCK / WAVE);
    if (blocks > (long long)per_cu * cus) blocks = (long long)per_cu * cus;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" int coral_cigar_scan(const coral_records_t *rec, int32_t min_gap, int32_t min_mapq, int32_t *summary, int32_t *gaps,
                                uint32_t *gap_count, uint32_t gap_cap, void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (rec->n_rec == 0) return CORAL_OK;
    if (!summary || !gap_count || (gap_cap && !gaps)) return set_err(CORAL_ERR_ARG, "cigar_scan: null output");
    if ((((uintptr_t)gaps) & 15u) || (((uintptr_t)summary) & 15u)) return set_err(CORAL_ERR_ARG, "cigar_scan: summary and gaps must be 16-byte aligned");
    static std::once_flag tuning_once;          // (the entry point is re-entrant: the overrides are read exactly once, by one thread)
    std::call_once(tuning_once, [] {
        const char *a = getenv("CORAL_SCAN_RING"), *b = getenv("CORAL_SCAN_WG_PER_CU"), *c = getenv("CORAL_SCAN_GROUP");
        g_wg_per_cu = b ? atoi(b) : 0;
        if (c && atoi(c) >= 1 && atoi(c) <= 64) g_group = atoi(c);
        g_ring = a ? atoi(a) : SCAN_RING_DEFAULT;
    });
#define LAUNCH_V4(R)                                                                                                              \
    do {                                                                                                                          \
        const int blocks_ = scan_grid<R>(rec->n_rec);                                                                             \
        hipLaunchKernelGGL(k_cigar_scan_v4<R>, dim3(blocks_), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,                           \
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->cigar_off, rec->cigar, (int)min_gap, (int)min_mapq, \
                           reinterpret_cast<int4 *>(summary), gaps, gap_count, gap_cap, g_group);                                 \
    } while (0)
    if (g_ring == 4) LAUNCH_V4(4);
    else if (g_ring == 8) LAUNCH_V4(8);
    else if (g_ring == 12) LAUNCH_V4(12);
    else LAUNCH_V4(6);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "cigar_scan launch");
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// K2  segment coverage
//   phase A (thread per record): classify against the sorted disjoint segment table; records fully
//            inside one segment add mbases; boundary-straddling records are compacted into a list;
//   phase B (wave per straddler): walk the CIGAR once for up to four overlapped segments.
// ---------------------------------------------------------------------------------------------
#define COV_BLOCK 256
#define COV_LDS_SEGS 2048

__device__ __forceinline__ int first_seg_ending_after(const int32_t *__restrict__ seg_tid,
                                                      const int32_t *__restrict__ seg_end, int n_seg, int tid, int p) {
    // first j with (seg_tid[j], seg_end[j]) > (tid, p)
    int lo = 0, hi = n_seg;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int t = seg_tid[mid];
        const bool gt = (t > tid) || (t == tid && seg_end[mid] > p);
        if (gt) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(COV_BLOCK) void k_seg_classify(
    long long n_rec, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
    const int32_t *__restrict__ end, const int32_t *__restrict__ flagmq, const int32_t *__restrict__ n_cigar,
    const int4 *__restrict__ summary, int n_seg,
    const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_end,
    unsigned long long *__restrict__ n_reads, unsigned long long *__restrict__ n_bases,
    uint32_t *__restrict__ strad, uint32_t *__restrict__ strad_count) {
    __shared__ unsigned long long bins[2 * COV_LDS_SEGS];
    const bool use_lds = n_seg <= COV_LDS_SEGS;
    if (use_lds) {
        for (int i = threadIdx.x; i < 2 * n_seg; i += COV_BLOCK) bins[i] = 0ull;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * COV_BLOCK;
    const long long n_round = (n_rec + stride - 1) / stride * stride;   // keep whole waves in the loop for ballots
    for (long long r = (long long)blockIdx.x * COV_BLOCK + threadIdx.x; r < n_round; r += stride) {
        bool is_strad = false;
        if (r < n_rec) {
            const int t = tid[r], p = pos[r], e = end[r];
            int j = (t >= 0) ? first_seg_ending_after(seg_tid, seg_end, n_seg, t, p) : n_seg;
            if (j < n_seg && seg_tid[j] == t && seg_start[j] < e) {
                const int4 sm = summary[r];               // mbases, qinfer, first block start, last block end
                const bool counts = sm.y > 0;
                const bool has_seq = ((flagmq[r] >> 24) & 1) && n_cigar[r] > 0;
                if (seg_start[j] <= p && e <= seg_end[j]) {
                    if (use_lds) {
                        if (counts) atomicAdd(&bins[2 * j], 1ull);
                        if (has_seq) atomicAdd(&bins[2 * j + 1], (unsigned long long)sm.x);
                    } else {
                        if (counts) atomicAdd(&n_reads[j], 1ull);
                        if (has_seq) atomicAdd(&n_bases[j], (unsigned long long)sm.x);
                    }
                } else {
                    for (; j < n_seg && seg_tid[j] == t && seg_start[j] < e; ++j) {
                        if (counts) {
                            if (use_lds) atomicAdd(&bins[2 * j], 1ull); else atomicAdd(&n_reads[j], 1ull);
                        }
                    }
                    is_strad = has_seq;
                }
            }
        }
        const unsigned long long m = __ballot(is_strad);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(strad_count, (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (is_strad) {
                const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                strad[base + __popcll(m & below)] = (uint32_t)r;
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_seg; i += COV_BLOCK) {
            const unsigned long long a = bins[2 * i], b = bins[2 * i + 1];
            if (a) atomicAdd(&n_reads[i], a);
            if (b) atomicAdd(&n_bases[i], b);
        }
    }
}

#define WALK_SEGS 4          // segments evaluated per walk of a record's CIGAR (a record rarely straddles more than two)

// first j with (seg_tid[j], seg_end[j]) > (tid, p) — the whole wave probes 64 table entries per step (two steps for 4096 segments
// instead of twelve dependent loads)
__device__ __forceinline__ int first_seg_ending_after_wave(const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_end,
                                                           int n_seg, int tid, int p, int lane) {
    int lo = 0, hi = n_seg;                       // answer in [lo, hi]
    while (hi - lo > 64) {
        const int idx = lo + (int)(((long long)(hi - lo) * (lane + 1)) / 65);       // lo < idx_0 < ... < idx_63 < hi
        const int t = seg_tid[idx];
        const unsigned long long m = __ballot((t > tid) || (t == tid && seg_end[idx] > p));
        const int f = m == 0ull ? 64 : (int)__builtin_ctzll(m);
        const int lo_new = f == 0 ? lo : lo + (int)(((long long)(hi - lo) * f) / 65) + 1;
        const int hi_new = f == 64 ? hi : lo + (int)(((long long)(hi - lo) * (f + 1)) / 65);
        lo = lo_new;
        hi = hi_new;
    }
    const int idx = lo + lane;
    const int ii = idx < n_seg ? idx : n_seg - 1;
    const int t = seg_tid[ii];
    const unsigned long long m = __ballot(idx < hi && ((t > tid) || (t == tid && seg_end[ii] > p)));
    return m == 0ull ? hi : lo + (int)__builtin_ctzll(m);
}

// phase B: one wave per straddling record; the record's CIGAR is walked once for up to WALK_SEGS overlapped segments, the next
// chunk and the next straddler's fields are requested while the current ones are evaluated.  The thousands of records that
// straddle the same segment boundary at amplicon depth all add to the same two counters: each workgroup takes a CONTIGUOUS
// piece of the straddler list (neighbours in file order), sums in LDS and issues one global atomic per touched segment (same-
// address global atomics run at ~80 M / s: one per straddler made this kernel 0.45 - 0.86 ms per call at config 3).
__global__ __launch_bounds__(SCAN_BLOCK) void k_seg_walk(
    const uint32_t *__restrict__ strad, const uint32_t *__restrict__ strad_count,
    const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, const int32_t *__restrict__ end,
    const int32_t *__restrict__ n_cigar, const int64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    int n_seg, const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_start,
    const int32_t *__restrict__ seg_end, unsigned long long *__restrict__ n_bases) {
    __shared__ unsigned long long bins[COV_LDS_SEGS];
    const bool use_lds = n_seg <= COV_LDS_SEGS;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_seg; i += SCAN_BLOCK) bins[i] = 0ull;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const long long n_strad = *strad_count;
    const cquad_t pad = {OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD};
    const long long per_block = (n_strad + gridDim.x - 1) / gridDim.x;
    const long long w_begin = (long long)blockIdx.x * per_block;
    const long long w_end = w_begin + per_block < n_strad ? w_begin + per_block : n_strad;
    const int nwaves = SCAN_BLOCK / WAVE;                        // stride inside the block's piece
    const long long wave = w_begin + wib;
    if (wave < w_end) {
    long long r_next = strad[wave];
    int t_next = tid[r_next], p_next = pos[r_next], e_next = end[r_next], nc_next = n_cigar[r_next];
    long long off_next = cigar_off[r_next];
    for (long long w = wave; w < w_end; w += nwaves) {
        const int t = t_next, p0 = p_next, e0 = e_next;
        const int nq = (nc_next + 3) >> 2;
        const cquad_t *__restrict__ q = reinterpret_cast<const cquad_t *>(cigar + off_next);
        if (w + nwaves < w_end) {                                // (wave-uniform) the next straddler of this wave
            r_next = strad[w + nwaves];
            t_next = tid[r_next]; p_next = pos[r_next]; e_next = end[r_next]; nc_next = n_cigar[r_next];
            off_next = cigar_off[r_next];
        }
        cquad_t cur = lane < nq ? q[lane] : pad;                 // chunk 0 is on its way while the segments are located
        int j = first_seg_ending_after_wave(seg_tid, seg_end, n_seg, t, p0, lane);
        while (j < n_seg && seg_tid[j] == t && seg_start[j] < e0) {
            // up to WALK_SEGS overlapped segments per walk, in record-relative coordinates (an unused slot is empty: s = e = 0)
            int s[WALK_SEGS], e[WALK_SEGS], e_max = 0, n_here = 0;
#pragma unroll
            for (int k = 0; k < WALK_SEGS; ++k) {
                const bool on = j + k < n_seg && seg_tid[j + k] == t && seg_start[j + k] < e0;
                s[k] = on ? seg_start[j + k] - p0 : 0;
                e[k] = on ? seg_end[j + k] - p0 : 0;
                e_max = on && e[k] > e_max ? e[k] : e_max;
                n_here += on ? 1 : 0;
            }
            int acc[WALK_SEGS] = {0, 0, 0, 0};       // per lane: < 2^31 (a record's aligned bases)
            int carry_ref = 0;
            for (int c = 0; c < nq && carry_ref < e_max; c += WAVE) {
                const cquad_t nxt = c + WAVE + lane < nq ? q[c + WAVE + lane] : pad;       // in flight while `cur` is evaluated
                int len[4], adv[4], fal[4];
                int tot = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v = cur[k];
                    len[k] = (int)(v >> 4);
                    adv[k] = len[k] & __builtin_amdgcn_sbfe((int)(MASK_REF * 0x10001u), v, 1u);
                    fal[k] = __builtin_amdgcn_sbfe((int)(MASK_ALN * 0x10001u), v, 1u);
                    tot += adv[k];
                }
                const int incl = wave_incl_scan_add_dpp(tot);
                int ref = carry_ref + incl - tot;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int a0 = ref, a1 = ref + (len[k] & fal[k]);            // the aligned block of this op (empty if not aligned)
#pragma unroll
                    for (int g = 0; g < WALK_SEGS; ++g) {
                        const int ov = min(a1, e[g]) - max(a0, s[g]);
                        acc[g] += ov > 0 ? ov : 0;
                    }
                    ref += adv[k];
                }
                carry_ref += __builtin_amdgcn_readlane(incl, 63);
                cur = nxt;
            }
#pragma unroll
            for (int g = 0; g < WALK_SEGS; ++g) {
                if (g < n_here) {                                                // (wave-uniform)
                    const long long sum = wave_reduce_add64((long long)acc[g]);
                    if (lane == 0 && sum) {
                        if (use_lds) atomicAdd(&bins[j + g], (unsigned long long)sum);
                        else atomicAdd(&n_bases[j + g], (unsigned long long)sum);
                    }
                }
            }
            j += n_here;
            if (j < n_seg && seg_tid[j] == t && seg_start[j] < e0) cur = lane < nq ? q[lane] : pad;     // another walk: from the top
        }
    }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_seg; i += SCAN_BLOCK) {
            const unsigned long long b = bins[i];
            if (b) atomicAdd(&n_bases[i], b);
        }
    }
}

static int walk_grid(long long n_rec) {          // one wave per straddling record, up to 32 waves per CU
    long long blocks = (n_rec + (SCAN_BLOCK / WAVE) - 1) / (SCAN_BLOCK / WAVE);
    if (blocks > 2048) blocks = 2048;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int coral_segment_coverage(const coral_records_t *rec, const int32_t *summary, int32_t n_seg, const int32_t *seg_tid, const int32_t *seg_start,
                                      const int32_t *seg_end, unsigned long long *n_reads,
                                      unsigned long long *n_bases, uint32_t *strad, uint32_t *strad_count,
                                      void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (n_seg < 0) return set_err(CORAL_ERR_ARG, "segment_coverage: n_seg < 0");
    if (n_seg == 0 || rec->n_rec == 0) return CORAL_OK;
    if (!summary || !seg_tid || !seg_start || !seg_end || !n_reads || !n_bases || !strad || !strad_count)
        return set_err(CORAL_ERR_ARG, "segment_coverage: null argument");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (rec->n_rec + COV_BLOCK - 1) / COV_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_seg_classify, dim3((int)blocks), dim3(COV_BLOCK), 0, s, (long long)rec->n_rec, rec->tid,
                       rec->pos, rec->end, rec->flagmq, rec->n_cigar, reinterpret_cast<const int4 *>(summary), (int)n_seg, seg_tid, seg_start,
                       seg_end, n_reads, n_bases, strad, strad_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "seg_classify launch");
    hipLaunchKernelGGL(k_seg_walk, dim3(walk_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, s, strad, strad_count, rec->tid,
                       rec->pos, rec->end, rec->n_cigar, rec->cigar_off, rec->cigar, (int)n_seg, seg_tid, seg_start,
                       seg_end, n_bases);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "seg_walk launch");
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// K3  point cover — one workgroup per query point.  Records are in (tid, pos) order, so the records covering (t, p) lie in
// the index window  [first record at (t, p - max_span + 1),  first record behind (t, p)):  two cooperative 64-ary searches, then
// a strided look at `end` inside the window only (round 1 searched the point table once per RECORD: 2 M searches per call).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long rec_key(const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, long long r) {
    return ((unsigned long long)(uint32_t)tid[r] << 32) | (unsigned long long)(uint32_t)pos[r];    // unmapped (tid -1) sorts last, as in the file
}

// first record r in [0, n_rec] with (tid, pos) >= key — 64 probes per step, all lanes of the wave take part
__device__ __forceinline__ long long first_rec_at_key(const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, long long n_rec,
                                                      unsigned long long key, int lane) {
    long long lo = 0, hi = n_rec;                 // answer in [lo, hi]
    while (hi - lo > 64) {
        const long long idx = lo + ((hi - lo) * (long long)(lane + 1)) / 65;        // lo < idx_0 < ... < idx_63 < hi
        const unsigned long long m = __ballot(rec_key(tid, pos, idx) >= key);       // 0..0 1..1 from some lane on
        const int f = m == 0ull ? 64 : (int)__builtin_ctzll(m);
        const long long lo_new = f == 0 ? lo : (lo + ((hi - lo) * (long long)f) / 65) + 1;
        const long long hi_new = f == 64 ? hi : lo + ((hi - lo) * (long long)(f + 1)) / 65;
        lo = lo_new;
        hi = hi_new;
    }
    const long long idx = lo + lane;
    const bool ge = idx < hi && rec_key(tid, pos, idx < n_rec ? idx : n_rec - 1) >= key;
    const unsigned long long m = __ballot(ge);
    return m == 0ull ? hi : lo + (long long)__builtin_ctzll(m);
}

#define POINT_SLICES 32
__global__ __launch_bounds__(COV_BLOCK) void k_point_cover(
    long long n_rec, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
    const int32_t *__restrict__ end, int n_pts, const int32_t *__restrict__ pt_tid,
    const int32_t *__restrict__ pt_pos, int max_span, unsigned long long *__restrict__ pairs,
    uint32_t *__restrict__ pair_count, uint32_t pair_cap) {
    __shared__ uint32_t wave_hits[COV_BLOCK / WAVE];
    __shared__ uint32_t job_base;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    // POINT_SLICES workgroups share the window of one point (thousands of records at amplicon depth), interleaved by rounds.
    // Two passes over the slice: count, reserve the output range with ONE atomic per workgroup (a shared cursor serves only
    // ~80 M requests / s: with one atomic per wave and round the launch took 0.45 ms whatever else it did), then write.
    const long long stride = (long long)POINT_SLICES * COV_BLOCK;
    for (long long job = blockIdx.x; job < (long long)n_pts * POINT_SLICES; job += gridDim.x) {
        const int j = (int)(job / POINT_SLICES), slice = (int)(job % POINT_SLICES);
        const int t = pt_tid[j], p = pt_pos[j];
        if (t < 0 || p < 0) continue;                            // (block-uniform) nothing covers a negative coordinate
        const long long first_pos = max_span > 0 && p - max_span + 1 > 0 ? p - max_span + 1 : 0;
        // every wave runs both searches (same probes: all but the first hit the cache)
        const long long lo = first_rec_at_key(tid, pos, n_rec, ((unsigned long long)(uint32_t)t << 32) | (unsigned long long)first_pos, lane);
        const long long hi = first_rec_at_key(tid, pos, n_rec, ((unsigned long long)(uint32_t)t << 32) | ((unsigned long long)(uint32_t)p + 1ull), lane);
        uint32_t mine = 0;
        for (long long k = (long long)slice * COV_BLOCK + threadIdx.x; lo + k < hi; k += stride) mine += end[lo + k] > p ? 1u : 0u;
        const uint32_t incl = (uint32_t)wave_incl_scan_add_dpp((int)mine);
        __syncthreads();                                         // the previous job's readers of the shared words are done
        if (lane == 63) wave_hits[wib] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int k = 0; k < COV_BLOCK / WAVE; ++k) total += wave_hits[k];
            job_base = total ? atomicAdd(pair_count, total) : 0u;
        }
        __syncthreads();
        uint32_t slot = job_base + incl - mine;
        for (int k = 0; k < wib; ++k) slot += wave_hits[k];
        for (long long k = (long long)slice * COV_BLOCK + threadIdx.x; lo + k < hi; k += stride) {
            const long long r = lo + k;
            if (end[r] > p) {                                    // tid == t and pos <= p hold for the whole window
                if (slot < pair_cap) pairs[slot] = ((unsigned long long)(uint32_t)j << 32) | (unsigned long long)(uint32_t)r;
                ++slot;
            }
        }
    }
}

extern "C" int coral_point_cover(const coral_records_t *rec, int32_t n_pts, const int32_t *pt_tid,
                                 const int32_t *pt_pos, int32_t max_span, unsigned long long *pairs, uint32_t *pair_count,
                                 uint32_t pair_cap, void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (n_pts < 0) return set_err(CORAL_ERR_ARG, "point_cover: n_pts < 0");
    if (n_pts == 0 || rec->n_rec == 0) return CORAL_OK;
    if (!pt_tid || !pt_pos || !pair_count || (pair_cap && !pairs)) return set_err(CORAL_ERR_ARG, "point_cover: null argument");
    const long long jobs = (long long)n_pts * POINT_SLICES;
    const int blocks = (int)(jobs < 8192 ? jobs : 8192);
    hipLaunchKernelGGL(k_point_cover, dim3(blocks), dim3(COV_BLOCK), 0, (hipStream_t)stream, (long long)rec->n_rec,
                       rec->tid, rec->pos, rec->end, (int)n_pts, pt_tid, pt_pos, (int)max_span, pairs, pair_count, pair_cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "point_cover launch");
    return CORAL_OK;
}

extern "C" int coral_read_counter(const uint32_t *dev_counter, uint32_t *host_value, void *stream) {
    if (!dev_counter || !host_value) return set_err(CORAL_ERR_ARG, "read_counter: null");
    hipError_t e = hipMemcpyAsync(host_value, dev_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return hip_err(e, "read_counter copy");
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_err(e, "read_counter sync");
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// K4  pair table — the breakpoint candidate of EVERY pair of local alignments of every chimeric read, in one launch.
//
//     alignment2bp / alignment2bp_l (/root/reference/src/breakpoint_utilities.py:70-96, :129-186) look at the pairs
//     (k, k + 1) and (k - 1, k + 1) of a read's (qs, qe)-sorted alignments; whether a pair yields a candidate depends on the
//     amplicon intervals of the moment, but WHAT the candidate is (interval2bp, bu:289-295: canonical end order, the
//     orientations, the query gap) and every interval-independent test (query gap vs min_bp_match_cutoff, the three MAPQ
//     thresholds, strand change, the |gr - grr| > max(gap_, |0.2 gr|) test of alignment2bp_l) is a pure function of the
//     two table rows.  So the table is built once per graph build, directly from coral_sa_table's device rows, and the
//     interval search only filters it (coral_search_step / coral_search_within, host side, no launches, no syncs).
//
//     Layout: slot 2 * g + kind for table row g; kind 0 = pair (g, g + 1), kind 1 = pair (g - 1, g + 1) ("skip one", centre g).
//     One thread per read writes the slots of all its rows (32 bytes per slot, two dwordx4 stores); slots whose pair leaves
//     the read have bits == 0.  No per-read limit on the number of alignments.
// ---------------------------------------------------------------------------------------------
struct PairRow {
    int32_t c1, p1, c2, p2, gap, bits, a, b;
};

__device__ __forceinline__ PairRow make_pair(const int32_t *__restrict__ rows, int a, int b, int mid, bool skip,
                                             const int32_t *__restrict__ chr_rank, int n_tid, int cutoff, int min_mapq,
                                             int gap_, int gap_mapq) {
    const int32_t *ra_ = rows + 8ll * a, *rb_ = rows + 8ll * b;
    const int qe_a = ra_[1], tid_a = ra_[2], A_ra = ra_[3], A_rb = ra_[4], st_a = ra_[5], mq_a = ra_[6];
    const int qs_b = rb_[0], tid_b = rb_[2], B_ra = rb_[3], B_rb = rb_[4], st_b = rb_[5], mq_b = rb_[6];
    const int gap = qs_b - qe_a;
    bool ok = mq_a >= min_mapq && mq_b >= min_mapq;
    if (skip) ok = ok && rows[8ll * mid + 6] < gap_mapq;
    else ok = ok && (gap + cutoff >= 0);
    const int c1r = (tid_a >= 0 && tid_a < n_tid) ? chr_rank[tid_a] : -1;
    const int c2r = (tid_b >= 0 && tid_b < n_tid) ? chr_rank[tid_b] : -1;
    const bool first_form = (c2r < c1r) || (c2r == c1r && B_ra < A_rb);
    // alignment2bp_l's distance test for same-strand pairs (bu:145-160, :173-184)
    const int grr = (st_b == 0) ? (B_ra - A_rb) : (A_rb - B_ra);
    const long long d = (long long)gap - (long long)grr;
    const double lim = fmax((double)gap_, fabs((double)gap * 0.2));
    const bool far = (double)(d < 0 ? -d : d) > lim;
    PairRow r;
    int o1, o2;
    if (first_form) { r.c1 = tid_a; r.p1 = A_rb; o1 = st_a; r.c2 = tid_b; r.p2 = B_ra; o2 = 1 - st_b; }
    else { r.c1 = tid_b; r.p1 = B_ra; o1 = 1 - st_b; r.c2 = tid_a; r.p2 = A_rb; o2 = st_a; }
    r.gap = gap;
    r.bits = 1 | (ok ? 2 : 0) | (o1 << 2) | (o2 << 3) | (first_form ? 0 : 16) | (st_a != st_b ? 32 : 0) | (far ? 64 : 0) |
             ((c1r < 0 || c2r < 0) ? 128 : 0) | ((mq_a & 0xff) << 8) | ((mq_b & 0xff) << 16);
    r.a = a;
    r.b = b;
    return r;
}

__global__ __launch_bounds__(256) void k_bp_pairs(int n_reads, const int32_t *__restrict__ off, const int32_t *__restrict__ rows,
                                                  const int32_t *__restrict__ chr_rank, int n_tid, int cutoff, int min_mapq,
                                                  int gap_, int gap_mapq, PairRow *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const int base = off[r], n = off[r + 1] - base;
    const PairRow none = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < n; ++k) {
        const int g = base + k;
        out[2ll * g] = (k + 1 < n) ? make_pair(rows, g, g + 1, g, false, chr_rank, n_tid, cutoff, min_mapq, gap_, gap_mapq) : none;
        out[2ll * g + 1] = (k >= 1 && k + 1 < n) ? make_pair(rows, g - 1, g + 1, g, true, chr_rank, n_tid, cutoff, min_mapq, gap_, gap_mapq) : none;
    }
}

extern "C" int coral_bp_pair_table(int32_t n_reads, int32_t n_rows, const int32_t *off, const int32_t *rows,
                                   const int32_t *chr_rank, int32_t n_tid, int32_t min_bp_match_cutoff, int32_t min_mapq,
                                   int32_t gap_, int32_t gap_mapq, int32_t *pairs, void *stream) {
    if (n_reads < 0 || n_rows < 0 || n_tid < 0) return set_err(CORAL_ERR_ARG, "bp_pair_table: negative size");
    if (n_reads == 0 || n_rows == 0) return CORAL_OK;
    if (!off || !rows || !chr_rank || !pairs) return set_err(CORAL_ERR_ARG, "bp_pair_table: null argument");
    if (((uintptr_t)pairs) & 15u) return set_err(CORAL_ERR_ARG, "bp_pair_table: pairs must be 16-byte aligned");
    hipLaunchKernelGGL(k_bp_pairs, dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)n_reads, off, rows, chr_rank,
                       (int)n_tid, (int)min_bp_match_cutoff, (int)min_mapq, (int)gap_, (int)gap_mapq, reinterpret_cast<PairRow *>(pairs));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "bp_pair_table launch");
    return CORAL_OK;
}
