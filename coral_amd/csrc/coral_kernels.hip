// coral_kernels.hip — gfx950 (MI355X, CDNA4) kernels behind include/coral_hip.h.
//
// All of this is integer / indexing work bounded by HBM bandwidth (no MFMA): 64-lane waves stream
// 16-byte CIGAR quads (1 KiB per wave-instruction), wave-level scans give every op its reference
// offset, ballot + popcount compaction emits the rare candidates, and integer atomics (order-free,
// hence bit-exact) reduce per-segment sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

static thread_local char g_err[512] = "";

static int set_err(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

static int hip_err(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return CORAL_ERR_HIP;
}

extern "C" const char *coral_version(void) { return "coral_hip 0.1 (gfx950)"; }
extern "C" const char *coral_last_error(void) { return g_err; }

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

__device__ __forceinline__ int wave_reduce_add(int x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

__device__ __forceinline__ long long wave_reduce_add64(long long x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

__device__ __forceinline__ int wave_reduce_min(int x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x = min(x, __shfl_xor(x, d));
    return x;
}

// ---------------------------------------------------------------------------------------------
// K1  cigar_scan — one wave per record, 256 ops (1 KiB) per wave-iteration, next chunk prefetched
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SCAN_BLOCK) void k_cigar_scan(
    long long n_rec, const int32_t *__restrict__ pos, const int32_t *__restrict__ flagmq,
    const int32_t *__restrict__ n_cigar, const int64_t *__restrict__ cigar_off,
    const uint32_t *__restrict__ cigar, int min_gap, int min_mapq, int32_t *__restrict__ mbases,
    int32_t *__restrict__ qinfer, int32_t *__restrict__ blk_first, int32_t *__restrict__ blk_last,
    int32_t *__restrict__ gaps, uint32_t *__restrict__ gap_count, uint32_t gap_cap) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (SCAN_BLOCK / WAVE) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (SCAN_BLOCK / WAVE);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const cquad_t pad = {OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD};

    for (long long r = wave; r < n_rec; r += nwaves) {
        const int n = n_cigar[r];
        const int nq = (n + 3) >> 2;
        const cquad_t *__restrict__ q = reinterpret_cast<const cquad_t *>(cigar + cigar_off[r]);
        const int p0 = pos[r];
        const bool gaps_on = ((flagmq[r] >> 16) & 0xff) >= min_mapq;

        int carry_ref = 0;   // reference bases consumed by earlier chunks
        int carry_end = 0;   // end (relative, >= 1) of the last aligned block seen so far; 0 = none
        int msum = 0, qsum = 0, first = 0x7fffffff;

        cquad_t cur = pad;
        if (lane < nq) cur = q[lane];
        for (int c = 0; c < nq; c += WAVE) {
            cquad_t nxt = pad;
            if (c + WAVE + lane < nq) nxt = q[c + WAVE + lane];   // prefetch the next KiB

            const uint32_t v[4] = {cur.x, cur.y, cur.z, cur.w};
            int len[4], adv[4];
            bool aln[4];
            int tot = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t op = v[k] & 15u;
                len[k] = (int)(v[k] >> 4);
                adv[k] = ((MASK_REF >> op) & 1u) ? len[k] : 0;
                aln[k] = (MASK_ALN >> op) & 1u;
                qsum += ((MASK_QRY >> op) & 1u) ? len[k] : 0;
                tot += adv[k];
            }
            const int incl = wave_incl_scan_add(tot, lane);
            int ref = carry_ref + incl - tot;      // reference offset of this lane's first op
            // end of the last aligned block inside this lane (0 = none)
            int lane_end = 0;
            {
                int rr = ref;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (aln[k]) lane_end = rr + len[k];
                    rr += adv[k];
                }
            }
            // previous aligned block end as seen by this lane's first aligned op
            const unsigned long long has = __ballot(lane_end != 0);
            const unsigned long long lower = has & below;
            const int src = lower ? (63 - __clzll(lower)) : 0;
            const int from_lane = __shfl(lane_end, src);
            int prev = lower ? from_lane : carry_end;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (aln[k]) {
                    if (prev > 0 && gaps_on && ref - prev > min_gap) {
                        const uint32_t slot = atomicAdd(gap_count, 1u);
                        if (slot < gap_cap) {
                            int4 row = make_int4((int)r, (c + lane) * 4 + k, p0 + prev, p0 + ref);
                            reinterpret_cast<int4 *>(gaps)[slot] = row;
                        }
                    }
                    first = min(first, ref);
                    prev = ref + len[k];
                    msum += len[k];
                }
                ref += adv[k];
            }
            carry_ref += __shfl(incl, 63);
            carry_end = __shfl(prev, 63);
            cur = nxt;
        }
        msum = wave_reduce_add(msum);
        qsum = wave_reduce_add(qsum);
        first = wave_reduce_min(first);
        if (lane == 0) {
            mbases[r] = msum;
            qinfer[r] = qsum;
            blk_first[r] = (carry_end > 0) ? p0 + first : -1;
            blk_last[r] = (carry_end > 0) ? p0 + carry_end : -1;
        }
    }
}

__device__ __forceinline__ cquad_t pad_or(const cquad_t *__restrict__ q, long long i, long long n) {
    cquad_t v = {0u, 0u, 0u, 0u};
    if (i < n) v = q[i];
    return v;
}

// ---------------------------------------------------------------------------------------------
// K1  k_cigar_scan_v2 — the production scan kernel (default instantiation <8, false, true, 8>: 8 KiB per wave in flight,
// conservative gap filter, tiles of 8 consecutive records per wave).  Same contract as k_cigar_scan above, restructured
// for the memory system:
//   * the (record, chunk) sequence of a wave is flattened and a whole batch of chunks is loaded ahead of the arithmetic,
//     across record boundaries, so a wave never idles on the metadata -> first-quad load chain at the start of a record;
//   * record metadata is wave-uniform and fetched with scalar loads, one record ahead of its use;
//   * the wave scan / reductions use DPP row shifts + row broadcasts (6 VALU ops, no LDS crossbar), and
//     wave-uniform values are taken with v_readlane instead of a shuffle;
//   * FILTER / TILE / FULLCHUNK: see the template below and profiles/r01_scan_variants.md.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_zero(int src) {
    return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, 0xf, true);
}

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

// BATCH = chunks (KiB) a wave loads back to back: BATCH KiB per wave in flight while the previous batch is processed
// FILTER = true (variant 7): a chunk first passes a cheap, conservative "can this chunk hold a large gap at all?" test and
// only the chunks that fail it run the exact gap search (max-scan over the wave + per-op distance tests).  A gap is the
// sum of the reference-advancing NON-aligned ops (D, N) between two consecutive aligned ops.  With G(l) = that sum over
// the four ops of lane l, a lane is flagged when it holds real ops but no aligned op, or when G(l) > min_gap / 2.  If no
// lane of a chunk is flagged and the last lane of the previous chunk of the record was not flagged either, then two
// consecutive aligned ops are at most one lane apart (a lane in between would have no aligned op), so every gap that ends
// in this chunk is <= G(l) + G(l + 1) <= min_gap: nothing to report, and the running "end of the last aligned block" is
// simply the one of the highest lane holding an aligned op.  Exact for every input; CIGARs of real reads trip the filter
// only around actual large deletions and at the (padded) end of a record.
// TILE > 1: a wave takes TILE consecutive records at a time (tiles interleaved over the waves) instead of every
// nwaves-th record, so its loads and its scalar metadata reads walk contiguous memory.
template <int BATCH, bool LIGHT = false, bool FILTER = false, int TILE = 1, bool FULLCHUNK = false>
__global__ __launch_bounds__(SCAN_BLOCK) void k_cigar_scan_v2(
    long long n_rec, const int32_t *__restrict__ pos, const int32_t *__restrict__ flagmq,
    const int32_t *__restrict__ n_cigar, const int64_t *__restrict__ cigar_off,
    const uint32_t *__restrict__ cigar, int min_gap, int min_mapq, int32_t *__restrict__ mbases,
    int32_t *__restrict__ qinfer, int32_t *__restrict__ blk_first, int32_t *__restrict__ blk_last,
    int32_t *__restrict__ gaps, uint32_t *__restrict__ gap_count, uint32_t gap_cap) {
    const int lane = threadIdx.x & 63;
    const long long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (SCAN_BLOCK / WAVE) + (threadIdx.x >> 6)));
    const long long nwaves = (long long)gridDim.x * (SCAN_BLOCK / WAVE);
    const cquad_t pad = {OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD};
    // per-op classes, 4 bits per BAM op code: bit0 advances the reference (M D N = X), bit1 aligned block (M = X),
    // bit2 counts towards infer_read_length (M I S H = X); op 15 (padding) and the unused codes are 0.
    const unsigned long long OPCLASS = 0x0000000770441147ull;

    // fetch cursor: (record, first chunk of the batch); runs one batch ahead of the arithmetic, across records.
    // Record metadata is wave-uniform (scalar loads) and requested ONE RECORD AHEAD of its use, so the wave never
    // stalls on the metadata -> first-quad dependency when it moves to its next record.
    auto next_rec = [&](long long r) -> long long {      // the record this wave handles after record r
        if (TILE == 1) return r + nwaves;
        return ((r + 1) % TILE != 0) ? r + 1 : r + 1 + (nwaves - 1) * TILE;
    };
    long long fr = wave * TILE;
    int fc = 0, fnq = 0;
    const cquad_t *__restrict__ fq = reinterpret_cast<const cquad_t *>(cigar);
    int f_nn = 0;            // n_cigar / cigar_off of record fr + nwaves (requested earlier)
    long long f_noff = 0;
    const long long last_rec = n_rec - 1;      // n_rec >= 1 (checked by the launcher)
    auto f_request = [&](long long r) {        // unconditional scalar loads (clamped): no select on the loaded value
        const long long rr = r < n_rec ? r : last_rec;
        f_nn = n_cigar[rr];
        f_noff = cigar_off[rr];
    };
    auto f_meta = [&]() {      // switch to record fr using the values requested earlier, request the one after
        fc = 0;
        fnq = fr < n_rec ? (f_nn + 3) >> 2 : 0;
        fq = reinterpret_cast<const cquad_t *>(cigar + f_noff);
        f_request(next_rec(fr));
    };
    auto f_fetch = [&](cquad_t (&dst)[BATCH]) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (FULLCHUNK && fc + (j + 1) * WAVE <= fnq) {          // (wave-uniform) whole chunk inside the record:
                const cquad_t *__restrict__ base = fq + fc + j * WAVE;      // SGPR base + lane offset, no masking, no padding fill
                dst[j] = base[lane];
            } else {
                dst[j] = pad;
                if (fc + j * WAVE + lane < fnq) dst[j] = fq[fc + j * WAVE + lane];     // exec-masked dwordx4
            }
        }
    };
    auto f_step = [&]() {
        fc += BATCH * WAVE;
        if (fc >= fnq) {
            fr = next_rec(fr);
            f_meta();
        }
    };
    // process cursor (its metadata is requested one record ahead as well)
    long long pr = wave * TILE;
    int pc = 0, pnq = 0, p0 = 0;
    bool gaps_on = false;
    int p_nn = 0, p_npos = 0, p_nfm = 0;
    auto p_request = [&](long long r) {
        const long long rr = r < n_rec ? r : last_rec;
        p_nn = n_cigar[rr];
        p_npos = pos[rr];
        p_nfm = flagmq[rr];
    };
    auto p_meta = [&]() {
        pc = 0;
        pnq = (p_nn + 3) >> 2;
        p0 = p_npos;
        gaps_on = ((p_nfm >> 16) & 0xff) >= min_mapq;
        p_request(next_rec(pr));
    };
    f_request(fr);
    p_request(pr);
    f_meta();
    p_meta();
    cquad_t cur[BATCH], nxt[BATCH];
    f_fetch(cur);
    f_step();

    int carry_ref = 0, carry_end = 0, msum = 0, qsum = 0, first = 0;
    bool tail_flagged = false;                  // FILTER: last lane of the record's previous chunk was flagged
    const int half_gap = min_gap >> 1;
    while (pr < n_rec) {
        // Touch the current batch: the compiler places its wait for these registers HERE, i.e. before the next
        // batch is issued, so the next 4 KiB stay in flight for the whole of this batch's arithmetic.
#pragma unroll
        for (int j = 0; j < BATCH; ++j) asm volatile("" : "+v"(cur[j]));
        f_fetch(nxt);
        f_step();

        if (LIGHT) {      // diagnostic build: same cursors and loads, no arithmetic (isolates the access pattern)
#pragma unroll
            for (int j = 0; j < BATCH; ++j) msum += (int)(cur[j][0] ^ cur[j][1] ^ cur[j][2] ^ cur[j][3]);
        } else
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (j > 0 && pc + j * WAVE >= pnq) break;        // wave-uniform
            const cquad_t b0 = cur[j];
            // ---- branch-free decode of the lane's four ops
            int len[4], adv[4], aend[4], ref[4];
            int fal[4];                          // 0 / -1 per op, "is an aligned block"
            int tot = 0, asum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = b0[k];
                len[k] = (int)(w >> 4);
                if (FILTER) {
                    // v_bfe_i32 takes its bit offset from the low 5 bits of the operand: op code + 16 * (length & 1);
                    // the class masks are replicated into both halves, so the op word itself is the offset.
                    const int fr_ = __builtin_amdgcn_sbfe((int)(MASK_REF * 0x10001u), w, 1u);
                    const int fa_ = __builtin_amdgcn_sbfe((int)(MASK_ALN * 0x10001u), w, 1u);
                    const int fq_ = __builtin_amdgcn_sbfe((int)(MASK_QRY * 0x10001u), w, 1u);
                    adv[k] = len[k] & fr_;
                    aend[k] = len[k] & fa_;
                    fal[k] = fa_;
                    qsum += len[k] & fq_;
                } else {
                    const uint32_t f = (uint32_t)(OPCLASS >> ((w << 2) & 60u));
                    adv[k] = len[k] & -(int)(f & 1u);
                    aend[k] = len[k] & -(int)((f >> 1) & 1u);
                    fal[k] = -(int)((f >> 1) & 1u);
                    qsum += len[k] & -(int)((f >> 2) & 1u);
                }
                asum += aend[k];
                tot += adv[k];
            }
            msum += asum;
            const int incl = wave_incl_scan_add_dpp(tot);
            ref[0] = carry_ref + incl - tot;
            bool exact = true;
            if (FILTER) {
                const int m3 = fal[3], m2 = m3 | fal[2], m1 = m2 | fal[1], m0 = m1 | fal[0];
                const unsigned long long flagged = (__ballot(m0 == 0) & __ballot(b0[0] != OP_PAD_QUAD)) | __ballot(tot - asum > half_gap);
                exact = flagged != 0ull || tail_flagged;
                tail_flagged = (flagged >> 63) != 0ull;
                if (!exact) {
                    const unsigned long long has = __ballot(m0 != 0);   // 0 only for a record without any op (all padding)
                    if (has != 0ull) {
                        if (carry_end == 0) {        // (wave-uniform) the record's first block starts in this chunk:
                            // reference-advancing ops in front of the lane's first aligned op (masks only, no selects)
                            const int n0 = ~fal[0], n1 = n0 & ~fal[1], n2 = n1 & ~fal[2];
                            const int lf = ref[0] + (adv[0] & n0) + (adv[1] & n1) + (adv[2] & n2);
                            first = __builtin_amdgcn_readlane(lf, (int)__builtin_ctzll(has));
                        }
                        // end of the last aligned block of the lane, relative to the lane's first op
                        const int off_end = adv[0] + (adv[1] & m1) + (adv[2] & m2) + (adv[3] & m3);
                        carry_end = __builtin_amdgcn_readlane(ref[0] + off_end, 63 - (int)__builtin_clzll(has));
                    }
                }
            }
            if (exact) {
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
            const int shifted = __builtin_amdgcn_update_dpp(carry_end, mx, 0x138, 0xf, 0xf, false);   // wave_shr:1
            const int prev_in = max(shifted, carry_end);
            if (carry_end == 0) {                    // (wave-uniform) still looking for the record's first block
                const unsigned long long has = __ballot(run[3] != 0);
                if (has != 0ull) {
                    int lf = ref[3];
                    lf = aln[2] ? ref[2] : lf;
                    lf = aln[1] ? ref[1] : lf;
                    lf = aln[0] ? ref[0] : lf;
                    first = __builtin_amdgcn_readlane(lf, (int)__builtin_ctzll(has));
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
                            int4 row = make_int4((int)pr, (pc + j * WAVE + lane) * 4 + k, p0 + pv[k], p0 + ref[k]);
                            reinterpret_cast<int4 *>(gaps)[slot] = row;
                        }
                    }
                }
            }
            carry_end = max(carry_end, __builtin_amdgcn_readlane(mx, 63));
            }
            carry_ref += __builtin_amdgcn_readlane(incl, 63);
        }

        if (pc + BATCH * WAVE >= pnq) {          // last batch of this record: write its summary, move on
            const int ms = wave_sum_dpp(msum);
            const int qs = wave_sum_dpp(qsum);
            if (lane == 0) {
                mbases[pr] = ms;
                qinfer[pr] = qs;
                blk_first[pr] = (carry_end > 0) ? p0 + first : -1;
                blk_last[pr] = (carry_end > 0) ? p0 + carry_end : -1;
            }
            carry_ref = 0; carry_end = 0; msum = 0; qsum = 0; first = 0;
            tail_flagged = false;
            pr = next_rec(pr);
            p_meta();
        } else {
            pc += BATCH * WAVE;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) cur[j] = nxt[j];
    }
}

// The alternative kernel structures that were measured and rejected (flat, packed, ring, ring + counted waits, tile stream)
#include "coral_scan_experiments.hip.inc"

// Streaming-read probe: what a plain grid-stride 16-byte-per-lane read of the same CIGAR bytes achieves
// (upper bound for any kernel that must touch every op once).
__global__ __launch_bounds__(256) void k_stream_probe(const cquad_t *__restrict__ q, long long n_quads, uint32_t *__restrict__ out) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    uint32_t acc = 0;
    for (long long i = (long long)blockIdx.x * 256 * 4 + threadIdx.x; i < n_quads; i += stride) {
        cquad_t a = q[i], b = pad_or(q, i + 256, n_quads), c = pad_or(q, i + 512, n_quads), d = pad_or(q, i + 768, n_quads);
        acc += a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    if (acc == 0x9e3779b9u) out[0] = acc;      // never true in practice; keeps the loads alive
}

// Probe 2: every wave streams its own contiguous region (n_quads / n_waves quads), 4 KiB per iteration.
__global__ __launch_bounds__(256) void k_stream_probe_regions(const cquad_t *__restrict__ q, long long n_quads, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    const long long per = (n_quads + nwaves - 1) / nwaves;
    const long long a = wave * per, b = (a + per < n_quads) ? a + per : n_quads;
    uint32_t acc = 0;
    for (long long i = a + lane; i < b; i += 256) {
        cquad_t x0 = q[i], x1 = pad_or(q, i + 64, b), x2 = pad_or(q, i + 128, b), x3 = pad_or(q, i + 192, b);
        acc += x0[0] ^ x0[1] ^ x0[2] ^ x0[3] ^ x1[0] ^ x1[1] ^ x1[2] ^ x1[3] ^ x2[0] ^ x2[1] ^ x2[2] ^ x2[3] ^ x3[0] ^ x3[1] ^ x3[2] ^ x3[3];
    }
    if (acc == 0x9e3779b9u) out[0] = acc;
}

static int g_probe_mode = 1;
extern "C" int coral_set_probe_mode(int m) { g_probe_mode = m; return CORAL_OK; }

extern "C" int coral_time_stream_read(const uint32_t *cigar, long long n_words, uint32_t *scratch, int iters, float *ms, void *stream) {
    if (!cigar || !scratch || !ms || iters < 1) return set_err(CORAL_ERR_ARG, "time_stream_read: bad arguments");
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipStream_t s = (hipStream_t)stream;
    (void)hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) {
        if (g_probe_mode == 1)
            hipLaunchKernelGGL(k_stream_probe, dim3(2048), dim3(256), 0, s, reinterpret_cast<const cquad_t *>(cigar), n_words / 4, scratch);
        else
            hipLaunchKernelGGL(k_stream_probe_regions, dim3(2048), dim3(256), 0, s, reinterpret_cast<const cquad_t *>(cigar), n_words / 4, scratch);
    }
    (void)hipEventRecord(b, s);
    hipError_t e = hipEventSynchronize(b);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) return hip_err(e, "time_stream_read");
    *ms = t / (float)iters;
    return CORAL_OK;
}

static int g_scan_variant = 15;  // 8 KiB per wave in flight + conservative gap filter + tiles of 8 consecutive records per wave: best launch time (profiles/r01_scan_variants.md)
extern "C" int coral_set_scan_variant(int v) {
    if (v < 1 || v > 27) return CORAL_ERR_ARG;
    g_scan_variant = v;
    return CORAL_OK;
}

static int scan_grid(long long n_rec) {
    long long blocks = (n_rec + (SCAN_BLOCK / WAVE) - 1) / (SCAN_BLOCK / WAVE);
    const long long cap = 256 * 8;   // 256 CUs x 8 workgroups of 4 waves = 32 waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static int tile_grid(long long n_rec, int tile) {      // one wave per tile up to the resident-wave cap
    const long long tiles = (n_rec + tile - 1) / tile;
    long long blocks = (tiles + (SCAN_BLOCK / WAVE) - 1) / (SCAN_BLOCK / WAVE);
    const long long cap = 256 * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static int check_records(const coral_records_t *rec) {
    if (!rec) return set_err(CORAL_ERR_ARG, "records: null");
    if (rec->n_rec < 0 || rec->n_rec > 0x7fffffffLL) return set_err(CORAL_ERR_ARG, "records: n_rec out of range");
    if (rec->n_rec > 0 && (!rec->tid || !rec->pos || !rec->end || !rec->flagmq || !rec->n_cigar || !rec->cigar_off || !rec->cigar))
        return set_err(CORAL_ERR_ARG, "records: null array");
    if (((uintptr_t)rec->cigar) & 15u) return set_err(CORAL_ERR_ARG, "records: cigar base must be 16-byte aligned");
    return CORAL_OK;
}

extern "C" int coral_cigar_scan(const coral_records_t *rec, int32_t min_gap, int32_t min_mapq,
                                int32_t *mbases, int32_t *qinfer, int32_t *blk_first, int32_t *blk_last,
                                int32_t *gaps, uint32_t *gap_count, uint32_t gap_cap, void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (rec->n_rec == 0) return CORAL_OK;
    if (!mbases || !qinfer || !blk_first || !blk_last || !gap_count || (gap_cap && !gaps))
        return set_err(CORAL_ERR_ARG, "cigar_scan: null output");
    if (((uintptr_t)gaps) & 15u) return set_err(CORAL_ERR_ARG, "cigar_scan: gaps must be 16-byte aligned");
    if (g_scan_variant == 1)
        hipLaunchKernelGGL(k_cigar_scan, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
#define LAUNCH_V2(B)                                                                                                  \
    hipLaunchKernelGGL(k_cigar_scan_v2<B>, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,     \
                       (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,        \
                       (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap)
    else if (g_scan_variant == 2)
        LAUNCH_V2(4);
    else if (g_scan_variant == 4)
        LAUNCH_V2(2);
    else if (g_scan_variant == 6)
        hipLaunchKernelGGL(k_cigar_scan_flat<4>, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 8)
        hipLaunchKernelGGL(k_cigar_scan_packed<8>, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 9)
        hipLaunchKernelGGL(k_cigar_scan_packed<4>, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
#define LAUNCH_RING(B, F)                                                                                             \
    hipLaunchKernelGGL((k_cigar_scan_ring<B, F>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, \
                       (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,        \
                       (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap)
#define LAUNCH_RING_ASM(B, F)                                                                                         \
    hipLaunchKernelGGL((k_cigar_scan_ring_asm<B, F>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, \
                       (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,        \
                       (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap)
    else if (g_scan_variant == 13)
        LAUNCH_RING_ASM(8, true);
    else if (g_scan_variant == 14)
        LAUNCH_RING_ASM(4, true);
    else if (g_scan_variant == 10)
        LAUNCH_RING(8, true);
    else if (g_scan_variant == 11)
        LAUNCH_RING(4, true);
    else if (g_scan_variant == 12)
        LAUNCH_RING(8, false);
#define LAUNCH_TILED(B, L, T)                                                                                         \
    hipLaunchKernelGGL((k_cigar_scan_v2<B, L, true, T>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, \
                       (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,        \
                       (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap)
#define LAUNCH_TILE(B, T)                                                                                             \
    hipLaunchKernelGGL((k_cigar_scan_tile<B, T, true>), dim3(tile_grid(rec->n_rec, T)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, \
                       (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,        \
                       (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap)
    else if (g_scan_variant == 20)
        LAUNCH_TILE(8, 8);
    else if (g_scan_variant == 21)
        LAUNCH_TILE(8, 16);
    else if (g_scan_variant == 22)
        LAUNCH_TILE(4, 8);
    else if (g_scan_variant == 23)
        hipLaunchKernelGGL((k_cigar_scan_v2<8, false, true, 8, true>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 24)
        hipLaunchKernelGGL((k_cigar_scan_v2<4, true, true, 8, true>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 25)
        LAUNCH_TILED(4, false, 8);
    else if (g_scan_variant == 26)
        LAUNCH_TILED(2, false, 8);
    else if (g_scan_variant == 27)
        LAUNCH_TILED(16, false, 8);
    else if (g_scan_variant == 17)
        LAUNCH_TILED(8, false, 4);
    else if (g_scan_variant == 18)
        LAUNCH_TILED(8, false, 16);
    else if (g_scan_variant == 19)
        LAUNCH_TILED(4, true, 8);            // diagnostic: loads and cursors only, tiled (outputs invalid)
    else if (g_scan_variant == 15)
        hipLaunchKernelGGL((k_cigar_scan_v2<8, false, true, 8>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 16)
        hipLaunchKernelGGL((k_cigar_scan_v2<8, false, true, 32>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 7)
        hipLaunchKernelGGL((k_cigar_scan_v2<8, false, true>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else if (g_scan_variant == 5)
        hipLaunchKernelGGL((k_cigar_scan_v2<4, true>), dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream,
                           (long long)rec->n_rec, rec->pos, rec->flagmq, rec->n_cigar, rec->cigar_off, rec->cigar,
                           (int)min_gap, (int)min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap);
    else
        LAUNCH_V2(8);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "cigar_scan launch");
    return CORAL_OK;
}

extern "C" int coral_time_cigar_scan(const coral_records_t *rec, int32_t min_gap, int32_t min_mapq,
                                     int32_t *mbases, int32_t *qinfer, int32_t *blk_first, int32_t *blk_last,
                                     int32_t *gaps, uint32_t *gap_count, uint32_t gap_cap, int32_t iters,
                                     float *ms_per_launch, void *stream) {
    if (!ms_per_launch || iters < 1) return set_err(CORAL_ERR_ARG, "time_cigar_scan: bad arguments");
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a);
    if (e != hipSuccess) return hip_err(e, "hipEventCreate");
    e = hipEventCreate(&b);
    if (e != hipSuccess) return hip_err(e, "hipEventCreate");
    hipStream_t s = (hipStream_t)stream;
    int rc = CORAL_OK;
    (void)hipEventRecord(a, s);
    for (int i = 0; i < iters && rc == CORAL_OK; ++i) {
        (void)hipMemsetAsync(gap_count, 0, sizeof(uint32_t), s);
        rc = coral_cigar_scan(rec, min_gap, min_mapq, mbases, qinfer, blk_first, blk_last, gaps, gap_count, gap_cap, stream);
    }
    (void)hipEventRecord(b, s);
    e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc) return rc;
    if (e != hipSuccess) return hip_err(e, "time_cigar_scan");
    *ms_per_launch = ms / (float)iters;
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// K2  segment coverage
//   phase A (thread per record): classify against the sorted disjoint segment table; records fully
//            inside one segment add mbases; boundary-straddling records are compacted into a list;
//   phase B (wave per straddler): walk the CIGAR once per overlapped segment.
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
    const int32_t *__restrict__ mbases, const int32_t *__restrict__ qinfer, int n_seg,
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
                const bool counts = qinfer[r] > 0;
                const bool has_seq = ((flagmq[r] >> 24) & 1) && n_cigar[r] > 0;
                if (seg_start[j] <= p && e <= seg_end[j]) {
                    if (use_lds) {
                        if (counts) atomicAdd(&bins[2 * j], 1ull);
                        if (has_seq) atomicAdd(&bins[2 * j + 1], (unsigned long long)mbases[r]);
                    } else {
                        if (counts) atomicAdd(&n_reads[j], 1ull);
                        if (has_seq) atomicAdd(&n_bases[j], (unsigned long long)mbases[r]);
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

__global__ __launch_bounds__(SCAN_BLOCK) void k_seg_walk(
    const uint32_t *__restrict__ strad, const uint32_t *__restrict__ strad_count,
    const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, const int32_t *__restrict__ end,
    const int32_t *__restrict__ n_cigar, const int64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    int n_seg, const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_start,
    const int32_t *__restrict__ seg_end, unsigned long long *__restrict__ n_bases) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (SCAN_BLOCK / WAVE) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (SCAN_BLOCK / WAVE);
    const long long n_strad = *strad_count;
    const cquad_t pad = {OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD, OP_PAD_QUAD};
    for (long long w = wave; w < n_strad; w += nwaves) {
        const long long r = strad[w];
        const int t = tid[r], p0 = pos[r], e0 = end[r];
        const int nq = (n_cigar[r] + 3) >> 2;
        const cquad_t *__restrict__ q = reinterpret_cast<const cquad_t *>(cigar + cigar_off[r]);
        int j = first_seg_ending_after(seg_tid, seg_end, n_seg, t, p0);
        for (; j < n_seg && seg_tid[j] == t && seg_start[j] < e0; ++j) {
            const int s = seg_start[j] - p0, e = seg_end[j] - p0;    // segment in record-relative coordinates
            long long acc = 0;
            int carry_ref = 0;
            for (int c = 0; c < nq && carry_ref < e; c += WAVE) {
                cquad_t cur = pad;
                if (c + lane < nq) cur = q[c + lane];
                const uint32_t v[4] = {cur.x, cur.y, cur.z, cur.w};
                int len[4], adv[4];
                bool aln[4];
                int tot = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t op = v[k] & 15u;
                    len[k] = (int)(v[k] >> 4);
                    adv[k] = ((MASK_REF >> op) & 1u) ? len[k] : 0;
                    aln[k] = (MASK_ALN >> op) & 1u;
                    tot += adv[k];
                }
                const int incl = wave_incl_scan_add(tot, lane);
                int ref = carry_ref + incl - tot;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (aln[k]) {
                        const int ov = min(ref + len[k], e) - max(ref, s);
                        if (ov > 0) acc += ov;
                    }
                    ref += adv[k];
                }
                carry_ref += __shfl(incl, 63);
            }
            acc = wave_reduce_add64(acc);
            if (lane == 0 && acc) atomicAdd(&n_bases[j], (unsigned long long)acc);
        }
    }
}

extern "C" int coral_segment_coverage(const coral_records_t *rec, const int32_t *mbases, const int32_t *qinfer,
                                      int32_t n_seg, const int32_t *seg_tid, const int32_t *seg_start,
                                      const int32_t *seg_end, unsigned long long *n_reads,
                                      unsigned long long *n_bases, uint32_t *strad, uint32_t *strad_count,
                                      void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (n_seg < 0) return set_err(CORAL_ERR_ARG, "segment_coverage: n_seg < 0");
    if (n_seg == 0 || rec->n_rec == 0) return CORAL_OK;
    if (!mbases || !qinfer || !seg_tid || !seg_start || !seg_end || !n_reads || !n_bases || !strad || !strad_count)
        return set_err(CORAL_ERR_ARG, "segment_coverage: null argument");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (rec->n_rec + COV_BLOCK - 1) / COV_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_seg_classify, dim3((int)blocks), dim3(COV_BLOCK), 0, s, (long long)rec->n_rec, rec->tid,
                       rec->pos, rec->end, rec->flagmq, rec->n_cigar, mbases, qinfer, (int)n_seg, seg_tid, seg_start,
                       seg_end, n_reads, n_bases, strad, strad_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "seg_classify launch");
    hipLaunchKernelGGL(k_seg_walk, dim3(scan_grid(rec->n_rec)), dim3(SCAN_BLOCK), 0, s, strad, strad_count, rec->tid,
                       rec->pos, rec->end, rec->n_cigar, rec->cigar_off, rec->cigar, (int)n_seg, seg_tid, seg_start,
                       seg_end, n_bases);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_err(e, "seg_walk launch");
    return CORAL_OK;
}

// ---------------------------------------------------------------------------------------------
// K3  point cover — thread per record, sorted query points
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COV_BLOCK) void k_point_cover(
    long long n_rec, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
    const int32_t *__restrict__ end, int n_pts, const int32_t *__restrict__ pt_tid,
    const int32_t *__restrict__ pt_pos, unsigned long long *__restrict__ pairs,
    uint32_t *__restrict__ pair_count, uint32_t pair_cap) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const long long stride = (long long)gridDim.x * COV_BLOCK;
    const long long n_round = (n_rec + stride - 1) / stride * stride;      // whole waves stay in the loop (ballots)
    for (long long r = (long long)blockIdx.x * COV_BLOCK + threadIdx.x; r < n_round; r += stride) {
        int j = n_pts, t = -1, e = 0;
        if (r < n_rec) {
            t = tid[r];
            const int p = pos[r];
            e = end[r];
            if (t >= 0) {        // first point with (pt_tid, pt_pos) >= (t, p)
                int lo = 0, hi = n_pts;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const int pt = pt_tid[mid];
                    const bool ge = (pt > t) || (pt == t && pt_pos[mid] >= p);
                    if (ge) hi = mid; else lo = mid + 1;
                }
                j = lo;
            }
        }
        // emit one covered point per round: ballot + popcount gives each lane its slot, one atomic per wave
        for (;;) {
            const bool hit = j < n_pts && pt_tid[j] == t && pt_pos[j] < e;
            const unsigned long long m = __ballot(hit);
            if (m == 0ull) break;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(pair_count, (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (hit) {
                const uint32_t slot = base + (uint32_t)__popcll(m & below);
                if (slot < pair_cap) pairs[slot] = ((unsigned long long)(uint32_t)j << 32) | (unsigned long long)(uint32_t)r;
                ++j;
            }
        }
    }
}

extern "C" int coral_point_cover(const coral_records_t *rec, int32_t n_pts, const int32_t *pt_tid,
                                 const int32_t *pt_pos, unsigned long long *pairs, uint32_t *pair_count,
                                 uint32_t pair_cap, void *stream) {
    int rc = check_records(rec);
    if (rc) return rc;
    if (n_pts < 0) return set_err(CORAL_ERR_ARG, "point_cover: n_pts < 0");
    if (n_pts == 0 || rec->n_rec == 0) return CORAL_OK;
    if (!pt_tid || !pt_pos || !pair_count || (pair_cap && !pairs)) return set_err(CORAL_ERR_ARG, "point_cover: null argument");
    long long blocks = (rec->n_rec + COV_BLOCK - 1) / COV_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_point_cover, dim3((int)blocks), dim3(COV_BLOCK), 0, (hipStream_t)stream, (long long)rec->n_rec,
                       rec->tid, rec->pos, rec->end, (int)n_pts, pt_tid, pt_pos, pairs, pair_count, pair_cap);
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
