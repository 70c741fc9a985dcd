// coral_bamgpu.hip — BAM (BGZF) decode ON THE GPU: compressed file bytes go over PCIe, everything else happens in HBM.
//
// Replaces, like coral_bam.cpp, what the reference gets from pysam.AlignmentFile(path, 'rb') + the whole-file fetch()
// (/root/reference/src/infer_breakpoint_graph.py:65, :140-158), and produces the same SoA records as the CPU pipeline
// (tests/test_bam_gpu.py compares the two field by field).  The CPU pipeline tops out at the host's inflate rate (zlib,
// ~0.3 GB/s per core); here the host only moves compressed bytes:
//
//   feeder thread   pread(file) -> ring of small pinned buffers -> BGZF block table of the batch -> H2D          (copy stream)
//   k_bgzf_inflate  one wave per BGZF block: DEFLATE decode (coral_inflate_core.h) into the batch's buffer (inflate stream)
//   k_bgzf_crc      one wave per block: CRC-32 of the inflated bytes against the block's trailer (coral_crc32.h)
//   k_bam_find      one wave per 128 KiB segment: first plausible record start (chained check) + hop along block_size
//   k_bam_verify    one workgroup: follows the chain of segment landings from the batch's KNOWN first record; a segment
//                   whose guess is not on the chain is re-walked exactly, so record boundaries never rest on a heuristic
//   k_bam_starts    record start offsets, in file order
//   k_bam_meta      one wave per record: fixed fields, tag walk (NM, SA, CG), sizes of what the record contributes
//   (hipcub scans)  offsets of CIGAR ops (padded to 4), read-name bytes and SA text
//   k_bam_emit      one wave per record: CIGAR -> padded SoA op array (+ reference / query lengths), SEQ scan for non-ACGT
//                   codes, read name and SA text -> compact blobs for the host
//   worker thread   per batch, a few hundred bytes per record: read names -> ids, SA text -> numeric rows; the rare
//                   records with non-ACGT bases are gathered whole (k_bam_gather) and handled by the CPU pipeline's own routine.
// Batches (64 MiB first, doubling up to 2.52 GiB inflated, coral_bamgpu_open) are double-buffered: while batch k is parsed, batch k + 1 is inflated
// and k + 2 is read.
// A record that straddles two batches is carried in front of the next batch's buffer.
//
// Multi-GPU: rank r of `world` decodes the BGZF blocks that start in its byte range of the file, with the
// same first-record / last-record rule as coral_bam_decode_range, so consecutive ranges neither drop nor repeat a record.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "../../include/coral_hip.h"
#include "coral_bam_common.h"
#include "coral_crc32.h"
#include "coral_inflate_core.h"

using namespace coral_bam;

namespace {

#define WAVE 64
#define SEG_BYTES (128ll << 10)          // granularity of the speculative record-start search
#define CARRY_CAP (64ll << 20)           // room in front of a batch for the head of a straddling record
#define COMP_SLACK 4096                  // readable bytes behind the compressed batch (the input window reads ahead)
#define OVERHANG_BLOCKS 1024             // blocks behind a byte range its last record may straddle into (as coral_bam.cpp)
#define N_STAGE 4                        // pinned staging buffers of the feeder (file -> pinned -> device, round robin)
#define STAGE_BYTES (32ull << 20)
#define FIRST_BATCH (64ull << 20)        // inflated bytes of the first batch; the following ones double up to the cap

struct BlockDesc {
    uint32_t src_off;     // first DEFLATE byte, relative to the compressed batch
    uint32_t src_len;     // DEFLATE bytes
    uint32_t dst_off;     // first output byte, relative to the batch's first inflated byte
    uint32_t isize;
};

// ---------------------------------------------------------------------------------------------
// K_inflate
// ---------------------------------------------------------------------------------------------
#ifndef RING_LOG
#define RING_LOG 11                      // 2 KiB: with the tables 5.75 KiB of LDS per wave -> 27 waves per CU (4 KiB: 19 waves, +14 % time:
#endif                                   // the kernel is bound by the latency of its LDS round trips, profiles/r03_pmc_inflate.md)
#define RING_BYTES (1 << RING_LOG)       // recent output per wave, in LDS: LZ77 matches read it instead of global memory
#define RING_MASK (RING_BYTES - 1)

// Backend of coral_inflate::Inflater for one wave (the symbol loop is Inflater::codes_vector).
// Output positions are counted in "aligned coordinates" A = (out & 255) + o, so that A = 0 is a 256-byte line of global memory:
// ring index = A & RING_MASK, global address = gbase + A.  Every byte goes to the ring first (a literal: one ds_write_b8 of all
// lanes to the same address; a match: ring -> ring, or global -> ring when the source has left the ring); each time the
// position crosses a 256-byte line, the completed lines go to global memory as one dword store per lane.
// Input: two registers of 64 dwords each, used alternately (the bit buffer is refilled with v_readlane, never from memory).
// ABLATE (timing experiments of tools/bench_inflate.py with a -DCORAL_EXPERIMENTS build only; the product instantiates 0): 1 = no stores to global memory, 2 = matches whose
// source left the ring read the ring anyway, 4 = no match copy, 8 = no literal write.
template <int ABLATE>
struct DevWaveT {
    static constexpr bool vector_loop = true;
    int lane;
    float lane_half;            // lane + 0.5
    const char *in_base;        // dword-aligned start of the stream's input window (global memory)
    int stream_off;             // first byte of the DEFLATE stream, relative to in_base (0..3)
    int stream_len;
    int base_dw;                // dword index (from in_base) the Inflater's `dwords` counts from
    int win;                    // dword index of lane 0 of r0
    int last_dw;                // dword index of the stream's last byte
    uint32_t r0, r1;            // lane i: dwords win + i (being consumed) and win + 64 + i (loaded ahead) of the input
    int idx;                    // next dword of r0 (0..63)
    uint8_t *gbase;             // out - (out & 255)
    uint8_t *ring;              // LDS, RING_BYTES
    int a0, aend;               // aligned coordinates of the block's first byte and of the byte behind its last
    int a;                      // next output byte
    int gdone;                  // first byte not yet in global memory
    int attend;                 // the symbol loop calls attention() once `a` has reached this: a 256-byte line is complete (or,
                                // set to a - 1 by the rare paths, `over` was raised)
    uint32_t badv;              // (vector) a distance reached in front of the output
    bool over;                  // more output than the block may have, more input than the stream holds, or fail(): stop
    int err;                    // the first fail() code

    __device__ __forceinline__ uint32_t uni(uint32_t x) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    __device__ __forceinline__ uint32_t vec(uint32_t x) const {      // keeps what is computed from x on the vector pipe
        asm("" : "+v"(x));
        return x;
    }
    __device__ __forceinline__ uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) const { return __builtin_amdgcn_ubfe(x, off, width); }
    __device__ __forceinline__ bool bad() const { return uni(badv) != 0u; }
    __device__ __forceinline__ void fail(int code) {                 // stop the symbol loop at its next round
        if (!err) err = code;
        over = true;
        attend = a - 1;
    }
    __device__ __forceinline__ bool failed() const { return over || err != 0 || bad(); }
    __device__ __forceinline__ int error_code() const { return err ? err : bad() ? (int)coral_inflate::ERR_DISTANCE : (int)coral_inflate::ERR_OVERFLOW; }
    __device__ __forceinline__ bool needs_attention() const { return a >= attend; }
    // Input window loads never start behind the stream's last dword: a corrupt stream that keeps asking for input re-reads the
    // end (and is stopped by the pull limit) instead of walking out of the compressed buffer (readable COMP_SLACK bytes beyond
    // its last stream, more than one 256-byte window).
    __device__ __forceinline__ uint32_t input_load(int dw) const {
        const int d = dw < last_dw ? dw : last_dw;
        return *reinterpret_cast<const uint32_t *>(in_base + (uint32_t)(d * 4 + lane * 4));
    }
    __device__ __forceinline__ void load_window() {
        r0 = input_load(win);
        r1 = input_load(win + WAVE);
        idx = 0;
    }
    // The next 32 input bits: one v_readlane; every 64th call moves on to the register loaded 64 dwords ago and starts the
    // load of its successor.
    __device__ __forceinline__ uint32_t next_dword() {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)r0, idx);
        if (__builtin_expect(++idx == WAVE, 0)) {
            win += WAVE;
            r0 = r1;
            r1 = input_load(win + WAVE);
            idx = 0;
            if (win > last_dw + WAVE) fail(coral_inflate::ERR_INPUT);    // a whole window behind the stream's end: corrupt
        }
        return v;
    }
    __device__ __forceinline__ bool input_exhausted() const { return win + idx > last_dw + 4; }
    __device__ __forceinline__ uint8_t *ring_at(uint32_t av) const { return ring + (av & RING_MASK); }
    // completed 256-byte lines ring -> global memory (the block's first line may start inside a line: byte stores there);
    // never beyond the block's last byte
    // Called by the symbol loop once per round when a >= attend; false = stop decoding.
    __device__ __forceinline__ bool attention() {
        drain();
        return !over;
    }
    __device__ __forceinline__ void drain() {
        if (a > aend) over = true;
        const int upto = (a < aend ? a : aend) & ~255;
        int g = gdone;
        if ((g & 255) && g < upto) {
            const int head = (g | 255) + 1;
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
            for (int p = g + lane; p < head; p += WAVE) gbase[p] = *ring_at((uint32_t)p);
            g = head;
        }
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
        for (; g < upto; g += 256) {
            const int p = g + 4 * lane;
            if (!(ABLATE & 1)) *reinterpret_cast<uint32_t *>(gbase + p) = *reinterpret_cast<const uint32_t *>(ring_at((uint32_t)p));
        }
        gdone = (int)uni((uint32_t)g);
        attend = over ? a - 1 : (gdone | 255) + 1;
    }
    __device__ __forceinline__ void finish() {                        // what is left goes out byte by byte
        drain();
        const int end = a < aend ? a : aend;
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
        for (int p = gdone + lane; p < end; p += WAVE) gbase[p] = *ring_at((uint32_t)p);
        gdone = end;
    }
    __device__ __forceinline__ void lit(uint32_t bytev) {
        if (!(ABLATE & 8)) *ring_at(vec((uint32_t)a)) = (uint8_t)bytev;
        ++a;
    }
    __device__ __forceinline__ void put_literal(uint32_t b) { lit(b); }
    // lenv, distv: the same value in every lane (vector registers)
    __device__ __forceinline__ void match(uint32_t lenv, uint32_t distv) {
        const uint32_t av = vec((uint32_t)a);
        const uint32_t room = vec((uint32_t)(aend - a));              // (a <= aend + 255; as unsigned a huge room then, `over` is raised by drain)
        const uint32_t far_back = distv > av - (uint32_t)a0 ? 1u : 0u;    // reaches in front of the block's output: corrupt
        badv |= far_back;
        lenv = far_back ? 0u : (lenv < room ? lenv : room);
        const int len = (int)uni(lenv);
        const uint32_t dist = uni(distv);
        if (ABLATE & 4) {
            a += len;
            return;
        }
        if (__builtin_expect(dist <= RING_BYTES - 64 || (ABLATE & 2), 1)) {
            // byte k of the match = byte (k mod dist) of the dist bytes in front of it (k mod dist = k when dist >= 64 > lane)
            const float rd = __builtin_amdgcn_rcpf((float)distv);
            auto copy = [&](uint32_t k, float kh) {
                const uint32_t q = (uint32_t)(kh * rd);               // floor((k + 0.5) / dist): exact, (k + 0.5) / dist is >= 0.5 / dist off an integer
                const uint32_t km = distv < 64u ? k - q * distv : k;
                if (k < lenv) *ring_at(av + k) = *ring_at(av - distv + km);
            };
            copy((uint32_t)lane, lane_half);
            if (__builtin_expect(len > WAVE, 0)) {                    // chunks in order: LDS operations of a wave execute in order
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
                for (int k = lane + WAVE; k < len; k += WAVE) copy((uint32_t)k, (float)k + 0.5f);
            }
        } else {
            // the source left the ring: it is in global memory (drained up to the last line boundary; a - dist + len lies below it)
            const uint8_t *src = gbase + (a - (int)dist);
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
            for (int k = lane; k < len; k += WAVE) *ring_at(av + (uint32_t)k) = src[k];
        }
        a += len;
    }
    // The hot path of the symbol loop, by hand (gfx950 ISA).  The compiler turns Inflater::codes_vector into a state machine of
    // boolean SGPR pairs and register copies at every join (≈42 scalar + 13 branch + 35 vector instructions per symbol); this loop
    // needs ≈20 + 7 + 24 (rocprofv3 counters): a literal is 9 scalar + 8 vector instructions behind its table
    // look-up, a short match ≈20 + 25.  It takes
    // what is common — literal / length / distance codes found in the LDS tables, refills from the current input register,
    // every match whose distance is known to lie inside the block (from the ring, or from global memory once the source has left
    // the ring) — and RETURNS, with nothing half-done, where it cannot go on:
    //   0  at a symbol boundary: the block's last 260 bytes begin, its first (partial) line is complete or fail() asked for a stop
    //      (completed whole lines are written to global memory inside the loop), the refill would switch the input register, or
    //      the code is not in the table (long code, end of block)
    //   1  a length has been read (lenv) and the input register must be switched, or the distance code is not in the table
    //   2  length and distance have been read (lenv, distv), but the distance reaches further back than this call's first byte
    //      was from the block's start (`span` is taken at entry: a distance the general path accepts may come back here, never
    //      the other way round)
    // and Inflater::codes_vector does that one step.  Everything is wave-uniform; the lanes only differ in the copy.  The bit
    // buffer lives in s[70:71] inside (its low half is needed on its own), s[72:73] hold a refill.
    // Hazards of gfx940-class hardware that the assembler does not pad inside inline assembly: a VALU result read by
    // v_readfirstlane needs one wait state, a transcendental result (v_rcp_f32) read by another VALU instruction one.
    static constexpr bool has_fast = (ABLATE == 0);
    __device__ __forceinline__ int fast(uint64_t &bb, int &bc, long long &dwords, const coral_inflate::Tables *T, uint32_t &lenv, uint32_t &distv) {
        int lim = attend < aend - 260 ? attend : aend - 260;
        const uint32_t span = (uint32_t)(a - a0);
        const uint32_t dlim = span < (uint32_t)(RING_BYTES - 64) ? span : (uint32_t)(RING_BYTES - 64);
        const uint32_t ring_lds = (uint32_t)(uintptr_t)ring, ll_lds = (uint32_t)(uintptr_t)T->ll, dt_lds = (uint32_t)(uintptr_t)T->dt;
        int state;
        const int idx_in = idx;
        uint32_t se, sn, sx, slen;
        uint32_t vb, ve, vt, vn, vxb, vx, vq, vs, out_len, out_dist;
        // (ring addresses: the ring is aligned to its size, so (x & RING_MASK) | ring is one v_and_or_b32 with the mask in a VGPR)
#define CORAL_INFL_REFILL                                           \
            "v_readlane_b32 s72, %[r0], %[idx]\n"                   \
            "s_mov_b32 s73, 0\n"                                    \
            "s_lshl_b64 s[72:73], s[72:73], %[bc]\n"                \
            "s_or_b64 s[70:71], s[70:71], s[72:73]\n"               \
            "s_add_i32 %[bc], %[bc], 32\n"                          \
            "s_add_i32 %[idx], %[idx], 1\n"
#define CORAL_INFL_LOOKUP                                           \
            "v_mov_b32 %[vb], s70\n"                                \
            "v_and_b32 %[vt], 0x3ff, %[vb]\n"                       \
            "v_lshl_add_u32 %[vt], %[vt], 1, %[llb]\n"              \
            "ds_read_u16 %[ve], %[vt]\n"                            \
            "s_waitcnt lgkmcnt(0)\n"                                \
            "v_readfirstlane_b32 %[se], %[ve]\n"                    \
            "s_bitcmp1_b32 %[se], 4\n"
#define CORAL_INFL_LITERAL                                          \
            "s_and_b32 %[sn], %[se], 15\n"                          \
            "v_mov_b32 %[vt], %[a]\n"                               \
            "s_lshr_b64 s[70:71], s[70:71], %[sn]\n"                \
            "v_lshrrev_b32 %[vx], 8, %[ve]\n"                       \
            "s_sub_i32 %[bc], %[bc], %[sn]\n"                       \
            "v_and_or_b32 %[vt], %[vt], %[vmask], %[ring]\n"        \
            "s_mov_b64 exec, 1\n"                                   \
            "ds_write_b8 %[vt], %[vx]\n"                            \
            "s_mov_b64 exec, -1\n"                                  \
            "s_add_i32 %[a], %[a], 1\n"
        asm volatile(
            "s_mov_b64 s[70:71], %[bb]\n"
            "v_mov_b32 %[olen], 0\n"
            "v_mov_b32 %[odist], 0\n"
            "Ltop_%=:\n"
            "s_cmp_ge_i32 %[a], %[lim]\n"
            "s_cbranch_scc1 Lattn_%=\n"
            "Lbits_%=:\n"
            "s_cmp_gt_i32 %[bc], 32\n"
            "s_cbranch_scc1 Lsym_%=\n"
            "s_cmp_eq_u32 %[idx], 63\n"
            "s_cbranch_scc1 Lexit0_%=\n"
            CORAL_INFL_REFILL
            "Lsym_%=:\n"                                       // more than 32 bits: first table look-up of the round
            CORAL_INFL_LOOKUP
            "s_cbranch_scc1 Lnotlit_%=\n"
            CORAL_INFL_LITERAL                                 // (one lane stores: 64 lanes storing to one address are 64 LDS accesses)
            CORAL_INFL_LOOKUP                                  // at least 23 bits left: second look-up without a refill check
            "s_cbranch_scc1 Lnotlit_%=\n"
            CORAL_INFL_LITERAL
            "s_branch Ltop_%=\n"
            "Lnotlit_%=:\n"                                    // ve / se = table entry, vb = the bits it was looked up with (>= 23)
            "s_and_b32 %[sn], %[se], 15\n"
            "v_and_b32 %[vn], 15, %[ve]\n"
            "v_bfe_u32 %[vxb], %[ve], 5, 3\n"                  // extra bits of the length code
            "s_cmp_eq_u32 %[sn], 0\n"
            "v_bfe_u32 %[vt], %[vb], %[vn], %[vxb]\n"
            "v_lshrrev_b32 %[vx], 8, %[ve]\n"                  // length base - 3
            "s_cbranch_scc1 Lexit0_%=\n"                       // not in the table (nothing consumed)
            "v_add_u32 %[vn], %[vn], %[vxb]\n"
            "v_add3_u32 %[olen], %[vx], %[vt], 3\n"
            "v_readfirstlane_b32 %[sn], %[vn]\n"               // (one instruction between the VALU write and this read)
            "s_lshr_b64 s[70:71], s[70:71], %[sn]\n"
            "s_sub_i32 %[bc], %[bc], %[sn]\n"
            "v_readfirstlane_b32 %[slen], %[olen]\n"
            "s_cmp_gt_i32 %[bc], 32\n"                         // the distance code needs up to 8 + 13 bits
            "s_cbranch_scc1 Ldist_%=\n"
            "s_cmp_eq_u32 %[idx], 63\n"
            "s_cbranch_scc1 Lexit1_%=\n"
            CORAL_INFL_REFILL
            "Ldist_%=:\n"
            "v_mov_b32 %[vb], s70\n"
            "v_and_b32 %[vt], 0xff, %[vb]\n"
            "v_lshl_add_u32 %[vt], %[vt], 2, %[dtb]\n"
            "ds_read_b32 %[ve], %[vt]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_and_b32 %[vn], 15, %[ve]\n"
            "v_bfe_u32 %[vxb], %[ve], 8, 4\n"
            "v_lshrrev_b32 %[vx], 16, %[ve]\n"
            "v_bfe_u32 %[vt], %[vb], %[vn], %[vxb]\n"
            "v_add_u32 %[vn], %[vn], %[vxb]\n"
            "v_add_u32 %[odist], %[vx], %[vt]\n"               // a code that is not in the table has entry 0: distance 0, no bits
            "v_readfirstlane_b32 %[sn], %[vn]\n"
            "v_add_u32 %[vt], -1, %[odist]\n"
            "s_lshr_b64 s[70:71], s[70:71], %[sn]\n"
            "s_sub_i32 %[bc], %[bc], %[sn]\n"
            "v_cmp_le_u32 vcc, %[dlim], %[vt]\n"               // dist - 1 >= dlim: not in the table (0), behind the ring, or maybe
            "s_cbranch_vccnz Lfarq_%=\n"                       // in front of the block's first byte
            "v_min_u32 %[vt], 64, %[odist]\n"                  // longer than 64 bytes, or overlapping its own source (dist < len):
            "v_cmp_lt_u32 vcc, %[vt], %[olen]\n"               // the general copy below; otherwise byte k comes from a - dist + k
            "s_cbranch_vccnz Llong_%=\n"
            "v_sub_u32 %[vs], %[a], %[odist]\n"
            "v_add_u32 %[vt], %[a], %[lane]\n"
            "v_add_u32 %[vs], %[vs], %[lane]\n"
            "v_and_or_b32 %[vt], %[vt], %[vmask], %[ring]\n"
            "v_and_or_b32 %[vs], %[vs], %[vmask], %[ring]\n"
            "v_cmp_gt_u32 vcc, %[slen], %[lane]\n"
            "s_mov_b64 exec, vcc\n"
            "ds_read_u8 %[vx], %[vs]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[vt], %[vx]\n"
            "s_mov_b64 exec, -1\n"
            "s_add_i32 %[a], %[a], %[slen]\n"
            "s_branch Ltop_%=\n"
            "Llong_%=:\n"                                      // source in the ring, any length: byte k of the match = byte (k mod dist) of the
            "v_cvt_f32_u32 %[vq], %[odist]\n"                  // dist bytes in front; chunks of 64 in order (a chunk may read what the chunk
                                                               // before it wrote: LDS operations of a wave execute in order)
            "v_rcp_f32 %[vn], %[vq]\n"
            "v_mov_b32 %[vxb], %[lane]\n"                      // k
            "v_mov_b32 %[ve], %[laneh]\n"                      // k + 0.5
            "s_mov_b32 %[sx], 0\n"
            "Lchunk_%=:\n"
            "v_mul_f32 %[vq], %[ve], %[vn]\n"
            "v_cvt_u32_f32 %[vq], %[vq]\n"
            "v_mul_lo_u32 %[vq], %[vq], %[odist]\n"
            "v_sub_u32 %[vq], %[vxb], %[vq]\n"                 // k mod dist (when dist < 64): floor((k + 0.5) / dist) is exact (DevWaveT::match)
            "v_cmp_gt_u32 vcc, 64, %[odist]\n"
            "v_cndmask_b32 %[vq], %[vxb], %[vq], vcc\n"        // dist >= 64: k itself
            "v_sub_u32 %[vs], %[a], %[odist]\n"
            "v_add_u32 %[vs], %[vs], %[vq]\n"
            "v_and_or_b32 %[vs], %[vs], %[vmask], %[ring]\n"
            "v_add_u32 %[vt], %[a], %[vxb]\n"
            "v_and_or_b32 %[vt], %[vt], %[vmask], %[ring]\n"
            "v_cmp_gt_u32 vcc, %[slen], %[vxb]\n"
            "s_mov_b64 exec, vcc\n"
            "ds_read_u8 %[vx], %[vs]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[vt], %[vx]\n"
            "s_mov_b64 exec, -1\n"
            "v_add_u32 %[vxb], 64, %[vxb]\n"
            "v_add_f32 %[ve], 0x42800000, %[ve]\n"             // + 64.0
            "s_add_i32 %[sx], %[sx], 64\n"
            "s_cmp_lt_u32 %[sx], %[slen]\n"
            "s_cbranch_scc1 Lchunk_%=\n"
            "s_add_i32 %[a], %[a], %[slen]\n"
            "s_branch Ltop_%=\n"
            "Lfarq_%=:\n"
            "v_cmp_eq_u32 vcc, 0, %[odist]\n"                  // distance code not in the table (nothing of it consumed)
            "s_cbranch_vccnz Lexit1_%=\n"
            "v_cmp_lt_u32 vcc, %[dspan], %[odist]\n"           // further back than this call's first output byte was from the
            "s_cbranch_vccnz Lexit2_%=\n"                      // block's start: the general path tells a bad distance from a good one
            // the source has left the ring (dist > RING_BYTES - 64): it is in global memory, drained up to the last 256-byte line
            // boundary, and a - dist + len lies below that (DevWaveT::match)
            "v_sub_u32 %[vs], %[a], %[odist]\n"
            "v_add_u32 %[vs], %[vs], %[lane]\n"
            "v_add_u32 %[vt], %[a], %[lane]\n"
            "v_mov_b32 %[vxb], %[lane]\n"
            "s_mov_b32 %[sx], 0\n"
            "Lfchunk_%=:\n"
            "v_and_or_b32 %[vq], %[vt], %[vmask], %[ring]\n"
            "v_cmp_gt_u32 vcc, %[slen], %[vxb]\n"
            "s_mov_b64 exec, vcc\n"
            "global_load_ubyte %[vx], %[vs], %[gb]\n"
            "s_waitcnt vmcnt(0)\n"
            "ds_write_b8 %[vq], %[vx]\n"
            "s_mov_b64 exec, -1\n"
            "v_add_u32 %[vs], 64, %[vs]\n"
            "v_add_u32 %[vt], 64, %[vt]\n"
            "v_add_u32 %[vxb], 64, %[vxb]\n"
            "s_add_i32 %[sx], %[sx], 64\n"
            "s_cmp_lt_u32 %[sx], %[slen]\n"
            "s_cbranch_scc1 Lfchunk_%=\n"
            "s_add_i32 %[a], %[a], %[slen]\n"
            "s_branch Ltop_%=\n"
            "Lattn_%=:\n"                                      // a >= lim.  Completed 256-byte lines go ring -> global memory right here
            "s_cmp_lt_i32 %[a], %[attend]\n"                  // (one dword per lane, DevWaveT::drain) unless the block's last bytes
            "s_cbranch_scc1 Lexit0_%=\n"                       // begin, the block's first line is a partial one, or fail() moved `attend`
            "s_cmp_gt_i32 %[a], %[aendz]\n"
            "s_cbranch_scc1 Lexit0_%=\n"
            "s_and_b32 %[sn], %[gdone], 255\n"
            "s_cmp_lg_u32 %[sn], 0\n"
            "s_cbranch_scc1 Lexit0_%=\n"
            "s_add_i32 %[sx], %[gdone], 256\n"
            "s_cmp_lg_u32 %[sx], %[attend]\n"
            "s_cbranch_scc1 Lexit0_%=\n"
            "Ldrain_%=:\n"
            "v_lshl_add_u32 %[vt], %[lane], 2, %[gdone]\n"
            "v_and_or_b32 %[vs], %[vt], %[vmask], %[ring]\n"
            "ds_read_b32 %[vx], %[vs]\n"
            "s_mov_b32 %[gdone], %[sx]\n"
            "s_add_i32 %[sx], %[sx], 256\n"
            "s_waitcnt lgkmcnt(0)\n"
            "global_store_dword %[vt], %[vx], %[gb]\n"
            "s_cmp_le_i32 %[sx], %[a]\n"
            "s_cbranch_scc1 Ldrain_%=\n"
            "s_mov_b32 %[attend], %[sx]\n"                     // = (gdone | 255) + 1
            "s_min_i32 %[lim], %[sx], %[aendz]\n"
            "s_cmp_ge_i32 %[a], %[lim]\n"                      // (only when the block's last 260 bytes have begun)
            "s_cbranch_scc1 Lexit0_%=\n"
            "s_branch Lbits_%=\n"
            "Lexit2_%=:\n"
            "s_mov_b32 %[state], 2\n"
            "s_branch Lout_%=\n"
            "Lexit1_%=:\n"
            "s_mov_b32 %[state], 1\n"
            "s_branch Lout_%=\n"
            "Lexit0_%=:\n"
            "s_mov_b32 %[state], 0\n"
            "Lout_%=:\n"
            "s_mov_b64 %[bb], s[70:71]\n"
            : [bb] "+s"(bb), [bc] "+s"(bc), [idx] "+s"(idx), [a] "+s"(a), [gdone] "+s"(gdone), [attend] "+s"(attend), [lim] "+s"(lim), [state] "=&s"(state), [se] "=&s"(se), [sn] "=&s"(sn),
              [sx] "=&s"(sx), [slen] "=&s"(slen), [vb] "=&v"(vb), [ve] "=&v"(ve), [vt] "=&v"(vt), [vn] "=&v"(vn), [vxb] "=&v"(vxb),
              [vx] "=&v"(vx), [vq] "=&v"(vq), [vs] "=&v"(vs), [olen] "=&v"(out_len), [odist] "=&v"(out_dist)
            : [aendz] "s"(aend - 260), [dlim] "s"(dlim), [dspan] "s"(span), [gb] "s"(gbase), [ring] "s"(ring_lds), [llb] "s"(ll_lds), [dtb] "s"(dt_lds),
              [r0] "v"(r0), [lane] "v"(lane), [laneh] "v"(lane_half), [vmask] "v"((uint32_t)RING_MASK)
            : "s70", "s71", "s72", "s73", "vcc", "scc", "memory");
#undef CORAL_INFL_REFILL
#undef CORAL_INFL_LOOKUP
#undef CORAL_INFL_LITERAL
        dwords += idx - idx_in;                             // (the input register is only switched by the general path)
        lenv = out_len;
        distv = out_dist;
        return state;
    }
    __device__ __forceinline__ bool copy_match(int len, int dist) {   // (the plain loop's interface; unused on the device)
        match((uint32_t)len, (uint32_t)dist);
        return true;
    }
    __device__ __forceinline__ bool copy_stored(int dwords, uint32_t n) {
        const int from = 4 * (base_dw + dwords) - stream_off;          // relative to the stream
        if (from < 0 || from + (int)n > stream_len) return false;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(in_base) + stream_off + from;
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
        for (uint32_t done = 0; done < n;) {                      // through the ring, 256 bytes at a time (it is drained in step)
            const uint32_t step = n - done < 256u ? n - done : 256u;
_Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
            for (uint32_t k = (uint32_t)lane; k < step; k += WAVE) *ring_at((uint32_t)a + k) = src[done + k];
            a += (int)step;
            drain();
            done += step;
        }
        return true;
    }
    __device__ __forceinline__ uint32_t reset_input_after_stored(int dwords, uint32_t n) {
        const int byte = 4 * (base_dw + dwords) + (int)n;
        base_dw = byte >> 2;
        win = base_dw;
        load_window();
        return (uint32_t)(byte & 3) * 8u;
    }
    __device__ __forceinline__ int produced() const { return a - a0; }
    __device__ __forceinline__ int capacity() const { return aend - a0; }
    __device__ __forceinline__ void add_count(uint32_t *c) { atomicAdd(c, 1u); }
    __device__ __forceinline__ void fence() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

typedef DevWaveT<0> DevWave;

#ifndef INFL_WAVES
#define INFL_WAVES 1                     // one wave per workgroup: LDS (tables + ring, 5.75 KiB) is allocated per wave
#endif
#ifdef INFL_WAVES_PER_EU                  // (occupancy experiments: caps the kernel's VGPRs so that this many waves fit a SIMD)
#define INFL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(INFL_WAVES_PER_EU, INFL_WAVES_PER_EU)))
#else
#define INFL_OCCUPANCY
#endif
template <int ABLATE>
__global__ __launch_bounds__(INFL_WAVES *WAVE) INFL_OCCUPANCY void k_bgzf_inflate(const uint8_t *__restrict__ comp, const BlockDesc *__restrict__ desc,
                                                                     int n_blocks, uint8_t *out, int32_t *__restrict__ status) {
    __shared__ coral_inflate::Tables tables[INFL_WAVES];
    __shared__ __attribute__((aligned(RING_BYTES))) uint8_t rings[INFL_WAVES][RING_BYTES];      // (DevWaveT::fast relies on the alignment)
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * INFL_WAVES + wib;
    if (b >= n_blocks) return;
    const BlockDesc d = desc[b];
    if (d.isize == 0) {
        if (lane == 0) status[b] = 0;
        return;
    }
    // (comp is a 4-byte aligned allocation of the caller; pointers keep their kernel-argument provenance so that the loads and
    // stores are global_*, not flat_*)
    const uint32_t mis = (uint32_t)(((uintptr_t)comp + d.src_off) & 3u);
    DevWaveT<ABLATE> w;
    w.lane = lane;
    w.lane_half = (float)lane + 0.5f;
    w.in_base = reinterpret_cast<const char *>(comp) + (d.src_off - mis);
    w.stream_off = (int)mis;
    w.stream_len = (int)d.src_len;
    w.base_dw = 0;
    w.win = 0;
    w.last_dw = (int)((mis + d.src_len) >> 2);
    w.a0 = (int)(((uintptr_t)out + d.dst_off) & 255u);
    w.aend = w.a0 + (int)d.isize;
    w.gbase = out + ((long long)d.dst_off - w.a0);
    w.ring = rings[wib];
    w.a = w.gdone = w.a0;
    w.attend = (w.a0 | 255) + 1;
    w.badv = 0;
    w.over = false;
    w.err = 0;
    w.load_window();
    coral_inflate::Inflater<DevWaveT<ABLATE>> inf(w, &tables[wib]);
    int rc = inf.run((int)mis * 8);
    w.finish();
    if (rc == coral_inflate::OK && w.failed()) rc = w.error_code();
    if (lane == 0) status[b] = rc;
}

// ---------------------------------------------------------------------------------------------
// K_crc: the CRC-32 of every inflated block against the one in its BGZF trailer (what htslib checks in bgzf_read_block)
// ---------------------------------------------------------------------------------------------
#define ERR_CRC 100
// One wave per block: every lane takes a contiguous chunk, computes its remainder, shifts it by the number of bytes behind its
// chunk (multiplication by x^(8 n) mod P, coral_crc32.h) and the wave XORs the 64 results.  A block that inflated cleanly but has
// another checksum gets status ERR_CRC.
// Round 3 (the decode timeline showed this kernel at 3.8 ms per 1 GiB batch, serialised behind the inflate): (i) chunks start on
// 64-byte lines of global memory (lane 0 also takes the block's unaligned head) and are read 64 bytes at a time as four aligned
// 16-byte loads, so a cache line is fetched once by the one lane that owns it — with 4-byte loads at a 1 KB lane stride every
// line was touched sixteen times; (ii) four 256-entry tables in LDS (slicing by 4): the four look-ups of a dword are independent
// instead of a chain of four; (iii) the kernel runs on a stream of its own, next to the following batch's inflate.
// One-wave workgroups (as every kernel that runs beside the next batch's inflate launch): that launch keeps 27 waves per CU
// resident — 7 + 7 + 7 + 6 on the four SIMDs, 504 of 512 VGPRs on the full ones — so a four-wave workgroup, which needs room on
// all four SIMDs at once, only started when the inflate launch was nearly over (rocprofv3 timeline: 25 ms instead of 1 ms);
// its 4 KiB of tables fit the LDS the inflate waves leave (27 x 5 888 B of 160 KiB).
__global__ __launch_bounds__(WAVE) void k_bgzf_crc(const uint8_t *__restrict__ out, const BlockDesc *__restrict__ desc, const uint32_t *__restrict__ want,
                                                    int n_blocks, int32_t *__restrict__ status) {
    __shared__ uint32_t T[4][256];
    for (uint32_t i = threadIdx.x; i < 256u; i += WAVE) T[0][i] = coral_crc::table_entry(i);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256u; i += WAVE) {
        uint32_t c = T[0][i];
        for (int k = 1; k < 4; ++k) {                     // T[k][i] = T[0][i] followed by k zero bytes
            c = (c >> 8) ^ T[0][c & 0xffu];
            T[k][i] = c;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int b = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (b >= n_blocks) return;
    const uint32_t n = desc[b].isize;
    if (n == 0 || status[b] != 0) return;
    const uint8_t *p = out + desc[b].dst_off;
    const uint32_t head = (uint32_t)((64u - (uint32_t)((uintptr_t)p & 63u)) & 63u);      // bytes in front of the first 64-byte line
    const uint32_t body = n > head ? n - head : 0u;
    const uint32_t per = ((((body + 63u) >> 6) + 63u) >> 6) << 6;                        // bytes per lane: whole lines
    const uint32_t a = lane == 0 ? 0u : min(n, head + (uint32_t)lane * per);
    const uint32_t e = min(n, head + (uint32_t)(lane + 1) * per);
    auto byte_step = [&](uint32_t r, uint32_t v) { return T[0][(r ^ v) & 0xffu] ^ (r >> 8); };
    auto dword_step = [&](uint32_t r, uint32_t w) {
        r ^= w;
        return T[3][r & 0xffu] ^ T[2][(r >> 8) & 0xffu] ^ T[1][(r >> 16) & 0xffu] ^ T[0][r >> 24];
    };
    uint32_t r = 0, k = a;
    for (; k < e && (((uintptr_t)(p + k)) & 15u); ++k) r = byte_step(r, p[k]);           // (lane 0 only: the unaligned head)
    for (; k + 64 <= e; k += 64) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p + k);
        const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        r = dword_step(r, v0.x); r = dword_step(r, v0.y); r = dword_step(r, v0.z); r = dword_step(r, v0.w);
        r = dword_step(r, v1.x); r = dword_step(r, v1.y); r = dword_step(r, v1.z); r = dword_step(r, v1.w);
        r = dword_step(r, v2.x); r = dword_step(r, v2.y); r = dword_step(r, v2.z); r = dword_step(r, v2.w);
        r = dword_step(r, v3.x); r = dword_step(r, v3.y); r = dword_step(r, v3.z); r = dword_step(r, v3.w);
    }
    for (; k + 4 <= e; k += 4) r = dword_step(r, *reinterpret_cast<const uint32_t *>(p + k));      // (4-byte aligned: k is, from the loop above)
    for (; k < e; ++k) r = byte_step(r, p[k]);
    uint32_t t = coral_crc::shift(r, n - e);
    if (lane == 0) t ^= coral_crc::shift(0xffffffffu, n);
    for (int d = 32; d > 0; d >>= 1) t ^= (uint32_t)__shfl_xor((int)t, d);
    if (lane == 0 && ~t != want[b]) status[b] = ERR_CRC;
}

// ---------------------------------------------------------------------------------------------
// record boundaries
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

// the plausibility test of coral_bam.cpp (plausible_record), on device memory
__device__ __forceinline__ bool plausible(const uint8_t *buf, long long q, long long data_end, int n_ref, long long *len) {
    const long long n = data_end - q;
    if (n < 36) return false;
    const uint8_t *p = buf + q;
    const uint32_t bs = ld32(p);
    if (bs < 34 || bs > (1u << 29)) return false;
    const int32_t refID = (int32_t)ld32(p + 4), pos = (int32_t)ld32(p + 8);
    const uint32_t l_name = p[12], n_cig = ld16(p + 16), l_seq = ld32(p + 20);
    const int32_t mate = (int32_t)ld32(p + 24), mpos = (int32_t)ld32(p + 28);
    if (refID < -1 || refID >= n_ref || mate < -1 || mate >= n_ref || pos < -1 || mpos < -1) return false;
    if (l_name < 2 || l_seq > (1u << 29)) return false;
    const unsigned long long fixed = 32ull + l_name + 4ull * n_cig + ((unsigned long long)l_seq + 1) / 2 + l_seq;
    if (fixed > bs) return false;
    if (n >= 36ll + l_name) {
        const uint8_t *nm = p + 36;
        if (nm[l_name - 1] != 0) return false;
        for (uint32_t k = 0; k + 1 < l_name; ++k)
            if (nm[k] < 33 || nm[k] > 126) return false;
    }
    *len = 4ll + bs;
    return true;
}

// Hop along block_size from x while the record starts inside [.., seg_end) and in front of `limit` and is complete in the
// batch.  Returns the landing position (first start not counted) and the count; *err is set for a record shorter than its
// fixed fields.  The SAME function serves the speculative pass and the exact re-walk.
__device__ __forceinline__ long long hop(const uint8_t *buf, long long x, long long seg_end, long long limit, long long data_end, int *count,
                                         int *err) {
    int c = 0;
    while (x < seg_end && x < limit) {
        if (x + 4 > data_end) break;
        const long long len = 4ll + ld32(buf + x);
        if (len < 36) { *err = 1; break; }
        if (x + len > data_end) break;
        ++c;
        x += len;
    }
    *count = c;
    return x;
}

// per segment s (buffer offsets [s * SEG, (s + 1) * SEG) clipped to [begin, data_end)): first plausible start, landing, count
__global__ __launch_bounds__(256) void k_bam_find(const uint8_t *__restrict__ buf, long long begin, long long data_end, long long limit,
                                                   int n_ref, int seg0, int n_seg, long long *__restrict__ seg_first,
                                                   long long *__restrict__ seg_land, int32_t *__restrict__ seg_count,
                                                   int32_t *__restrict__ seg_valid) {
    const int lane = threadIdx.x & 63;
    const int s = seg0 + (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (s >= seg0 + n_seg) return;
    const long long a = max((long long)s * SEG_BYTES, begin), b = min((long long)(s + 1) * SEG_BYTES, data_end);
    long long first = -1;
    // 256 positions per round: the first eight bytes of all four positions of a lane (block_size, refID) are loaded together —
    // ONE memory round trip per round, and that filter alone rejects nearly every position that is not a record start (round 3:
    // with one position per lane and round the kernel was a chain of ~280 dependent round trips per segment)
    for (long long x0 = a; x0 < b && first < 0; x0 += 4 * WAVE) {
        bool pre[4];
        unsigned long long head[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long x = x0 + j * WAVE + lane;
            pre[j] = x < b && x + 36 <= data_end;
            head[j] = 0;
            if (pre[j]) __builtin_memcpy(&head[j], buf + x, 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bs = (uint32_t)head[j];
            const int32_t refID = (int32_t)(head[j] >> 32);
            pre[j] = pre[j] && bs >= 34 && bs <= (1u << 29) && refID >= -1 && refID < n_ref;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (first >= 0) break;
            bool ok = false;
            if (pre[j]) {
                long long q = x0 + j * WAVE + lane;
                int chain = 0;
                ok = true;
                while (chain < 8 && q + 36 <= data_end) {
                    long long len;
                    if (!plausible(buf, q, data_end, n_ref, &len)) { ok = false; break; }
                    q += len;
                    ++chain;
                }
                ok = ok && chain >= 3;
            }
            const unsigned long long m = __ballot(ok);
            if (m != 0ull) first = x0 + j * WAVE + (long long)__builtin_ctzll(m);
        }
    }
    int count = 0, err = 0;
    long long land = first;
    if (first >= 0) land = hop(buf, first, b, limit, data_end, &count, &err);
    if (lane == 0) {
        seg_first[s] = err ? -1 : first;           // (a guess that runs into a malformed record is no guess)
        seg_land[s] = land;
        seg_count[s] = count;
        seg_valid[s] = 0;
    }
}

// result: [0] records, [1] carry position (first start not counted; may lie behind data_end), [2] done (limit reached),
//         [3] error (1 = malformed record, 2 = no record start found while searching), [4] segments re-walked, [5] first start
__global__ __launch_bounds__(WAVE) void k_bam_verify(const uint8_t *__restrict__ buf, long long known_start, long long begin, long long data_end,
                                                      long long limit, int seg0, int n_seg, long long *__restrict__ seg_first,
                                                      long long *__restrict__ seg_land, int32_t *__restrict__ seg_count,
                                                      int32_t *__restrict__ seg_valid, long long *__restrict__ seg_base,
                                                      long long *__restrict__ result) {
    const int lane = threadIdx.x;
    long long cur = known_start;
    if (cur < 0) {                                   // first batch of a byte range: the lowest confirmed candidate
        for (int s0 = seg0; s0 < seg0 + n_seg && cur < 0; s0 += WAVE) {
            const int s = s0 + lane;
            const long long f = s < seg0 + n_seg ? seg_first[s] : -1;
            const unsigned long long m = __ballot(f >= 0);
            if (m != 0ull) cur = __shfl(f, (int)__builtin_ctzll(m));
        }
        if (cur < 0) {
            if (lane == 0) { result[0] = 0; result[1] = data_end; result[2] = 0; result[3] = 2; result[4] = 0; result[5] = -1; }
            return;
        }
    }
    const long long first_start = cur;
    long long total = 0, fixups = 0;
    int error = 0, done = 0;
    const int s_end = seg0 + n_seg;
    // The walk is serial (the next segment is where this one's records land), but its memory round trips need not be: lane k
    // loads the guess of segment sb + k, and the chain is followed through those 64 segments in registers with v_readlane.
    // Positions inside a batch fit 32 bits (coral_bamgpu_open keeps CARRY_CAP + the batch below 4 GiB), which halves the
    // arithmetic of the loop — one wave, one instruction at a time: the instruction count IS the time (round 3: one dependent
    // round trip and ~140 instructions per segment were 6 ms per 2.4 GiB batch, on the parse stream's critical path).
    const uint32_t lim32 = limit > 0xffffffffll ? 0xffffffffu : (uint32_t)limit, end32 = (uint32_t)data_end;
    uint32_t c32 = (uint32_t)cur;
    for (bool stop = false; !stop;) {
        if (c32 >= lim32) { done = 1; break; }
        if ((unsigned long long)c32 + 4 > (unsigned long long)data_end) break;
        const int sb = (int)(c32 / (uint32_t)SEG_BYTES), sl = sb + lane;
        const bool have = sl < s_end;
        const long long f64 = have ? seg_first[sl] : -1, land64 = have ? seg_land[sl] : -1;
        const int f_l = f64 < 0 ? -1 : (int)(uint32_t)f64;             // (a position is never 0xffffffff: it would lie beyond data_end)
        const int land_l = (int)(uint32_t)land64, cnt_l = have ? seg_count[sl] : 0;
        long long base_l = 0;
        bool visited_l = false;
        for (;;) {
            const int k = __builtin_amdgcn_readfirstlane((int)(c32 / (uint32_t)SEG_BYTES) - sb);
            if (k >= WAVE) break;                        // the next 64 segments
            const uint32_t seg_end = min((uint32_t)(sb + k + 1) * (uint32_t)SEG_BYTES - 1u, end32 - 1u) + 1u;    // (no overflow at 4 GiB)
            uint32_t land = (uint32_t)__builtin_amdgcn_readlane(land_l, k);
            int cnt = __builtin_amdgcn_readlane(cnt_l, k);
            if ((uint32_t)__builtin_amdgcn_readlane(f_l, k) != c32) {           // the guess is not on the chain (or there was none): exact walk
                int err = 0;
                const int s = sb + k;
                const long long l64 = hop(buf, (long long)c32, (long long)seg_end, limit, data_end, &cnt, &err);
                if (err) { error = 1; stop = true; break; }
                land = (uint32_t)l64;
                ++fixups;
                if (lane == 0) { seg_first[s] = (long long)c32; seg_land[s] = l64; seg_count[s] = cnt; }
            }
            if (lane == k) { base_l = total; visited_l = true; }
            total += cnt;
            c32 = land;
            if (c32 >= lim32) { done = 1; stop = true; break; }
            if (c32 < seg_end || (unsigned long long)c32 + 4 > (unsigned long long)data_end) { stop = true; break; }   // stopped inside its segment: an incomplete record starts here
        }
        if (visited_l) { seg_base[sl] = base_l; seg_valid[sl] = 1; }
    }
    cur = (long long)c32;
    if (lane == 0) {
        result[0] = total; result[1] = cur; result[2] = done; result[3] = error; result[4] = fixups; result[5] = first_start;
    }
}

__global__ void k_bam_starts(const uint8_t *__restrict__ buf, int seg0, int n_seg, const long long *__restrict__ seg_first,
                             const int32_t *__restrict__ seg_count, const int32_t *__restrict__ seg_valid,
                             const long long *__restrict__ seg_base, long long *__restrict__ rec_start) {
    const int s = seg0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= seg0 + n_seg || !seg_valid[s]) return;
    long long x = seg_first[s];
    const long long base = seg_base[s];
    const int n = seg_count[s];
    for (int i = 0; i < n; ++i) {
        rec_start[base + i] = x;
        x += 4ll + ld32(buf + x);
    }
}

// ---------------------------------------------------------------------------------------------
// K_meta: one wave per record
// ---------------------------------------------------------------------------------------------
struct MetaArrays {
    int32_t *tid, *pos, *flag, *mapq, *l_seq, *nm, *n_cigar;       // what the host mirrors need (+ end, qlen from k_bam_emit)
    long long *cig_src, *seq_src, *sa_src;                         // buffer offsets
    long long *pad_ops, *name_len, *sa_len;                        // scan inputs (n + 1 entries, the last one 0)
};

enum { REC_ERR_SHORT = 1, REC_ERR_FIELDS = 2, REC_ERR_TAG_B = 3, REC_ERR_TAG_TYPE = 4, REC_ERR_TAG_OVERRUN = 5 };

__global__ __launch_bounds__(256) void k_bam_meta(const uint8_t *__restrict__ buf, const long long *__restrict__ rec_start, long long n_rec,
                                                   MetaArrays M, int32_t *__restrict__ error) {
    const int lane = threadIdx.x & 63;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i > n_rec) return;
    if (i == n_rec) {                                               // the extra scan slot
        if (lane == 0) { M.pad_ops[i] = 0; M.name_len[i] = 0; M.sa_len[i] = 0; }
        return;
    }
    const long long x = rec_start[i];
    const uint8_t *r = buf + x;
    const uint32_t block_size = ld32(r);
    const uint8_t *p = r + 4;
    int err = 0;
    int32_t refID = -1, pos = -1, nm = 0;
    uint32_t l_read_name = 0, mapq = 0, n_cigar_op = 0, flag = 0, l_seq = 0;
    long long cig_src = 0, seq_src = 0, sa_src = 0, sa_len = 0;
    if (block_size < 32) {
        err = REC_ERR_SHORT;
    } else {
        refID = (int32_t)ld32(p);
        pos = (int32_t)ld32(p + 4);
        l_read_name = p[8];
        mapq = p[9];
        n_cigar_op = ld16(p + 12);
        flag = ld16(p + 14);
        l_seq = ld32(p + 16);
        const uint8_t *name = p + 32;
        const uint8_t *cig = name + l_read_name;
        const uint8_t *seq = cig + 4ull * n_cigar_op;
        const uint8_t *qual = seq + ((unsigned long long)l_seq + 1) / 2;
        const uint8_t *tags = qual + l_seq;
        const uint8_t *endp = p + block_size;
        if (tags > endp || l_read_name == 0) {
            err = REC_ERR_FIELDS;
        } else {
            const uint8_t *cg = nullptr;
            uint32_t cg_n = 0;
            const uint8_t *t = tags;
            while (t + 3 <= endp) {                                  // wave-uniform walk; Z / H strings are searched 64 bytes per step
                const uint8_t ta = t[0], tb = t[1], ty = t[2];
                const uint8_t *v = t + 3;
                unsigned long long sz = 0;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') {
                    long long found = -1;
                    for (const uint8_t *q = v; q < endp; q += WAVE) {
                        const uint8_t c = (q + lane < endp) ? q[lane] : (uint8_t)1;
                        const unsigned long long m = __ballot(c == 0);
                        if (m != 0ull) { found = (q - v) + (long long)__builtin_ctzll(m); break; }
                    }
                    sz = found >= 0 ? (unsigned long long)found + 1 : (unsigned long long)(endp - v) + 1;
                } else if (ty == 'B') {
                    if (v + 5 > endp) { err = REC_ERR_TAG_B; break; }
                    const uint8_t sub = v[0];
                    const uint32_t cnt = ld32(v + 1);
                    const unsigned long long es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if (ta == 'C' && tb == 'G' && sub == 'I') { cg = v + 5; cg_n = cnt; }
                    sz = 5 + es * cnt;
                } else { err = REC_ERR_TAG_TYPE; break; }
                if (v + sz > endp) { err = REC_ERR_TAG_OVERRUN; break; }
                if (ta == 'N' && tb == 'M') {
                    if (ty == 'c') nm = (int8_t)v[0];
                    else if (ty == 'C') nm = v[0];
                    else if (ty == 's') nm = (int16_t)ld16(v);
                    else if (ty == 'S') nm = (int32_t)ld16(v);
                    else if (ty == 'i' || ty == 'I') nm = (int32_t)ld32(v);
                } else if (ta == 'S' && tb == 'A' && ty == 'Z') {
                    sa_src = v - buf;
                    sa_len = (long long)sz;                          // with the terminating NUL
                }
                t = v + sz;
            }
            const uint8_t *cig_from = cig;
            if (!err && cg && n_cigar_op == 2 && (ld32(cig) & 15u) == 4 && (ld32(cig) >> 4) == l_seq && (ld32(cig + 4) & 15u) == 3) {
                cig_from = cg;                                       // long CIGAR in the CG tag (SAM spec §4.2.2)
                n_cigar_op = cg_n;
            }
            cig_src = cig_from - buf;
            seq_src = seq - buf;
        }
    }
    if (lane == 0) {
        if (err) atomicMax(error, err);
        M.tid[i] = refID; M.pos[i] = pos; M.flag[i] = (int32_t)flag; M.mapq[i] = (int32_t)mapq; M.l_seq[i] = (int32_t)l_seq;
        M.nm[i] = nm; M.n_cigar[i] = err ? 0 : (int32_t)n_cigar_op;
        M.cig_src[i] = cig_src; M.seq_src[i] = seq_src; M.sa_src[i] = sa_src;
        M.pad_ops[i] = err ? 0 : (long long)(((unsigned long long)n_cigar_op + 3ull) & ~3ull);
        M.name_len[i] = err ? 0 : (long long)l_read_name;
        M.sa_len[i] = err ? 0 : sa_len;
    }
}

// ---------------------------------------------------------------------------------------------
// K_emit: one wave per record
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bam_emit(const uint8_t *__restrict__ buf, const long long *__restrict__ rec_start, long long n_rec,
                                                   MetaArrays M, const long long *__restrict__ cig_off, const long long *__restrict__ name_off,
                                                   const long long *__restrict__ sa_off, uint32_t *__restrict__ cigar_dst,
                                                   int32_t *__restrict__ end_out, int32_t *__restrict__ qlen_out, uint8_t *__restrict__ names,
                                                   uint8_t *__restrict__ sa_text, int32_t *__restrict__ na_list, int32_t *__restrict__ na_count) {
    const int lane = threadIdx.x & 63;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_rec) return;
    const int n = M.n_cigar[i];
    const long long padded = M.pad_ops[i];
    const uint8_t *cig = buf + M.cig_src[i];
    uint32_t *dst = cigar_dst + cig_off[i];
    long long rlen = 0, qinf = 0;
    for (long long k = lane; k < padded; k += WAVE) {
        const uint32_t w = k < n ? ld32(cig + 4 * k) : 15u;
        dst[k] = w;
        const uint32_t op = w & 15u, len = w >> 4;
        rlen += ((0x18Du >> op) & 1u) ? len : 0;                      // M D N = X
        qinf += ((0x1B3u >> op) & 1u) ? len : 0;                      // M I S H = X
    }
    for (int d = 32; d > 0; d >>= 1) {
        rlen += __shfl_xor(rlen, d);
        qinf += __shfl_xor(qinf, d);
    }
    const int32_t flag = M.flag[i], l_seq = M.l_seq[i], pos = M.pos[i];
    if ((flag & 4) || n == 0) rlen = 0;                               // htslib bam_endpos
    if (lane == 0) {
        end_out[i] = pos + (int32_t)(rlen > 0 ? rlen : 1);
        qlen_out[i] = l_seq > 0 ? l_seq : (int32_t)qinf;
    }
    // read name (with its NUL) and SA text (with its NUL)
    {
        const uint8_t *src = buf + rec_start[i] + 36;
        const long long nl = M.name_len[i];
        uint8_t *d = names + name_off[i];
        for (long long k = lane; k < nl; k += WAVE) d[k] = src[k];
        const long long sl = M.sa_len[i];
        if (sl) {
            const uint8_t *s = buf + M.sa_src[i];
            uint8_t *e = sa_text + sa_off[i];
            for (long long k = lane; k < sl; k += WAVE) e[k] = s[k];
        }
    }
    // aligned non-ACGT bases are rare: only FIND the records that have any non-ACGT code (pysam count_coverage counts A/C/G/T)
    if (l_seq > 0 && !(flag & 4) && n > 0) {
        const uint8_t *seq = buf + M.seq_src[i];
        const long long full = l_seq / 2;
        uint32_t bad = 0;
        const long long words = full / 4;
        for (long long k = lane; k < words; k += WAVE) {
            const uint32_t x = ld32(seq + 4 * k);
            const uint32_t lo = x & 0x0f0f0f0fu, hi = (x >> 4) & 0x0f0f0f0fu;
            const uint32_t tl = (lo | 0x10101010u) - 0x01010101u, th = (hi | 0x10101010u) - 0x01010101u;
            bad |= (tl & lo & 0x0f0f0f0fu) | (~tl & 0x10101010u) | (th & hi & 0x0f0f0f0fu) | (~th & 0x10101010u);
        }
        for (long long k = words * 4 + lane; k < full; k += WAVE) {
            const uint32_t h = seq[k] >> 4, l = seq[k] & 15u;
            bad |= (h == 0 || (h & (h - 1))) | (l == 0 || (l & (l - 1)));
        }
        if ((l_seq & 1) && lane == 0) {
            const uint32_t h = seq[full] >> 4;
            bad |= (h == 0 || (h & (h - 1)));
        }
        if (__ballot(bad != 0) != 0ull && lane == 0) na_list[atomicAdd(na_count, 1)] = (int32_t)i;
    }
}

// whole records (the rare ones with non-ACGT bases) -> one contiguous buffer for the host: one wave per record
__global__ __launch_bounds__(256) void k_bam_gather(const uint8_t *__restrict__ buf, const long long *__restrict__ src, const long long *__restrict__ dst,
                                                     const long long *__restrict__ len, int n, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (j >= n) return;
    const uint8_t *s = buf + src[j];
    uint8_t *d = out + dst[j];
    const long long m = len[j];
    for (long long k = lane; k < m; k += WAVE) d[k] = s[k];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BatchInfo {
    int n_blocks = 0;
    uint64_t file_off = 0, comp_bytes = 0, infl_bytes = 0;
    uint64_t ubase = 0;              // offset of the batch's first inflated byte in this range's uncompressed stream
    bool has_limit = false;          // the next byte range begins at or in front of this batch's end:
    long long limit_rel = 0;         //   at this offset from the batch's first inflated byte (negative: in an earlier batch)
    bool last = false;               // nothing follows
};

// what a batch leaves for the host-side worker: read names -> ids, SA text -> rows, non-ACGT records -> positions
struct HostJob {
    size_t base = 0;                 // ordinal of the batch's first record in the decoded range
    long long n = 0;
    std::vector<long long> pad, name_off, sa_off;
    std::vector<uint8_t> names, sa_text;
    std::vector<int32_t> na_list;    // (sorted) batch-local ordinals of the records with a non-ACGT code
    std::vector<long long> na_off;   // their bytes in na_raw (n + 1 offsets); a record = its bytes behind block_size
    std::vector<uint8_t> na_raw;
};

struct GpuDecoder {
    MappedFile f;
    Decoded D;
    RefIds ref_id;
    size_t hdr_bytes = 0;
    int rank = 0, world = 1, n_threads = 1, device = 0;
    uint64_t byte_lo = 0, byte_hi = 0, first_block = 0;
    bool last_rank = true;
    std::string error;
    // capacities
    size_t infl_cap = 0, comp_cap = 0, max_blocks = 0, rec_cap = 0, nseg_cap = 0;
    uint64_t first_batch = FIRST_BATCH;
    size_t ws_bytes = 0;
    // device workspace (carved from the caller's allocation)
    uint8_t *d_comp[2] = {nullptr, nullptr}, *d_infl[2] = {nullptr, nullptr};
    BlockDesc *d_desc[2] = {nullptr, nullptr};
    uint32_t *d_crc[2] = {nullptr, nullptr};
    int32_t *d_status[2] = {nullptr, nullptr};
    long long *d_seg_first = nullptr, *d_seg_land = nullptr, *d_seg_base = nullptr, *d_result = nullptr, *d_rec_start = nullptr;
    int32_t *d_seg_count = nullptr, *d_seg_valid = nullptr, *d_error = nullptr, *d_na_list = nullptr, *d_na_count = nullptr;
    MetaArrays M{};
    long long *d_cig_off = nullptr, *d_name_off = nullptr, *d_sa_off = nullptr;
    int32_t *d_end = nullptr, *d_qlen = nullptr;
    uint8_t *d_names = nullptr, *d_sa_text = nullptr;
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0, names_cap = 0, sa_cap = 0;
    // pinned staging + streams
    uint8_t *h_stage[N_STAGE] = {};
    BlockDesc *h_desc[2] = {nullptr, nullptr};      // (pinned; the blocks' trailer CRCs follow the table in the same buffer)
    hipStream_t s_copy = nullptr, s_infl = nullptr, s_crc = nullptr;
    hipEvent_t ev_stage[N_STAGE] = {};
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_infl[2] = {nullptr, nullptr}, ev_parsed[2] = {nullptr, nullptr};
    hipEvent_t ev_crc[2] = {nullptr, nullptr};          // the slot's checksum kernel has run (its own stream, beside the next inflate)
    double t_read = 0, t_alloc = 0, t_wait_staged = 0, t_wait_gpu = 0;
    // feeder
    std::thread feeder;
    std::mutex m;
    std::condition_variable cv;
    std::vector<BatchInfo> staged;            // batches whose H2D has been issued
    bool feeder_done = false, stop = false;
    int inflate_launched = 0;                 // batches whose inflate kernel has been enqueued
    int emitted = 0;                          // batches whose parse has been enqueued completely (ev_parsed recorded)
    std::string feeder_error, launch_error;
    // walk state
    int k = 0;                                // next batch to parse
    bool searching = false, finished = false;
    long long known_start = 0;                // buffer offset of the first record of batch k (CARRY_CAP-based)
    long long carry_len = 0;                  // bytes carried in front of batch k
    // the batch between next() and emit()
    BatchInfo cur;
    long long cur_n_rec = 0, cur_ops = 0, cur_name_bytes = 0, cur_sa_bytes = 0;
    bool have_cur = false;
    // host-side worker
    std::thread worker;
    std::mutex wm;
    std::condition_variable wcv;
    std::deque<std::unique_ptr<HostJob>> jobs;
    bool worker_stop = false, worker_busy = false;
    std::string worker_error;
    long long cur_carry_pos = 0;
    // statistics
    double t_open = 0, seconds = 0, host_seconds = 0;
    int64_t fixups = 0, n_batches = 0, na_records = 0;
    std::chrono::steady_clock::time_point t_start;

    ~GpuDecoder() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        if (feeder.joinable()) feeder.join();
        {
            std::lock_guard<std::mutex> lk(wm);
            worker_stop = true;
        }
        wcv.notify_all();
        if (worker.joinable()) worker.join();
        // (an error may leave kernels and copies in flight: they work in the caller's workspace, which is about to be released)
        if (s_copy) (void)hipStreamSynchronize(s_copy);
        if (s_infl) (void)hipStreamSynchronize(s_infl);
        if (s_crc) (void)hipStreamSynchronize(s_crc);
        for (int i = 0; i < N_STAGE; ++i) {
            if (h_stage[i]) (void)hipHostFree(h_stage[i]);
            if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]);
        }
        for (int i = 0; i < 2; ++i) {
            if (h_desc[i]) (void)hipHostFree(h_desc[i]);
            if (ev_h2d[i]) (void)hipEventDestroy(ev_h2d[i]);
            if (ev_infl[i]) (void)hipEventDestroy(ev_infl[i]);
            if (ev_crc[i]) (void)hipEventDestroy(ev_crc[i]);
            if (ev_parsed[i]) (void)hipEventDestroy(ev_parsed[i]);
        }
        if (s_copy) (void)hipStreamDestroy(s_copy);
        if (s_infl) (void)hipStreamDestroy(s_infl);
        if (s_crc) (void)hipStreamDestroy(s_crc);
    }
};

#define HIP_OK(call, what)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            G->error = std::string(what) + ": " + hipGetErrorString(e_);                    \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

// parallel pread of [off, off + n) into dst
bool read_range(int fd, uint64_t off, size_t n, uint8_t *dst, int n_threads) {
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, n / (4u << 20) + 1));
    std::atomic<bool> bad{false};
    auto work = [&](int t) {
        size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
        while (a < b) {
            const ssize_t got = pread(fd, dst + a, b - a, (off_t)(off + a));
            if (got <= 0) { bad = true; return; }
            a += (size_t)got;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    return !bad;
}

bool launch_inflate(GpuDecoder *G, int kb, const BatchInfo &bi);

// The feeder: cuts the byte range into batches (whole BGZF blocks, <= infl_cap inflated and <= comp_cap compressed bytes),
// reads each into pinned memory, builds its block table and sends both to the device.
void feeder_main(GpuDecoder *G) {
    (void)hipSetDevice(G->device);
    uint64_t at = G->first_block;
    uint64_t ubase = 0;
    long long own_bytes = -1;                 // known once the first block of the next range has been seen
    size_t overhang_left = OVERHANG_BLOCKS;
    int kb = 0, chunk_no = 0;
    auto fail = [&](const std::string &msg) {
        std::lock_guard<std::mutex> lk(G->m);
        G->feeder_error = msg;
        G->feeder_done = true;
        G->cv.notify_all();
    };
    std::vector<BlockDesc> desc;
    std::vector<uint32_t> crcs;
    for (;;) {
        const int slot = kb & 1;
        if (at >= G->f.size || (own_bytes >= 0 && overhang_left == 0)) break;
        {   // the slot's device buffers are free once the inflate of batch kb - 2 has run
            std::unique_lock<std::mutex> lk(G->m);
            G->cv.wait(lk, [&] { return G->stop || G->inflate_launched >= kb - 1; });
            if (G->stop) return;
        }
        if (kb >= 2 && (hipEventSynchronize(G->ev_infl[slot]) != hipSuccess || hipEventSynchronize(G->ev_crc[slot]) != hipSuccess))
            return fail("hipEventSynchronize failed in the feeder");
        // the first batches are small so that the GPU has something to inflate almost at once; then they double up to the cap
        const uint64_t infl_cap = std::min<uint64_t>(G->infl_cap, G->first_batch << std::min(kb, 20));
        const uint64_t comp_cap = std::min<uint64_t>(G->comp_cap, std::max<uint64_t>(infl_cap / 2, 1u << 20));
        BatchInfo bi;
        bi.file_off = at;
        bi.ubase = ubase;
        desc.clear();
        crcs.clear();
        uint64_t comp = 0, infl = 0;          // bytes of the batch so far
        bool full = false;
        while (!full && at < G->f.size) {
            // one chunk: read -> walk its whole BGZF blocks -> send them behind what the batch already has
            const int cs = chunk_no % N_STAGE;
            if (chunk_no >= N_STAGE && hipEventSynchronize(G->ev_stage[cs]) != hipSuccess) return fail("hipEventSynchronize failed in the feeder");
            const size_t want = (size_t)std::min<uint64_t>(std::min<uint64_t>(STAGE_BYTES, comp_cap - comp), G->f.size - at);
            if (want < 28) break;
            uint8_t *data = G->h_stage[cs];
            const auto t0 = std::chrono::steady_clock::now();
            if (!read_range(G->f.fd, at, want, data, G->n_threads)) return fail("reading the BAM file failed");
            G->t_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t p = 0;
            while (p < want) {
                if (desc.size() >= G->max_blocks) { full = true; break; }
                Block b;
                if (!bgzf_header(data + p, want - p, b)) {
                    if (want - p >= 65536 + 26 || at + want >= G->f.size) return fail("not a BGZF block");
                    if (want < STAGE_BYTES) full = true;           // the block does not fit what is left of the batch
                    break;                                         // the block continues behind what was read
                }
                if (infl + b.isize > infl_cap) { full = true; break; }
                const bool owned = at + p < G->byte_hi;
                if (!owned) {
                    if (own_bytes < 0) own_bytes = (long long)(ubase + infl);
                    if (overhang_left == 0) { full = true; break; }
                    --overhang_left;
                }
                BlockDesc d;
                d.src_off = (uint32_t)(comp + p + b.hdr);
                d.src_len = b.csize - b.hdr - 8;
                d.dst_off = (uint32_t)infl;
                d.isize = b.isize;
                desc.push_back(d);
                crcs.push_back(rd32(data + p + b.csize - 8));
                infl += b.isize;
                p += b.csize;
                if (owned) { G->D.compressed_bytes += b.csize; G->D.uncompressed_bytes += b.isize; ++G->D.n_blocks; }
            }
            if (p == 0) {
                if (desc.empty()) return fail("a BGZF block does not fit the batch buffers");
                break;
            }
            if (hipMemcpyAsync(G->d_comp[slot] + comp, data, p, hipMemcpyHostToDevice, G->s_copy) != hipSuccess ||
                hipEventRecord(G->ev_stage[cs], G->s_copy) != hipSuccess)
                return fail("host-to-device copy of compressed bytes failed");
            ++chunk_no;
            comp += p;
            at += p;
            if (comp + 65536 + 26 > comp_cap) full = true;
        }
        if (desc.empty()) break;
        bi.n_blocks = (int)desc.size();
        bi.comp_bytes = comp;
        bi.infl_bytes = infl;
        if (own_bytes >= 0) {
            bi.has_limit = true;
            bi.limit_rel = own_bytes - (long long)ubase;
        }
        ubase += infl;
        bi.last = at >= G->f.size || (own_bytes >= 0 && overhang_left == 0);
        // the block table and the trailer CRCs go through their own pinned buffer (one per batch slot; host and device copies are
        // free again once the slot's inflate + checksum kernels have run: see the wait at the top of the round)
        memcpy(G->h_desc[slot], desc.data(), desc.size() * sizeof(BlockDesc));
        uint32_t *h_crc = reinterpret_cast<uint32_t *>(G->h_desc[slot] + G->max_blocks);
        memcpy(h_crc, crcs.data(), crcs.size() * 4);
        if (hipMemcpyAsync(G->d_desc[slot], G->h_desc[slot], desc.size() * sizeof(BlockDesc), hipMemcpyHostToDevice, G->s_copy) != hipSuccess ||
            hipMemcpyAsync(G->d_crc[slot], h_crc, crcs.size() * 4, hipMemcpyHostToDevice, G->s_copy) != hipSuccess ||
            hipEventRecord(G->ev_h2d[slot], G->s_copy) != hipSuccess)
            return fail("host-to-device copy of a block table failed");
        {
            std::lock_guard<std::mutex> lk(G->m);
            G->staged.push_back(bi);
        }
        G->cv.notify_all();
        // ... and its inflate is enqueued from here as well, as soon as the batch that used this slot's inflated buffer before
        // (kb - 2) has been parsed: the GPU never waits for the caller's thread to come round
        {
            std::unique_lock<std::mutex> lk(G->m);
            G->cv.wait(lk, [&] { return G->stop || G->emitted >= kb - 1; });
            if (G->stop) return;
        }
        if (!launch_inflate(G, kb, bi)) return fail(G->launch_error);
        ++kb;
        if (bi.last) break;
    }
    {
        std::lock_guard<std::mutex> lk(G->m);
        G->feeder_done = true;
    }
    G->cv.notify_all();
}

// The host-side worker: batches in file order.  Touches only D.cigar_off, D.name_id, D.names, D.sa, D.sa_nm, D.sa_off, D.na_rec,
// D.na_pos (the caller's thread fills the per-record integer columns).
void worker_main(GpuDecoder *G) {
    Decoded &D = G->D;
    for (;;) {
        std::unique_ptr<HostJob> job;
        {
            std::unique_lock<std::mutex> lk(G->wm);
            G->wcv.wait(lk, [&] { return G->worker_stop || !G->jobs.empty(); });
            if (G->jobs.empty()) return;
            job = std::move(G->jobs.front());
            G->jobs.pop_front();
            G->worker_busy = true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::string err;
        HostJob &J = *job;
        for (long long i = 0; i < J.n && err.empty(); ++i) {
            D.cigar_off.push_back(D.cigar_off.back() + J.pad[(size_t)i]);
            const char *s = (const char *)J.names.data() + J.name_off[(size_t)i];
            const size_t len = (size_t)(J.name_off[(size_t)i + 1] - J.name_off[(size_t)i]);
            // as the CPU pipeline: the bytes in front of the last one, cut at a NUL
            D.name_id.push_back(D.names.intern(s, len ? strnlen(s, len - 1) : 0));
            int32_t cnt = 0;
            if (J.sa_off[(size_t)i + 1] > J.sa_off[(size_t)i]) {
                const char *q = (const char *)J.sa_text.data() + J.sa_off[(size_t)i];
                const char *qe = q + (J.sa_off[(size_t)i + 1] - J.sa_off[(size_t)i]);
                while (q < qe && *q) {
                    const char *e = q;
                    while (e < qe && *e && *e != ';') ++e;
                    if (e > q) {
                        int32_t row[8], snm = 0;
                        if (!parse_sa_entry(q, e, G->ref_id, row, &snm)) { err = "malformed SA entry"; break; }
                        D.sa.insert(D.sa.end(), row, row + 8);
                        D.sa_nm.push_back(snm);
                        ++cnt;
                    }
                    q = (e < qe && *e == ';') ? e + 1 : e;
                }
            }
            D.sa_off.push_back(D.sa_off.back() + cnt);
        }
        // records with a non-ACGT code: their aligned non-ACGT positions come from the CPU pipeline's own routine
        for (size_t j = 0; j < J.na_list.size() && err.empty(); ++j) {
            Partial pt;
            if (!decode_record(J.na_raw.data() + J.na_off[j], (uint32_t)(J.na_off[j + 1] - J.na_off[j]), G->ref_id, pt, err)) break;
            for (size_t q = 0; q < pt.na_pos.size(); ++q) {
                D.na_rec.push_back((int64_t)J.base + J.na_list[j]);
                D.na_pos.push_back(pt.na_pos[q]);
            }
        }
        {
            std::lock_guard<std::mutex> lk(G->wm);
            G->host_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!err.empty() && G->worker_error.empty()) G->worker_error = err;
            G->worker_busy = false;
        }
        G->wcv.notify_all();
    }
}

// Lays the device workspace out (ws == nullptr: only its size is computed, the pointers stay meaningless).
bool carve(GpuDecoder *G, void *ws, size_t bytes) {
    const uintptr_t p = (uintptr_t)ws;
    size_t used = 0;
    auto take = [&](size_t n) -> void * {
        void *q = (void *)(p + used);
        used += up256(n);
        return q;
    };
    for (int i = 0; i < 2; ++i) {
        G->d_comp[i] = (uint8_t *)take(G->comp_cap + COMP_SLACK);
        G->d_infl[i] = (uint8_t *)take((size_t)CARRY_CAP + G->infl_cap + COMP_SLACK);
        G->d_desc[i] = (BlockDesc *)take(G->max_blocks * sizeof(BlockDesc));
        G->d_crc[i] = (uint32_t *)take(G->max_blocks * 4);
        G->d_status[i] = (int32_t *)take(G->max_blocks * 4);
    }
    const size_t ns = G->nseg_cap, nr = G->rec_cap + 1;
    G->d_seg_first = (long long *)take(ns * 8);
    G->d_seg_land = (long long *)take(ns * 8);
    G->d_seg_base = (long long *)take(ns * 8);
    G->d_seg_count = (int32_t *)take(ns * 4);
    G->d_seg_valid = (int32_t *)take(ns * 4);
    G->d_result = (long long *)take(64);
    G->d_error = (int32_t *)take(16);
    G->d_na_count = (int32_t *)take(16);
    G->d_rec_start = (long long *)take(nr * 8);
    G->M.tid = (int32_t *)take(nr * 4); G->M.pos = (int32_t *)take(nr * 4); G->M.flag = (int32_t *)take(nr * 4);
    G->M.mapq = (int32_t *)take(nr * 4); G->M.l_seq = (int32_t *)take(nr * 4); G->M.nm = (int32_t *)take(nr * 4);
    G->M.n_cigar = (int32_t *)take(nr * 4);
    G->M.cig_src = (long long *)take(nr * 8); G->M.seq_src = (long long *)take(nr * 8); G->M.sa_src = (long long *)take(nr * 8);
    G->M.pad_ops = (long long *)take(nr * 8); G->M.name_len = (long long *)take(nr * 8); G->M.sa_len = (long long *)take(nr * 8);
    G->d_cig_off = (long long *)take(nr * 8); G->d_name_off = (long long *)take(nr * 8); G->d_sa_off = (long long *)take(nr * 8);
    G->d_end = (int32_t *)take(nr * 4); G->d_qlen = (int32_t *)take(nr * 4);
    G->d_na_list = (int32_t *)take(nr * 4);
    G->d_scan_tmp = take(G->scan_tmp_bytes);
    // read names and SA text of one batch can never exceed its inflated bytes (+ what was carried)
    G->names_cap = G->sa_cap = (size_t)CARRY_CAP + G->infl_cap;
    G->d_names = (uint8_t *)take(G->names_cap);
    G->d_sa_text = (uint8_t *)take(G->sa_cap);
    if (ws && used > bytes) return false;
    G->ws_bytes = used;
    return true;
}

#define HIP_LAUNCH_OK(call, what)                                                           \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            G->launch_error = std::string(what) + ": " + hipGetErrorString(e_);             \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

bool launch_inflate(GpuDecoder *G, int kb, const BatchInfo &bi) {
    const int slot = kb & 1;
    HIP_LAUNCH_OK(hipStreamWaitEvent(G->s_infl, G->ev_h2d[slot], 0), "hipStreamWaitEvent");
    if (kb >= 2) HIP_LAUNCH_OK(hipStreamWaitEvent(G->s_infl, G->ev_parsed[slot], 0), "hipStreamWaitEvent");     // the buffer's previous batch has been parsed
    const int grid = (bi.n_blocks + INFL_WAVES - 1) / INFL_WAVES;
    hipLaunchKernelGGL(k_bgzf_inflate<0>, dim3(grid), dim3(INFL_WAVES * WAVE), 0, G->s_infl, G->d_comp[slot], G->d_desc[slot], bi.n_blocks,
                       G->d_infl[slot] + CARRY_CAP, G->d_status[slot]);
    HIP_LAUNCH_OK(hipGetLastError(), "k_bgzf_inflate");
    HIP_LAUNCH_OK(hipEventRecord(G->ev_infl[slot], G->s_infl), "hipEventRecord");
    // checksums of the inflated blocks (as htslib's bgzf_read_block): a mismatch becomes the block's status.  On a stream of its own,
    // behind this batch's inflate and beside the next one's (the inflate is bound by scalar issue, the checksum by memory and
    // LDS: they overlap well).  The batch slot's block table, CRCs and status words are free again when ev_crc fires: the
    // feeder waits for it before it re-stages the slot, the caller before it reads the status words.
    HIP_LAUNCH_OK(hipStreamWaitEvent(G->s_crc, G->ev_infl[slot], 0), "hipStreamWaitEvent");
    hipLaunchKernelGGL(k_bgzf_crc, dim3((unsigned)bi.n_blocks), dim3(WAVE), 0, G->s_crc, G->d_infl[slot] + CARRY_CAP, G->d_desc[slot], G->d_crc[slot],
                       bi.n_blocks, G->d_status[slot]);
    HIP_LAUNCH_OK(hipGetLastError(), "k_bgzf_crc");
    HIP_LAUNCH_OK(hipEventRecord(G->ev_crc[slot], G->s_crc), "hipEventRecord");
    {
        std::lock_guard<std::mutex> lk(G->m);
        G->inflate_launched = kb + 1;
    }
    G->cv.notify_all();
    return true;
}

// wait until batch kb has been staged; false = there is no such batch (or the feeder failed: G->error set)
bool wait_staged(GpuDecoder *G, int kb, BatchInfo *bi) {
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(G->m);
    G->cv.wait(lk, [&] { return (int)G->staged.size() > kb || G->feeder_done; });
    G->t_wait_staged += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!G->feeder_error.empty()) { G->error = G->feeder_error; return false; }
    if ((int)G->staged.size() <= kb) return false;
    *bi = G->staged[(size_t)kb];
    return true;
}

const char *rec_error_text(int e) {
    switch (e) {
        case REC_ERR_SHORT: return "record shorter than its fixed fields";
        case REC_ERR_FIELDS: return "record fields overrun the record";
        case REC_ERR_TAG_B: return "truncated B tag";
        case REC_ERR_TAG_TYPE: return "unknown tag type";
        case REC_ERR_TAG_OVERRUN: return "tag overruns the record";
    }
    return "malformed record";
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int coral_bamgpu_open(const char *path, int32_t n_threads, int32_t rank, int32_t world, int64_t batch_bytes, void **handle,
                                 int64_t *workspace_bytes) {
    if (!path || !handle || !workspace_bytes || world < 1 || rank < 0 || rank >= world) return CORAL_ERR_ARG;
    std::unique_ptr<GpuDecoder> G(new GpuDecoder());
    G->t_start = std::chrono::steady_clock::now();
    if (!G->f.open(path, G->error) || !read_bam_header(G->f, G->D, G->ref_id, &G->hdr_bytes)) {
        set_error(G->error.empty() ? G->D.error : G->error);
        return CORAL_ERR_FORMAT;
    }
    G->rank = rank;
    G->world = world;
    G->n_threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    G->last_rank = rank == world - 1;
    G->byte_lo = rank == 0 ? 0 : G->f.size / (uint64_t)world * (uint64_t)rank;
    G->byte_hi = G->last_rank ? G->f.size : G->f.size / (uint64_t)world * (uint64_t)(rank + 1);
    G->first_block = 0;
    if (rank > 0 && !find_block(G->f, G->byte_lo, &G->first_block)) G->first_block = G->f.size;
    if (G->first_block >= G->byte_hi) G->first_block = G->f.size;          // no block starts in this range: nothing to do
    G->searching = rank > 0;
    // batch size: at most `batch_bytes` inflated, no more than the range can need.  Default 2.52 GiB = 6 x 6 912 BGZF blocks of
    // 65 280 bytes (htslib's block size): the inflate kernel keeps 27 one-wave workgroups per CU x 256 CUs resident, blocks of
    // equal size finish in rounds, and a batch that is a whole number of rounds has no part-filled last round; bigger batches
    // also mean fewer of them (1 GiB batches: 1.31 s for the 2 M-read file, 2.4 GiB ones: 1.13 s with the same kernel).  Offsets
    // inside a batch are 32-bit: CARRY_CAP + the batch must stay below 4 GiB.
    const uint64_t range = G->byte_hi > G->first_block ? G->byte_hi - G->first_block : 0;
    uint64_t cap = batch_bytes > 0 ? (uint64_t)batch_bytes : 6ull * 6912ull * 65280ull;
    if (cap > (3ull << 30) + (512ull << 20)) cap = (3ull << 30) + (512ull << 20);
    cap = std::min<uint64_t>(cap, std::max<uint64_t>(16ull << 20, (range * 6 + (64ull << 20) + 0xffff) & ~0xffffull));
    G->infl_cap = (size_t)std::max<uint64_t>(cap, 1ull << 20);
    G->comp_cap = std::max<size_t>(G->infl_cap / 2, 1u << 20);
    G->max_blocks = G->comp_cap / 28 < (1u << 22) ? std::max<size_t>(G->comp_cap / 28, 64) : (1u << 22);   // a block is >= 28 bytes
    if (G->max_blocks > G->infl_cap / 256 + 65536) G->max_blocks = G->infl_cap / 256 + 65536;
    G->rec_cap = ((size_t)CARRY_CAP + G->infl_cap) / 36 + 2;
    G->nseg_cap = ((size_t)CARRY_CAP + G->infl_cap) / (size_t)SEG_BYTES + 4;
    size_t tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (long long *)nullptr, (long long *)nullptr, (int)std::min<size_t>(G->rec_cap + 1, 0x7fffffff));
    G->scan_tmp_bytes = tmp + 256;
    carve(G.get(), nullptr, 0);
    if (const char *fb = getenv("CORAL_BAMGPU_FIRST_BATCH")) G->first_batch = std::max<uint64_t>(1u << 20, strtoull(fb, nullptr, 10));      // tuning
    G->known_start = CARRY_CAP + (long long)(rank == 0 ? G->hdr_bytes : 0);
    *workspace_bytes = (int64_t)G->ws_bytes;
    *handle = G.release();
    return CORAL_OK;
}

extern "C" int coral_bamgpu_start(void *handle, void *workspace, int64_t workspace_bytes) {
    GpuDecoder *G = (GpuDecoder *)handle;
    if (!G || !workspace || workspace_bytes < (int64_t)G->ws_bytes || (((uintptr_t)workspace) & 255)) return CORAL_ERR_ARG;
    auto bad = [&](const char *what, hipError_t e) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return CORAL_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipGetDevice(&G->device)) != hipSuccess) return bad("hipGetDevice", e);
    carve(G, workspace, (size_t)workspace_bytes);
    const auto t_alloc0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N_STAGE; ++i) {
        if ((e = hipHostMalloc((void **)&G->h_stage[i], STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) return bad("hipHostMalloc", e);
        if ((e = hipEventCreateWithFlags(&G->ev_stage[i], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", e);
    }
    for (int i = 0; i < 2; ++i) {
        if ((e = hipHostMalloc((void **)&G->h_desc[i], up256(G->max_blocks * (sizeof(BlockDesc) + 4)), hipHostMallocDefault)) != hipSuccess) return bad("hipHostMalloc", e);
        if ((e = hipEventCreateWithFlags(&G->ev_h2d[i], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&G->ev_infl[i], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&G->ev_crc[i], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&G->ev_parsed[i], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", e);
    }
    if ((e = hipStreamCreateWithFlags(&G->s_copy, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate", e);
    {   // the inflate stream gets the LOWEST priority: when a batch's inflate ends, the parse kernels of that batch (caller's
        // stream) and the next batch's inflate become runnable at the same moment; an inflate launch fills every CU with waves
        // that run for milliseconds, and the record-start search behind it took 26 ms instead of 1.2 ms (rocprofv3 timeline)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        if ((e = hipStreamCreateWithPriority(&G->s_infl, hipStreamNonBlocking, least)) != hipSuccess) return bad("hipStreamCreate", e);
    }
    if ((e = hipStreamCreateWithFlags(&G->s_crc, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate", e);
    G->t_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc0).count();
    G->feeder = std::thread(feeder_main, G);
    G->worker = std::thread(worker_main, G);
    return CORAL_OK;
}

// Parse the next batch up to the sizes of what it contributes.  out: [0] records, [1] padded CIGAR ops, [2] 1 = a batch was
// parsed (call coral_bamgpu_emit next), 0 = the file is done.
extern "C" int coral_bamgpu_next(void *handle, int64_t out[4], void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GpuDecoder *G = (GpuDecoder *)handle;
    if (!G || !out) return CORAL_ERR_ARG;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (G->have_cur) { set_error("coral_bamgpu_next: the previous batch has not been emitted"); return CORAL_ERR_ARG; }
    auto fail = [&](int code) {
        set_error(G->error);
        return code;
    };
    if (G->finished) return CORAL_OK;
    BatchInfo bi;
    if (!wait_staged(G, G->k, &bi)) {
        if (!G->error.empty()) return fail(CORAL_ERR_FORMAT);
        if (G->carry_len > 0 && !G->searching) { G->error = G->last_rank ? "truncated record at the end of the file" : "a record straddles further than the supported overhang"; return fail(CORAL_ERR_FORMAT); }
        G->finished = true;
        return CORAL_OK;
    }
    const int kb = G->k, slot = kb & 1;
    uint8_t *buf = G->d_infl[slot];
    {   // the feeder enqueues the inflate of a batch right after staging it
        std::unique_lock<std::mutex> lk(G->m);
        G->cv.wait(lk, [&] { return G->inflate_launched > kb || !G->feeder_error.empty(); });
        if (G->inflate_launched <= kb) { G->error = G->feeder_error; return fail(CORAL_ERR_HIP); }
    }
    // this batch's inflate must be complete before the parse kernels read its bytes
    if (hipStreamWaitEvent(stream, G->ev_infl[slot], 0) != hipSuccess) { G->error = "hipStreamWaitEvent failed"; return fail(CORAL_ERR_HIP); }
    const long long data_end = CARRY_CAP + (long long)bi.infl_bytes;
    const long long begin = G->searching ? (long long)CARRY_CAP - G->carry_len : std::min(G->known_start, data_end);
    const long long limit = (G->last_rank || !bi.has_limit) ? (1ll << 62) : (long long)CARRY_CAP + bi.limit_rel;
    const int seg0 = (int)(begin / SEG_BYTES);
    const int n_seg = (int)((data_end + SEG_BYTES - 1) / SEG_BYTES) - seg0;
    const int n_ref = (int)G->D.ref_names.size();
    long long res[6] = {0, 0, 0, 0, 0, 0};
    if (n_seg > 0) {
        hipLaunchKernelGGL(k_bam_find, dim3(n_seg), dim3(WAVE), 0, stream, buf, begin, data_end, limit, n_ref, seg0, n_seg, G->d_seg_first,
                           G->d_seg_land, G->d_seg_count, G->d_seg_valid);
        hipLaunchKernelGGL(k_bam_verify, dim3(1), dim3(WAVE), 0, stream, buf, G->searching ? -1ll : G->known_start, begin, data_end, limit, seg0, n_seg,
                           G->d_seg_first, G->d_seg_land, G->d_seg_count, G->d_seg_valid, G->d_seg_base, G->d_result);
    }
    int32_t status_bad = 0;
    if (n_seg > 0) {
        const auto t_gpu0 = std::chrono::steady_clock::now();
        const bool synced = hipMemcpyAsync(res, G->d_result, sizeof(res), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        G->t_wait_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gpu0).count();
        if (!synced) {
            G->error = std::string("record walk failed: ") + hipGetErrorString(hipGetLastError());
            return fail(CORAL_ERR_HIP);
        }
    } else {
        res[1] = G->known_start;
        if (hipStreamSynchronize(stream) != hipSuccess) { G->error = "hipStreamSynchronize failed"; return fail(CORAL_ERR_HIP); }
    }
    {   // every block of the batch must have inflated cleanly and have the checksum of its trailer
        if (hipEventSynchronize(G->ev_crc[slot]) != hipSuccess) { G->error = "hipEventSynchronize failed"; return fail(CORAL_ERR_HIP); }
        std::vector<int32_t> st((size_t)bi.n_blocks);
        if (hipMemcpy(st.data(), G->d_status[slot], st.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { G->error = "hipMemcpy failed"; return fail(CORAL_ERR_HIP); }
        for (int32_t s : st) if (s != 0) { status_bad = s; break; }
        if (status_bad == ERR_CRC) { G->error = "CRC32 of an inflated BGZF block differs from the one in its trailer"; return fail(CORAL_ERR_FORMAT); }
        if (status_bad) { G->error = "inflate failed (corrupt BGZF block), code " + std::to_string(status_bad); return fail(CORAL_ERR_FORMAT); }
    }
    if (res[3] == 1) { G->error = "record shorter than its fixed fields"; return fail(CORAL_ERR_FORMAT); }
    bool still_searching = false;
    if (res[3] == 2) {
        // no confirmed record start in what this byte range has seen so far (the chained check needs three records in a row): keep
        // the bytes and look again with the next batch behind them, as the host pipeline does; at the end of the range it simply
        // has no record of its own (it lies inside one record of the previous range)
        if (data_end - begin > CARRY_CAP) { G->error = "no record start found in 64 MiB at the beginning of the byte range"; return fail(CORAL_ERR_FORMAT); }
        res[0] = 0;
        res[1] = bi.last ? data_end : begin;
        res[2] = 0;
        still_searching = !bi.last;
    }
    const long long n_rec = res[0], carry_pos = res[1];
    G->cur_carry_pos = carry_pos < data_end ? carry_pos : data_end;
    G->fixups += res[4];
    if (n_rec > (long long)G->rec_cap) { G->error = "more records in a batch than its workspace holds"; return fail(CORAL_ERR_FORMAT); }
    G->searching = still_searching;
    G->cur = bi;
    G->cur_n_rec = n_rec;
    G->cur_ops = G->cur_name_bytes = G->cur_sa_bytes = 0;
    if (n_rec > 0) {
        hipLaunchKernelGGL(k_bam_starts, dim3((n_seg + WAVE - 1) / WAVE), dim3(WAVE), 0, stream, buf, seg0, n_seg, G->d_seg_first, G->d_seg_count, G->d_seg_valid,
                           G->d_seg_base, G->d_rec_start);
        (void)hipMemsetAsync(G->d_error, 0, 4, stream);
        const long long waves = n_rec + 1;
        hipLaunchKernelGGL(k_bam_meta, dim3((unsigned)waves), dim3(WAVE), 0, stream, buf, G->d_rec_start, n_rec, G->M, G->d_error);
        size_t tmp = G->scan_tmp_bytes;
        (void)hipcub::DeviceScan::ExclusiveSum(G->d_scan_tmp, tmp, G->M.pad_ops, G->d_cig_off, (int)(n_rec + 1), stream);
        tmp = G->scan_tmp_bytes;
        (void)hipcub::DeviceScan::ExclusiveSum(G->d_scan_tmp, tmp, G->M.name_len, G->d_name_off, (int)(n_rec + 1), stream);
        tmp = G->scan_tmp_bytes;
        (void)hipcub::DeviceScan::ExclusiveSum(G->d_scan_tmp, tmp, G->M.sa_len, G->d_sa_off, (int)(n_rec + 1), stream);
        long long totals[3] = {0, 0, 0};
        int32_t rec_err = 0;
        if (hipMemcpyAsync(&totals[0], G->d_cig_off + n_rec, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(&totals[1], G->d_name_off + n_rec, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(&totals[2], G->d_sa_off + n_rec, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(&rec_err, G->d_error, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            G->error = std::string("record parse failed: ") + hipGetErrorString(hipGetLastError());
            return fail(CORAL_ERR_HIP);
        }
        if (rec_err) { G->error = rec_error_text(rec_err); return fail(CORAL_ERR_FORMAT); }
        G->cur_ops = totals[0];
        G->cur_name_bytes = totals[1];
        G->cur_sa_bytes = totals[2];
    }
    // what the next batch starts with
    const bool done = res[2] != 0;
    if (done) {
        G->finished = true;
        G->carry_len = 0;
    } else if (carry_pos < data_end) {
        G->carry_len = data_end - carry_pos;
        if (G->carry_len > CARRY_CAP) { G->error = "a record larger than 64 MiB straddles two batches"; return fail(CORAL_ERR_FORMAT); }
        G->known_start = CARRY_CAP - G->carry_len;
    } else {
        G->carry_len = 0;
        G->known_start = CARRY_CAP + (carry_pos - data_end);
    }
    if (bi.last && !done) {
        if (G->carry_len > 0) { G->error = G->last_rank ? "truncated record at the end of the file" : "a record straddles further than the supported overhang"; return fail(CORAL_ERR_FORMAT); }
        G->finished = true;
    }
    G->have_cur = true;
    ++G->n_batches;
    out[0] = n_rec;
    out[1] = G->cur_ops;
    out[2] = 1;
    return CORAL_OK;
}

// Write the batch's CIGAR ops (padded SoA, `cigar_dst` device pointer with room for out[1] of coral_bamgpu_next words) and,
// if wanted, the batch-local op offsets (`cigar_off_dst`, device, n_rec + 1 int64; may be NULL — the offsets over the whole
// file are part of the host-side result), and take the batch's host-side fields in.
extern "C" int coral_bamgpu_emit(void *handle, uint32_t *cigar_dst, int64_t *cigar_off_dst, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GpuDecoder *G = (GpuDecoder *)handle;
    if (!G || !G->have_cur) return CORAL_ERR_ARG;
    auto fail = [&](int code, const std::string &msg) {
        set_error(msg);
        return code;
    };
    const int kb = G->k, slot = kb & 1;
    uint8_t *buf = G->d_infl[slot];
    const long long n = G->cur_n_rec;
    Decoded &D = G->D;
    if (n > 0) {
        if (G->cur_ops > 0 && !cigar_dst) return CORAL_ERR_ARG;
        (void)hipMemsetAsync(G->d_na_count, 0, 4, stream);
        hipLaunchKernelGGL(k_bam_emit, dim3((unsigned)n), dim3(WAVE), 0, stream, buf, G->d_rec_start, n, G->M, G->d_cig_off, G->d_name_off,
                           G->d_sa_off, cigar_dst, G->d_end, G->d_qlen, G->d_names, G->d_sa_text, G->d_na_list, G->d_na_count);
        if (cigar_off_dst && hipMemcpyAsync(cigar_off_dst, G->d_cig_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return fail(CORAL_ERR_HIP, "device copy of the op offsets failed");
    }
    // the head of a straddling record moves in front of the next batch's bytes (same stream: after the kernels above)
    if (G->carry_len > 0 && !G->finished) {
        const long long data_end = CARRY_CAP + (long long)G->cur.infl_bytes;
        if (hipMemcpyAsync(G->d_infl[slot ^ 1] + CARRY_CAP - G->carry_len, buf + data_end - G->carry_len, (size_t)G->carry_len, hipMemcpyDeviceToDevice,
                           stream) != hipSuccess)
            return fail(CORAL_ERR_HIP, "device copy of the carried bytes failed");
    }
    if (n > 0) {
        std::unique_ptr<HostJob> job(new HostJob());
        HostJob &J = *job;
        const size_t base = D.tid.size();
        J.base = base;
        J.n = n;
        auto grow = [&](std::vector<int32_t> &v) { v.resize(base + (size_t)n); return v.data() + base; };
        int32_t *tid = grow(D.tid), *pos = grow(D.pos), *end = grow(D.end), *flag = grow(D.flag), *mapq = grow(D.mapq), *qlen = grow(D.qlen),
                *has_seq = grow(D.has_seq), *nm = grow(D.nm), *n_cigar = grow(D.n_cigar);
        J.pad.resize((size_t)n);
        J.name_off.resize((size_t)n + 1);
        J.sa_off.resize((size_t)n + 1);
        J.names.resize((size_t)G->cur_name_bytes);
        J.sa_text.resize((size_t)G->cur_sa_bytes);
        int32_t na_count = 0;
        auto get = [&](void *dst, const void *src, size_t bytes) { return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess; };
        const size_t n4 = (size_t)n * 4;
        const auto t_gpu0 = std::chrono::steady_clock::now();
        bool ok = get(tid, G->M.tid, n4) && get(pos, G->M.pos, n4) && get(end, G->d_end, n4) && get(flag, G->M.flag, n4) && get(mapq, G->M.mapq, n4) &&
                  get(qlen, G->d_qlen, n4) && get(has_seq, G->M.l_seq, n4) && get(nm, G->M.nm, n4) && get(n_cigar, G->M.n_cigar, n4) &&
                  get(J.pad.data(), G->M.pad_ops, (size_t)n * 8) && get(J.name_off.data(), G->d_name_off, (size_t)(n + 1) * 8) &&
                  get(J.sa_off.data(), G->d_sa_off, (size_t)(n + 1) * 8) && get(J.names.data(), G->d_names, J.names.size()) &&
                  get(J.sa_text.data(), G->d_sa_text, J.sa_text.size()) && get(&na_count, G->d_na_count, 4);
        if (!ok || hipStreamSynchronize(stream) != hipSuccess) return fail(CORAL_ERR_HIP, std::string("copy of the batch's host fields failed: ") + hipGetErrorString(hipGetLastError()));
        G->t_wait_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gpu0).count();
        for (long long i = 0; i < n; ++i) has_seq[i] = has_seq[i] > 0 ? 1 : 0;       // (arrived as l_seq)
        if (na_count > 0) {
            // the records with a non-ACGT code, whole, in one gather + one copy (the offset arrays of the scans are free by now)
            J.na_list.resize((size_t)na_count);
            std::vector<long long> starts((size_t)n);
            if (hipMemcpy(J.na_list.data(), G->d_na_list, (size_t)na_count * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(starts.data(), G->d_rec_start, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(CORAL_ERR_HIP, "copy of the non-ACGT list failed");
            std::sort(J.na_list.begin(), J.na_list.end());
            std::vector<long long> src((size_t)na_count), dst((size_t)na_count), len((size_t)na_count);
            J.na_off.assign((size_t)na_count + 1, 0);
            for (int32_t j = 0; j < na_count; ++j) {
                const int32_t li = J.na_list[(size_t)j];
                const long long rec_end = li + 1 < n ? starts[(size_t)li + 1] : G->cur_carry_pos;
                src[(size_t)j] = starts[(size_t)li] + 4;
                len[(size_t)j] = rec_end - starts[(size_t)li] - 4;
                dst[(size_t)j] = J.na_off[(size_t)j];
                J.na_off[(size_t)j + 1] = J.na_off[(size_t)j] + len[(size_t)j];
            }
            const long long total = J.na_off[(size_t)na_count];
            J.na_raw.resize((size_t)total);
            if ((size_t)total > G->names_cap) return fail(CORAL_ERR_FORMAT, "records with non-ACGT bases exceed the batch workspace");
            if (hipMemcpy(G->d_cig_off, src.data(), (size_t)na_count * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(G->d_name_off, dst.data(), (size_t)na_count * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(G->d_sa_off, len.data(), (size_t)na_count * 8, hipMemcpyHostToDevice) != hipSuccess)
                return fail(CORAL_ERR_HIP, "upload of the non-ACGT gather list failed");
            hipLaunchKernelGGL(k_bam_gather, dim3((unsigned)na_count), dim3(WAVE), 0, stream, buf, G->d_cig_off, G->d_name_off, G->d_sa_off, (int)na_count,
                               G->d_names);
            if (hipMemcpyAsync(J.na_raw.data(), G->d_names, (size_t)total, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
                return fail(CORAL_ERR_HIP, "copy of the records with non-ACGT bases failed");
            G->na_records += na_count;
        }
        {
            std::lock_guard<std::mutex> lk(G->wm);
            G->jobs.push_back(std::move(job));
        }
        G->wcv.notify_all();
    } else if (hipStreamSynchronize(stream) != hipSuccess) {
        return fail(CORAL_ERR_HIP, "hipStreamSynchronize failed");
    }
    // this buffer may be inflated into again (batch k + 2) once everything above has run
    if (hipEventRecord(G->ev_parsed[slot], stream) != hipSuccess) return fail(CORAL_ERR_HIP, "hipEventRecord failed");
    {
        std::lock_guard<std::mutex> lk(G->m);
        ++G->emitted;
    }
    G->cv.notify_all();
    G->have_cur = false;
    ++G->k;
    if (G->finished) D.seconds = G->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - G->t_start).count();
    return CORAL_OK;
}

// The host-side half of the result as a handle for coral_bam_decode_sizes / coral_bam_decode_fill / coral_bam_decode_stats
// (cigar: pass NULL to coral_bam_decode_fill — the ops are on the device).  Owned by the GPU decoder: do not close it.
extern "C" int coral_bamgpu_host(void *handle, void **decoded) {
    GpuDecoder *G = (GpuDecoder *)handle;
    if (!G || !decoded) return CORAL_ERR_ARG;
    {   // the worker must have taken every batch in
        std::unique_lock<std::mutex> lk(G->wm);
        G->wcv.wait(lk, [&] { return G->jobs.empty() && !G->worker_busy; });
        if (!G->worker_error.empty()) {
            set_error(G->worker_error);
            return CORAL_ERR_FORMAT;
        }
    }
    G->D.seconds = G->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - G->t_start).count();
    *decoded = &G->D;
    return CORAL_OK;
}

// stats: [0] batches, [1] segments whose speculative start was replaced by the exact walk, [2] records fetched for the
// non-ACGT list, [3] batch capacity (inflated bytes); seconds: [0] total, [1] host-side field handling (worker thread),
// [2] file reads of the feeder, [3] pinned buffers / streams set-up, [4] caller waiting for the feeder, [5] caller waiting for the GPU
extern "C" int coral_bamgpu_stats(void *handle, int64_t stats[4], double seconds[6]) {
    GpuDecoder *G = (GpuDecoder *)handle;
    if (!G || !stats || !seconds) return CORAL_ERR_ARG;
    stats[0] = G->n_batches; stats[1] = G->fixups; stats[2] = G->na_records; stats[3] = (int64_t)G->infl_cap;
    seconds[0] = G->seconds; seconds[1] = G->host_seconds; seconds[2] = G->t_read; seconds[3] = G->t_alloc; seconds[4] = G->t_wait_staged;
    seconds[5] = G->t_wait_gpu;
    return CORAL_OK;
}

extern "C" int coral_bamgpu_close(void *handle) {
    delete (GpuDecoder *)handle;
    return CORAL_OK;
}

// One BGZF-style inflate launch on caller-provided buffers (tests; tools/bench_inflate.py): `desc` = n_blocks x 4 uint32
// (src_off, src_len, dst_off, isize) on the device, `comp` readable COMP_SLACK bytes beyond its last stream.
extern "C" int coral_bgzf_inflate(const uint8_t *comp, const uint32_t *desc, int32_t n_blocks, uint8_t *out, int32_t *status, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks < 0 || (n_blocks > 0 && (!comp || !desc || !out || !status))) return CORAL_ERR_ARG;
    if (n_blocks == 0) return CORAL_OK;
    const dim3 grid((n_blocks + INFL_WAVES - 1) / INFL_WAVES), block(INFL_WAVES * WAVE);
#define CORAL_INFL_LAUNCH(M) hipLaunchKernelGGL(k_bgzf_inflate<M>, grid, block, 0, stream, comp, (const BlockDesc *)desc, n_blocks, out, status)
#ifdef CORAL_EXPERIMENTS
    // timing experiments (tools/bench_inflate.py with a -DCORAL_EXPERIMENTS build): variants that skip parts of the work and
    // produce wrong output by design; not compiled into the product library
    const char *ab = getenv("CORAL_INFLATE_ABLATE");
    switch (ab ? atoi(ab) : 0) {
        case 0: CORAL_INFL_LAUNCH(0); break;
        case 1: CORAL_INFL_LAUNCH(1); break;
        case 2: CORAL_INFL_LAUNCH(2); break;
        case 4: CORAL_INFL_LAUNCH(4); break;
        case 12: CORAL_INFL_LAUNCH(12); break;
        case 13: CORAL_INFL_LAUNCH(13); break;
        default: set_error("CORAL_INFLATE_ABLATE: unknown mode"); return CORAL_ERR_ARG;
    }
#else
    CORAL_INFL_LAUNCH(0);
#endif
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("k_bgzf_inflate: ") + hipGetErrorString(e)); return CORAL_ERR_HIP; }
    return CORAL_OK;
}
