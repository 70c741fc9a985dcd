// Lab: the middle of the blocks-per-wave continuum — EIGHT BGZF blocks per wave, 8 lanes per block.  The 8 lanes of a group decode
// their block's symbols redundantly (the same values in all 8: group-uniform work is vector work, eight blocks share one
// instruction stream under divergence) and split what is parallel: match copies (16 bytes per step).  Per group in LDS: a 10-bit
// literal/length table, an 8-bit distance table and a 1 KiB ring of recent output (3.5 KiB; 28 KiB per wave).
// Stand-alone like inflate_simt.hip: every block is checked against zlib.
//     hipcc --offload-arch=gfx950 -O3 -o inflate_g8 inflate_g8.hip -lz && ./inflate_g8 file.bam [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#ifndef GL
#define GL 8                        // lanes per block
#endif
#define GROUPS (64 / GL)
#define LLB 10
#define DB 8
#define RING 1024
#define GROUP_BYTES ((1 << LLB) * 2 + (1 << DB) * 2 + RING)
#define COPY_CHUNK (2 * GL)

struct Desc { uint32_t src_off, src_len, dst_off, isize; };
struct Scratch {                    // per lane, global memory: touched by the header parse, the table build and the rare long codes
    uint16_t cnt[16], dcnt[16];
    uint16_t sym[288], dsym[32];
    uint8_t lens[320];
};

enum { NL = 16, LONGCODE = 0x0010 };
__device__ __forceinline__ uint32_t ll_entry(uint32_t s, uint32_t nb) {      // as coral_inflate_core.h
    if (s < 256) return nb | (s << 8);
    if (s == 256) return LONGCODE;                                            // end of block: left to the canonical decode
    const uint32_t i = s - 257;
    if (i > 28) return LONGCODE;
    if (i < 8) return nb | NL | (i << 8);
    if (i == 28) return nb | NL | (255u << 8);
    const uint32_t e = (i >> 2) - 1;
    return nb | NL | (e << 5) | (((4 + (i & 3)) << e) << 8);
}

struct Lane {
    const uint8_t *in;
    int in_len, ip;
    uint64_t bb;
    int bc;
    uint8_t *out;
    int cap, o;
    int err;
    __device__ __forceinline__ void refill() {
        if (bc <= 32) {
            uint32_t w;
            __builtin_memcpy(&w, in + ip, 4);        // (the compressed buffer is readable a few KiB beyond its end)
            bb |= (uint64_t)w << bc;
            bc += 32;
            ip += 4;
            if (ip > in_len + 12) err = 8;
        }
    }
    __device__ __forceinline__ uint32_t bits(int n) {
        const uint32_t v = (uint32_t)bb & ~(~0u << n);
        bb >>= n;
        bc -= n;
        return v;
    }
    __device__ __forceinline__ uint32_t bits0(int n) { return n ? bits(n) : 0u; }
};

// canonical decode, bit by bit (long codes, the end-of-block code, the code-length code)
__device__ __forceinline__ int decode_slow(Lane &L, const uint16_t *cnt, const uint16_t *sym) {
    uint32_t code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (uint32_t)(L.bb >> (l - 1)) & 1u;
        const uint32_t c = cnt[l];
        if (code - first < c) {
            L.bb >>= l;
            L.bc -= l;
            return sym[index + (code - first)];
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return -1;
}

// counts, sorted symbols and the primary table (16-bit entries) of one code; returns false for an over-subscribed code
__device__ __forceinline__ bool build(const uint8_t *lens, int n, uint16_t *cnt, uint16_t *sym, uint16_t *table, int bits_, bool is_dist) {
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int s = 0; s < n; ++s) cnt[lens[s]]++;
    int left = 1;
    uint32_t offs[16], next[16];
    uint32_t code = 0, at = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)cnt[l];
        if (left < 0) return false;
        next[l] = code;
        code = (code + cnt[l]) << 1;
        offs[l] = at;
        at += cnt[l];
    }
    const int size = 1 << bits_;
    for (int i = 0; i < size; ++i) table[i] = is_dist ? 0 : LONGCODE;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        sym[offs[l]++] = (uint16_t)s;
        const uint32_t c = next[l]++;
        if (l <= bits_) {
            const uint32_t r = __brev(c) >> (32 - l);
            const uint32_t e = is_dist ? (s > 29 ? 0u : ((uint32_t)l | ((uint32_t)s << 4))) : ll_entry((uint32_t)s, (uint32_t)l);
            for (uint32_t k = r; k < (uint32_t)size; k += 1u << l) table[k] = (uint16_t)e;
        }
    }
    return true;
}

__global__ __launch_bounds__(64) void k_inflate_g8(const uint8_t *__restrict__ comp, const Desc *__restrict__ desc, int n_blocks, uint8_t *__restrict__ out,
                                                      int *__restrict__ status, Scratch *__restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[GROUPS * GROUP_BYTES];
    const int lane = threadIdx.x, grp = lane / GL, gl = lane % GL;
    const int b = blockIdx.x * GROUPS + grp;
    uint16_t *ll = reinterpret_cast<uint16_t *>(lds + grp * GROUP_BYTES), *dt = ll + (1 << LLB);
    uint8_t *ring = reinterpret_cast<uint8_t *>(dt + (1 << DB));
    Scratch &S = scratch[(size_t)blockIdx.x * GROUPS + grp];
    const float glf = (float)gl + 0.5f;
    Lane L;
    L.err = 0;
    bool finished = b >= n_blocks;
    if (!finished) {
        const Desc d = desc[b];
        L.in = comp + d.src_off;
        L.in_len = (int)d.src_len;
        L.ip = 0;
        L.bb = 0;
        L.bc = 0;
        L.out = out + d.dst_off;
        L.cap = (int)d.isize;
        L.o = 0;
        if (d.isize == 0) finished = true;
    }
    int copy_left = 0, copy_dist = 0, copy_done = 0, copy_from = 0;
    bool last = false;
    while (__ballot(!finished) != 0ull) {                 // one DEFLATE block of every unfinished lane per round
        bool active = false;
        if (!finished) {
            L.refill();
            last = L.bits(1) != 0;
            const uint32_t type = L.bits(2);
            if (type == 0) {                               // stored
                L.bits(L.bc & 7);
                L.refill();
                const uint32_t len = L.bits(16), nlen = L.bits(16);
                if ((len ^ 0xffffu) != nlen || L.o + (int)len > L.cap) L.err = 2;
                else {
                    // back to byte positions: the bit buffer holds bc / 8 whole bytes of the stream
                    int p = L.ip - L.bc / 8;
                    for (uint32_t k = (uint32_t)gl; k < len; k += GL) {
                        const uint8_t v = L.in[p + (int)k];
                        L.out[L.o + (int)k] = v;
                        ring[(L.o + (int)k) & (RING - 1)] = v;
                    }
                    L.o += (int)len;
                    L.ip = p + (int)len;
                    L.bb = 0;
                    L.bc = 0;
                    if (L.ip > L.in_len) L.err = 8;
                }
            } else if (type == 3) {
                L.err = 1;
            } else {
                int n_ll = 288, n_d = 30;
                if (type == 1) {
                    for (int s = 0; s < 320; ++s) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                } else {
                    L.refill();
                    n_ll = (int)L.bits(5) + 257;
                    n_d = (int)L.bits(5) + 1;
                    const int hclen = (int)L.bits(4) + 4;
                    if (n_ll > 286 || n_d > 30) L.err = 3;
                    uint8_t cl[19];
                    for (int i = 0; i < 19; ++i) cl[i] = 0;
                    for (int i = 0; i < hclen; ++i) {
                        L.refill();
                        const int k = i - 4;
                        const int pos = i < 3 ? 16 + i : i == 3 ? 0 : (k & 1) ? 7 - (k >> 1) : 8 + (k >> 1);
                        cl[pos] = (uint8_t)L.bits(3);
                    }
                    // the code-length code, canonical decode only (reuses the distance slots of the scratch)
                    for (int l = 0; l < 16; ++l) S.dcnt[l] = 0;
                    for (int s = 0; s < 19; ++s) S.dcnt[cl[s]]++;
                    {
                        int left = 1;
                        uint32_t at[16], a = 0;
                        for (int l = 1; l <= 15; ++l) { left = (left << 1) - (int)S.dcnt[l]; at[l] = a; a += S.dcnt[l]; }
                        if (left < 0) L.err = 4;
                        for (int s = 0; s < 19; ++s) if (cl[s]) S.dsym[at[cl[s]]++] = (uint16_t)s;
                    }
                    int i = 0;
                    uint32_t prev = 0;
                    const int total = n_ll + n_d;
                    while (i < total && !L.err) {
                        L.refill();
                        const int s = decode_slow(L, S.dcnt, S.dsym);
                        if (s < 0) { L.err = 3; break; }
                        if (s < 16) { S.lens[i++] = (uint8_t)s; prev = (uint32_t)s; continue; }
                        uint32_t rep, val;
                        if (s == 16) { if (i == 0) { L.err = 3; break; } rep = 3 + L.bits(2); val = prev; }
                        else if (s == 17) { rep = 3 + L.bits(3); val = 0; }
                        else { rep = 11 + L.bits(7); val = 0; }
                        if (i + (int)rep > total) { L.err = 3; break; }
                        for (uint32_t k = 0; k < rep; ++k) S.lens[i + (int)k] = (uint8_t)val;
                        prev = val;
                        i += (int)rep;
                    }
                    if (!L.err && S.lens[256] == 0) L.err = 3;
                }
                if (!L.err) {
                    if (!build(S.lens, n_ll, S.cnt, S.sym, ll, LLB, false) || !build(S.lens + n_ll, n_d, S.dcnt, S.dsym, dt, DB, true)) L.err = 4;
                }
                active = !L.err;
            }
            if (L.err || (!active && last)) finished = true;
        }
        copy_left = 0;
        while (__ballot(active) != 0ull) {                 // one step of every active lane per round: a symbol, or a chunk of a pending copy
            if (active) {
                if (copy_left > 0) {
                    // 16 bytes of the pending match per step, two per lane: byte m of the match = byte (m mod dist) of the dist bytes
                    // in front of it (from the ring when they are still there, from global memory otherwise)
                    const int nn = copy_left < COPY_CHUNK ? copy_left : COPY_CHUNK;
                    const float rd = __builtin_amdgcn_rcpf((float)copy_dist);
                    for (int j = 0; j < COPY_CHUNK / GL; ++j) {
                        const int k = gl + GL * j;                  // offset inside the chunk
                        if (k < nn) {
                            const int m = copy_done + k;            // offset inside the match
                            const int q = (int)(((float)m + 0.5f) * rd);
                            const int src = copy_from - copy_dist + (copy_dist < COPY_CHUNK ? m - q * copy_dist : m);
                            const uint8_t v = copy_dist <= RING - 2 * COPY_CHUNK ? ring[src & (RING - 1)] : L.out[src];
                            L.out[copy_from + m] = v;
                            ring[(copy_from + m) & (RING - 1)] = v;
                        }
                    }
                    (void)glf;
                    L.o += nn;
                    copy_done += nn;
                    copy_left -= nn;
                } else {
                    L.refill();
                    uint32_t e = ll[(uint32_t)L.bb & ((1u << LLB) - 1u)];
                    bool have_len = false;
                    uint32_t len = 0;
                    if (!(e & NL)) {
                        L.bits((int)(e & 15u));
                        if (L.o < L.cap) { ring[L.o & (RING - 1)] = (uint8_t)(e >> 8); L.out[L.o++] = (uint8_t)(e >> 8); }
                        else L.err = 7;
                    } else if ((e & 15u) == 0) {
                        const int s = decode_slow(L, S.cnt, S.sym);
                        if (s < 0) L.err = 5;
                        else if (s < 256) { if (L.o < L.cap) { ring[L.o & (RING - 1)] = (uint8_t)s; L.out[L.o++] = (uint8_t)s; } else L.err = 7; }
                        else if (s == 256) { active = false; if (last) finished = true; }
                        else {
                            const uint32_t x = ll_entry((uint32_t)s, 1);
                            if (x == LONGCODE) L.err = 5;
                            else { len = 3u + (x >> 8) + L.bits0((int)((x >> 5) & 7u)); have_len = true; }
                        }
                    } else {
                        const int nb = (int)(e & 15u), xb = (int)((e >> 5) & 7u);
                        L.bits(nb);
                        len = 3u + (e >> 8) + L.bits0(xb);
                        have_len = true;
                    }
                    if (have_len) {
                        L.refill();
                        const uint32_t d = dt[(uint32_t)L.bb & ((1u << DB) - 1u)];
                        int ds;
                        if ((d & 15u) == 0) ds = decode_slow(L, S.dcnt, S.dsym);
                        else { L.bits((int)(d & 15u)); ds = (int)(d >> 4); }
                        if (ds < 0 || ds > 29) L.err = 5;
                        else {
                            const int ex = ds < 4 ? 0 : (ds >> 1) - 1;
                            const uint32_t dist = (ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << ex)) + L.bits0(ex);
                            if ((int)dist > L.o) L.err = 6;
                            else if (L.o + (int)len > L.cap) L.err = 7;
                            else { copy_left = (int)len; copy_dist = (int)dist; copy_done = 0; copy_from = L.o; }
                        }
                    }
                }
                if (L.err) { active = false; finished = true; }
            }
        }
    }
    if (b < n_blocks && gl == 0) status[b] = L.err ? L.err : (L.o == L.cap ? 0 : 9);
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [iters]\n", argv[0]); return 2; }
    const int iters = argc > 2 ? atoi(argv[2]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw((size_t)sz + 8192, 0);
    if (fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) return 2;
    fclose(f);
    std::vector<Desc> desc;
    uint64_t out_off = 0;
    for (long at = 0; at + 18 <= sz;) {
        const uint32_t xlen = raw[at + 10] | raw[at + 11] << 8, bsize = (raw[at + 16] | raw[at + 17] << 8) + 1u;
        uint32_t isize;
        memcpy(&isize, &raw[at + bsize - 4], 4);
        desc.push_back({(uint32_t)at + 12 + xlen, bsize - 12 - xlen - 8, (uint32_t)out_off, isize});
        out_off += isize;
        at += bsize;
        if (out_off > 3500000000ull) break;                // (32-bit offsets in this lab program)
    }
    const int n = (int)desc.size();
    printf("%d blocks, %.1f MB compressed, %.1f MB inflated\n", n, sz / 1e6, out_off / 1e6);
    uint8_t *d_comp, *d_out;
    Desc *d_desc;
    int *d_status;
    Scratch *d_scratch;
    const int groups = (n + GROUPS - 1) / GROUPS;
    hipMalloc(&d_comp, raw.size());
    hipMalloc(&d_out, out_off + 4096);
    hipMalloc(&d_desc, sizeof(Desc) * (size_t)n);
    hipMalloc(&d_status, 4 * (size_t)n);
    hipMalloc(&d_scratch, sizeof(Scratch) * (size_t)groups * GROUPS);
    hipMemcpy(d_comp, raw.data(), raw.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_desc, desc.data(), sizeof(Desc) * (size_t)n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it <= iters; ++it) {
        hipMemset(d_status, 0xff, 4 * (size_t)n);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_inflate_g8, dim3(groups), dim3(64), 0, 0, d_comp, d_desc, n, d_out, d_status, d_scratch);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it) { printf("launch %d: %.3f ms\n", it, ms); if (ms < best) best = ms; }
    }
    std::vector<int> status((size_t)n);
    std::vector<uint8_t> got(out_off);
    hipMemcpy(status.data(), d_status, 4 * (size_t)n, hipMemcpyDeviceToHost);
    hipMemcpy(got.data(), d_out, out_off, hipMemcpyDeviceToHost);
    int bad = 0;
    std::vector<uint8_t> ref(70000);
    for (int b = 0; b < n; ++b) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        inflateInit2(&zs, -15);
        zs.next_in = raw.data() + desc[b].src_off;
        zs.avail_in = desc[b].src_len;
        zs.next_out = ref.data();
        zs.avail_out = 70000;
        const int rc = desc[b].isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || status[b] != 0 || memcmp(ref.data(), got.data() + desc[b].dst_off, desc[b].isize) != 0) {
            if (bad++ < 5) printf("block %d differs (status %d, zlib rc %d)\n", b, status[b], rc);
        }
    }
    printf("checked %d blocks against zlib: %d bad\n", n, bad);
    printf("8-lanes-per-block inflate: best %.3f ms = %.1f GB/s of output\n", best, out_off / best / 1e6);
    return bad != 0;
}
