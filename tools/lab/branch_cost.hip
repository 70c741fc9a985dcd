// Lab: what does a wave pay for (a) a dependent scalar instruction, (b) a taken scalar branch, (c) an LDS lookup + readfirstlane
// round trip?  One wave per CU-ish, timed with s_memtime.    hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 4096
__global__ void k(uint32_t seed, long long *out, uint32_t *sink) {
    __shared__ uint16_t tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (uint16_t)((i * 2654435761u >> 7) | 1);
    __syncthreads();
    uint32_t x = __builtin_amdgcn_readfirstlane(seed);
    long long t0, t1, t2, t3, t4;
    // (a) dependent scalar chain: 8 SALU per iteration
    t0 = __builtin_readcyclecounter();
#pragma clang loop unroll(disable)
    for (int i = 0; i < N; ++i) {
        x = x * 5 + 1; x ^= x >> 3; x += 7; x ^= x << 2; x = x * 3 + i; x ^= x >> 5; x += 11; x ^= x << 1;
    }
    t1 = __builtin_readcyclecounter();
    // (b) the same with 4 data-dependent taken / not taken branches per iteration (asm volatile labels keep the blocks apart)
#pragma clang loop unroll(disable)
    for (int i = 0; i < N; ++i) {
        x = x * 5 + 1;
        if (x & 1) { x ^= x >> 3; asm volatile("s_nop 0"); } else { x += 9; asm volatile("s_nop 1"); }
        x += 7;
        if (x & 2) { x ^= x << 2; asm volatile("s_nop 0"); } else { x += 5; asm volatile("s_nop 1"); }
        x = x * 3 + i;
        if (x & 4) { x ^= x >> 5; asm volatile("s_nop 0"); } else { x += 3; asm volatile("s_nop 1"); }
        x += 11;
        if (x & 8) { x ^= x << 1; asm volatile("s_nop 0"); } else { x += 1; asm volatile("s_nop 1"); }
    }
    t2 = __builtin_readcyclecounter();
    // (c) LDS lookup round trip: index from a scalar, read, readfirstlane, feed back
#pragma clang loop unroll(disable)
    for (int i = 0; i < N; ++i) {
        uint32_t v = x;
        asm("" : "+v"(v));
        const uint32_t e = __builtin_amdgcn_readfirstlane(tab[v & 1023]);
        x = (x >> (e & 7)) ^ (e << 9) ^ i;
    }
    t3 = __builtin_readcyclecounter();
    // (d) (c) plus one literal-style LDS byte write of all lanes to one address
    __shared__ uint8_t ring[4096];
#pragma clang loop unroll(disable)
    for (int i = 0; i < N; ++i) {
        uint32_t v = x;
        asm("" : "+v"(v));
        const uint32_t ev = tab[v & 1023];
        const uint32_t e = __builtin_amdgcn_readfirstlane(ev);
        uint32_t a = i;
        asm("" : "+v"(a));
        ring[a & 4095] = (uint8_t)(ev >> 8);
        x = (x >> (e & 7)) ^ (e << 9) ^ i;
    }
    t4 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t2; out[blockIdx.x * 4 + 3] = t4 - t3;
        sink[blockIdx.x] = x + ring[x & 4095];
    }
}
int main() {
    long long *d; uint32_t *s; const int B = 8;
    hipMalloc(&d, B * 4 * 8); hipMalloc(&s, B * 4);
    for (int waves = 1; waves <= 2; ++waves) {
        hipLaunchKernelGGL(k, dim3(waves == 1 ? 1 : 256 * 16), dim3(64), 0, 0, 12345u, d, s);
        hipDeviceSynchronize();
        long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("%s: per iteration (cycle-counter ticks): scalar chain of 8 %.1f | with 4 branches %.1f | LDS lookup round trip %.1f | + LDS byte write %.1f\n",
               waves == 1 ? "one wave alone" : "4096 waves (16 per CU)", h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N);
    }
    return 0;
}
