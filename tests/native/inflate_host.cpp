// Host build of the DEFLATE core of the GPU BGZF path (coral_amd/csrc/coral_inflate_core.h) for tests/test_inflate_core.py:
// the decode logic (header parsing, canonical tables, symbol loop, stored / fixed blocks, error paths) is the code the device
// runs; only the backend differs (lanes are loops, memory is plain arrays).  Test infrastructure, never part of libcoral_hip.so.
#include <string.h>

#include "../../coral_amd/csrc/coral_inflate_core.h"

using namespace coral_inflate;

template <bool VECTOR, bool FAST = false>
struct HostWaveT {
    static constexpr bool vector_loop = VECTOR;
    // FAST: a stand-in for the device backend's hand-written fast path (DevWaveT::fast in coral_bamgpu.hip) with the same
    // contract — it decodes table-hit symbols itself and comes back in state 0 / 1 / 2 (see Inflater::codes_vector) — leaving at
    // pseudo-random points as well, so that every way of resuming the general loop is exercised against zlib here on the host.
    static constexpr bool has_fast = FAST;
    uint32_t rng = 2463534242u;
    bool coin(int one_in) {
        rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
        return rng % (uint32_t)one_in == 0;
    }
    int fast(uint64_t &bb, int &bc, long long &dwords, const Tables *T, uint32_t &lenv, uint32_t &distv) {
        for (;;) {
            if (over || coin(23)) return 0;
            if (bc <= 32) {
                if (coin(5)) return 0;
                bb |= (uint64_t)next_dword() << bc;
                bc += 32;
                ++dwords;
            }
            uint32_t e = T->ll[(uint32_t)bb & ((1u << LL_BITS) - 1u)];
            if (!(e & LL_NOT_LITERAL)) {
                bb >>= e & 15u; bc -= (int)(e & 15u);
                lit(e >> 8);
                e = T->ll[(uint32_t)bb & ((1u << LL_BITS) - 1u)];
                if (!(e & LL_NOT_LITERAL)) {
                    bb >>= e & 15u; bc -= (int)(e & 15u);
                    lit(e >> 8);
                    continue;
                }
            }
            if ((e & 15u) == 0) return 0;
            {
                const uint32_t nb = e & 15u, xb = (e >> 5) & 7u;
                lenv = 3u + (e >> 8) + bfe((uint32_t)bb, nb, xb);
                bb >>= nb + xb; bc -= (int)(nb + xb);
            }
            if (bc <= 32) {
                if (coin(3)) return 1;
                bb |= (uint64_t)next_dword() << bc;
                bc += 32;
                ++dwords;
            }
            const uint32_t d = T->dt[(uint32_t)bb & ((1u << D_BITS) - 1u)];
            if ((d & 15u) == 0 || coin(11)) return 1;
            {
                const uint32_t nb = d & 15u, xb = (d >> 8) & 15u;
                distv = (d >> 16) + bfe((uint32_t)bb, nb, xb);
                bb >>= nb + xb; bc -= (int)(nb + xb);
            }
            if (lenv > 64u || distv > (uint32_t)o || (int)lenv > cap - o || coin(7)) return 2;
            match(lenv, distv);
        }
    }
    const uint8_t *src;
    long long src_len;
    uint8_t *out;
    int cap;
    int o = 0;
    long long win0 = 0;       // byte position of dword 0 of the current window
    long long pulled = 0;     // dwords pulled from the current window
    uint32_t uni(uint32_t x) const { return x; }
    uint32_t next_dword() {
        uint32_t v = 0;
        const long long p = win0 + 4 * pulled;
        for (int k = 0; k < 4; ++k)
            if (p + k < src_len) v |= (uint32_t)src[p + k] << (8 * k);
        ++pulled;
        return v;
    }
    bool input_exhausted() const { return win0 + 4 * pulled > src_len + 8; }
    // the vector loop's backend operations, with the device backend's error behaviour: nothing is tested per symbol — a bad
    // length / distance is clamped and remembered, output beyond the capacity is dropped and raises `over`
    bool over = false, bad_ = false;
    uint32_t vec(uint32_t x) const { return x; }
    uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) const { return width ? (x >> off) & (~0u >> (32 - width)) : 0u; }
    int err_ = 0;
    bool needs_attention() const { return over; }
    bool attention() { return !over; }
    void fail(int code) { if (!err_) err_ = code; over = true; }
    bool failed() const { return over || bad_ || err_ != 0; }
    int error_code() const { return err_ ? err_ : bad_ ? (int)ERR_DISTANCE : (int)ERR_OVERFLOW; }
    void lit(uint32_t b) {
        if (o >= cap) { over = true; return; }
        out[o++] = (uint8_t)b;
    }
    void match(uint32_t len, uint32_t dist) {
        if (dist > (uint32_t)o) { bad_ = true; return; }
        if ((int)len > cap - o) { over = true; len = (uint32_t)(cap - o); }
        for (uint32_t k = 0; k < len; ++k) out[o + k] = out[o + k - dist];
        o += (int)len;
        if (input_exhausted()) over = true;
    }
    void put_literal(uint32_t b) { lit(b); }
    bool copy_match(int len, int dist) {
        if (dist > o) return false;
        for (int k = 0; k < len; ++k) out[o + k] = out[o + k - dist];
        o += len;
        return true;
    }
    bool copy_stored(long long dwords, uint32_t n) {
        const long long from = win0 + 4 * dwords;
        if (from + (long long)n > src_len) return false;
        memcpy(out + o, src + from, n);
        o += (int)n;
        return true;
    }
    uint32_t reset_input_after_stored(long long dwords, uint32_t n) {
        win0 = win0 + 4 * dwords + n;
        pulled = 0;
        return 0;
    }
    int produced() const { return o; }
    int capacity() const { return cap; }
    void add_count(uint32_t *c) { ++*c; }
    void fence() {}
};

template <bool VECTOR, bool FAST = false>
static int run(const uint8_t *src, long long n, uint8_t *out, int cap, int *produced) {
    static Tables T;
    HostWaveT<VECTOR, FAST> w;
    w.src = src; w.src_len = n; w.out = out; w.cap = cap;
    Inflater<HostWaveT<VECTOR, FAST>> inf(w, &T);
    int rc = inf.run();
    if (rc == OK && w.failed()) rc = w.error_code();
    *produced = w.o;
    return rc;
}

// paired = 1: the symbol loop the device runs (Inflater::codes_vector); 2: the same with a fast path in front of it that
// comes back in every state (what the device's hand-written loop does); 0: the plain loop
extern "C" int coral_test_inflate(const uint8_t *src, long long n, uint8_t *out, int cap, int *produced, int paired) {
    return paired == 2 ? run<true, true>(src, n, out, cap, produced) : paired ? run<true>(src, n, out, cap, produced) : run<false>(src, n, out, cap, produced);
}
