// Host build of the DEFLATE core of the GPU BGZF path (coral_amd/csrc/coral_inflate_core.h) for tests/test_inflate_core.py:
// the decode logic (header parsing, canonical tables, symbol loop, stored / fixed blocks, error paths) is the code the device
// runs; only the backend differs (lanes are loops, memory is plain arrays).  Test infrastructure, never part of libcoral_hip.so.
#include <string.h>

#include "../../coral_amd/csrc/coral_inflate_core.h"

using namespace coral_inflate;

template <bool PAIRED>
struct HostWaveT {
    static constexpr bool paired_literals = PAIRED;
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
    // paired loop: output beyond the capacity is dropped and flagged, as the device backend does (its ring absorbs it)
    bool over = false;
    uint32_t vec(uint32_t x) const { return x; }
    void clamp() {}
    void put_literal(uint32_t b) {
        if (o >= cap) { over = true; return; }
        out[o++] = (uint8_t)b;
    }
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

template <bool PAIRED>
static int run(const uint8_t *src, long long n, uint8_t *out, int cap, int *produced) {
    static Tables T;
    HostWaveT<PAIRED> w;
    w.src = src; w.src_len = n; w.out = out; w.cap = cap;
    Inflater<HostWaveT<PAIRED>> inf(w, &T);
    const int rc = inf.run();
    *produced = w.o;
    return rc;
}

// paired = 1: the symbol loop the device runs (Inflater::codes_paired); 0: the plain loop
extern "C" int coral_test_inflate(const uint8_t *src, long long n, uint8_t *out, int cap, int *produced, int paired) {
    return paired ? run<true>(src, n, out, cap, produced) : run<false>(src, n, out, cap, produced);
}
