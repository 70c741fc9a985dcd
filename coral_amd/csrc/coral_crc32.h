// coral_crc32.h — CRC-32 (the gzip / BGZF trailer checksum, reflected polynomial 0xedb88320) in pieces that can be computed
// independently and put together: the 64 lanes of a wave each take a contiguous chunk of a BGZF block's inflated bytes, and
// the chunks' remainders are combined with the "multiply by x^(8 n) modulo P" operator (the structure of zlib's
// crc32_combine, restated; zlib itself is only the oracle of tests/test_crc32.py).  Host + device.
//
// With R0(M) the register after feeding M into a ZERO register (no pre- or post-conditioning):
//   R0(A || B) = shift(R0(A), len(B)) ^ R0(B)          shift(v, n) = v * x^(8 n) mod P  (n zero bytes fed behind v)
//   crc32(M)   = ~( R0(M) ^ shift(0xffffffff, len(M)) )
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CORAL_CRC_HD __host__ __device__ __forceinline__
#else
#define CORAL_CRC_HD inline
#endif

namespace coral_crc {

enum : uint32_t { POLY = 0xedb88320u };

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
CORAL_CRC_HD uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ POLY : b >> 1;
    }
    return p;
}

// x^(8 n) mod P by square and multiply (x^8 in the reflected representation is bit 23)
CORAL_CRC_HD uint32_t x8n_modp(uint32_t n) {
    uint32_t result = 1u << 31;            // x^0
    uint32_t sq = 1u << 23;                // x^8
    while (n) {
        if (n & 1u) result = multmodp(result, sq);
        sq = multmodp(sq, sq);
        n >>= 1;
    }
    return result;
}

CORAL_CRC_HD uint32_t shift(uint32_t v, uint32_t n_bytes) { return multmodp(x8n_modp(n_bytes), v); }

// one table entry: the register after feeding byte value i into a zero register
CORAL_CRC_HD uint32_t table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ POLY : c >> 1;
    return c;
}

// R0 of a byte string, byte at a time, with a 256-entry table (LDS on the device)
CORAL_CRC_HD uint32_t raw(const uint32_t *table, const uint8_t *p, uint32_t n, uint32_t r = 0) {
    for (uint32_t k = 0; k < n; ++k) r = table[(r ^ p[k]) & 0xffu] ^ (r >> 8);
    return r;
}

}  // namespace coral_crc
