// Host build of coral_amd/csrc/coral_crc32.h for tests/test_crc32.py: the CRC-32 of a byte string put together from `chunks`
// independently computed pieces, exactly as k_bgzf_crc does it with 64 lanes.  Test infrastructure.
#include "../../coral_amd/csrc/coral_crc32.h"

extern "C" uint32_t coral_test_crc32(const uint8_t *p, uint32_t n, uint32_t chunks) {
    using namespace coral_crc;
    uint32_t table[256];
    for (uint32_t i = 0; i < 256; ++i) table[i] = table_entry(i);
    const uint32_t per = (((n + chunks - 1) / chunks) + 3u) & ~3u;
    uint32_t total = shift(0xffffffffu, n);
    for (uint32_t l = 0; l < chunks; ++l) {
        const uint32_t a = l * per < n ? l * per : n, e = a + per < n ? a + per : n;
        total ^= shift(raw(table, p + a, e - a), n - e);
    }
    return ~total;
}
