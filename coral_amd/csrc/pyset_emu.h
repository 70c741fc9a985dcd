// pyset_emu.h — replay of CPython 3.10's `set` (Objects/setobject.c) on (item id, hash) pairs; shared by coral_host.cpp
// (coral_pyset_* / coral_reach_*) and coral_search.cpp (the interval search).  See the comment block in coral_host.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace coral_detail {
struct PySetEmu {
    std::vector<int32_t> key;      // item id, -1 = empty slot
    std::vector<int64_t> hash;
    size_t mask = 7, fill = 0, used = 0;
    PySetEmu() : key(8, -1), hash(8, 0) {}

    static void insert_clean(std::vector<int32_t> &k, std::vector<int64_t> &h, size_t mask, int32_t item, int64_t hv) {
        size_t perturb = (size_t)hv;
        size_t i = (size_t)hv & mask;
        for (;;) {
            size_t e = i;
            if (k[e] < 0) { k[e] = item; h[e] = hv; return; }
            if (i + 9 <= mask) {
                for (int j = 0; j < 9; ++j) {
                    ++e;
                    if (k[e] < 0) { k[e] = item; h[e] = hv; return; }
                }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(size_t minused) {
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<int32_t> nk(newsize, -1);
        std::vector<int64_t> nh(newsize, 0);
        const size_t nm = newsize - 1;
        for (size_t e = 0; e <= mask; ++e)
            if (key[e] >= 0) insert_clean(nk, nh, nm, key[e], hash[e]);
        key.swap(nk);
        hash.swap(nh);
        mask = nm;
        fill = used;
    }
    void add(int32_t item, int64_t hv) {
        size_t perturb = (size_t)hv;
        size_t i = (size_t)hv & mask;
        for (;;) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (key[e] < 0) {                      // unused slot (there are never dummies: nothing is deleted)
                    key[e] = item;
                    hash[e] = hv;
                    ++fill;
                    ++used;
                    if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (hash[e] == hv && key[e] == item) return;       // already present
                ++e;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void merge(const PySetEmu &o) {                    // self |= o
        if (&o == this || o.used == 0) return;
        if ((fill + o.used) * 5 >= mask * 3) resize((used + o.used) * 2);
        if (fill == 0 && mask == o.mask && o.fill == o.used) {
            key = o.key;
            hash = o.hash;
            fill = o.fill;
            used = o.used;
            return;
        }
        if (fill == 0) {
            fill = o.used;
            used = o.used;
            for (size_t e = 0; e <= o.mask; ++e)
                if (o.key[e] >= 0) insert_clean(key, hash, mask, o.key[e], o.hash[e]);
            return;
        }
        for (size_t e = 0; e <= o.mask; ++e)
            if (o.key[e] >= 0) add(o.key[e], o.hash[e]);
    }
};
}  // namespace coral_detail
