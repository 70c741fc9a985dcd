// coral_names.h — read names as ONE byte blob + offsets, and the hash joins over them.
//
// The reference keeps read names as Python str keys of dicts (`chimeric_alignments[r.query_name]`,
// /root/reference/src/infer_breakpoint_graph.py:141-151); all the graph build needs of a name is its identity (name id =
// order of first appearance in the file, the dicts' insertion order) until something prints or iterates a name.  So the
// decoders and the multi-GPU merge keep names as bytes: `NameIndex` interns names of one decode in first-seen order,
// `unify` joins the name tables of several consecutive byte ranges of one file (one per rank) into the numbering a
// single-process decode gives.  No std::string per name, no node-based map.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace coral_names {

// 64-bit hash of a byte string (8 bytes per step, multiply-xorshift mixing; not CPython's hash — that one is only needed for
// the set-order replay and comes from the interpreter itself, coral_pyobjects.cpp:hash_names)
inline uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return x;
}
inline uint64_t hash_bytes(const char *s, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n * 0xFF51AFD7ED558CCDull;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, s, 8);
        h = mix64(h ^ w) + 0x165667B19E3779F9ull;
        s += 8;
        n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, s, n);
        h = mix64(h ^ w ^ ((uint64_t)n << 56));
    }
    return mix64(h);
}

// Names of one decode, interned in first-seen order: id = position in `off`.
struct NameIndex {
    std::vector<char> blob;          // names back to back, no terminators
    std::vector<int64_t> off{0};     // [n + 1]
    std::vector<uint64_t> slot;      // open addressing: (hash & ~mask_id) | (id + 1), 0 = empty;  id in the low 32 bits
    size_t n_slots = 0;

    size_t size() const { return off.size() - 1; }
    void reserve_names(size_t n, size_t bytes) {
        blob.reserve(bytes);
        off.reserve(n + 1);
        grow(n * 2 + 16);
    }
    void grow(size_t want) {
        size_t cap = 1024;
        while (cap < want) cap <<= 1;
        if (cap <= n_slots) return;
        std::vector<uint64_t> old;
        old.swap(slot);
        slot.assign(cap, 0);
        n_slots = cap;
        for (uint64_t v : old)
            if (v) {
                // re-derive the position from the stored high hash bits is not possible: re-hash the name
                const uint32_t id = (uint32_t)(v & 0xFFFFFFFFu) - 1;
                const uint64_t h = hash_bytes(blob.data() + off[id], (size_t)(off[id + 1] - off[id]));
                size_t at = (size_t)(h >> 32) & (n_slots - 1);
                while (slot[at]) at = (at + 1) & (n_slots - 1);
                slot[at] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)(id + 1);
            }
    }
    // id of the name (len bytes at s), adding it when new
    int32_t intern(const char *s, size_t len) {
        if ((size() + 1) * 2 > n_slots) grow(n_slots ? n_slots * 2 : 1024);
        const uint64_t h = hash_bytes(s, len);
        const uint64_t tag = h & 0xFFFFFFFF00000000ull;
        size_t at = (size_t)(h >> 32) & (n_slots - 1);
        for (;;) {
            const uint64_t v = slot[at];
            if (!v) break;
            if ((v & 0xFFFFFFFF00000000ull) == tag) {
                const uint32_t id = (uint32_t)(v & 0xFFFFFFFFu) - 1;
                if ((size_t)(off[id + 1] - off[id]) == len && memcmp(blob.data() + off[id], s, len) == 0) return (int32_t)id;
            }
            at = (at + 1) & (n_slots - 1);
        }
        const uint32_t id = (uint32_t)size();
        slot[at] = tag | (uint64_t)(id + 1);
        blob.insert(blob.end(), s, s + len);
        off.push_back((int64_t)blob.size());
        return (int32_t)id;
    }
};

// Name tables of consecutive pieces of one file (piece p: n[p] names, local ids in first-seen order within the piece) ->
// global ids in first-seen order over the whole file.  lut[p][local id] = global id; out_blob / out_off = the global table.
// Exact (bytes are compared, the hash only routes).  Parallel: names are partitioned by hash, every thread joins one
// partition's names (in file order), then one linear pass numbers the first occurrences in (piece, local id) order.
inline int64_t unify(int32_t n_pieces, const int64_t *n, const uint8_t *const *blob, const int64_t *const *off, int32_t *const *lut,
                     uint8_t *out_blob, int64_t *out_off, int n_threads) {
    std::vector<int64_t> base((size_t)n_pieces + 1, 0);
    for (int32_t p = 0; p < n_pieces; ++p) base[(size_t)p + 1] = base[(size_t)p] + n[p];
    const int64_t total = base[(size_t)n_pieces];
    if (total == 0) {
        out_off[0] = 0;
        return 0;
    }
    if (total > 0x7FFFFFFF) return -1;
    const bool trace = getenv("CORAL_NAMES_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "coral_names::unify %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    n_threads = std::max(1, std::min(n_threads, 64));
    int parts = 1;
    while (parts < n_threads) parts <<= 1;              // partitions = power of two >= threads (top hash bits)
    if (total < 50000) parts = 1, n_threads = 1;
    std::vector<uint64_t> hv((size_t)total);
    // first[g] = flat index (piece base + local id) of the first occurrence of the name at flat index g  (total < 2^31: ids are int32)
    std::vector<int32_t> first((size_t)total);
    auto name_ptr = [&](int64_t g, size_t &len) -> const char * {
        const int32_t p = (int32_t)(std::upper_bound(base.begin(), base.end(), g) - base.begin()) - 1;
        const int64_t l = g - base[(size_t)p];
        len = (size_t)(off[p][l + 1] - off[p][l]);
        return (const char *)blob[p] + off[p][l];
    };
    lap("alloc");
    {   // hashes, in parallel over flat ranges
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t]() {
                const int64_t a = total * t / n_threads, b = total * (t + 1) / n_threads;
                int32_t p = (int32_t)(std::upper_bound(base.begin(), base.end(), a) - base.begin()) - 1;
                for (int64_t g = a; g < b; ++g) {
                    while (g >= base[(size_t)p + 1]) ++p;
                    const int64_t l = g - base[(size_t)p];
                    hv[(size_t)g] = hash_bytes((const char *)blob[p] + off[p][l], (size_t)(off[p][l + 1] - off[p][l]));
                }
            });
        for (auto &x : th) x.join();
    }
    lap("hash");
    {   // join, one partition per task
        std::atomic<int> next{0};
        const int shift = parts > 1 ? 64 - __builtin_ctz((unsigned)parts) : 0;
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&]() {
                std::vector<int32_t> mine;
                std::vector<int32_t> table;
                for (;;) {
                    const int part = next.fetch_add(1);
                    if (part >= parts) break;
                    mine.clear();
                    for (int64_t g = 0; g < total; ++g)
                        if (parts == 1 || (int)(hv[(size_t)g] >> shift) == part) mine.push_back((int32_t)g);
                    size_t cap = 1024;
                    while (cap < mine.size() * 2 + 2) cap <<= 1;
                    table.assign(cap, -1);
                    for (int32_t g : mine) {
                        const uint64_t h = hv[(size_t)g];
                        size_t at = (size_t)(h * 0x9E3779B97F4A7C15ull >> 20) & (cap - 1);
                        size_t len = 0;
                        const char *s = nullptr;
                        for (;;) {
                            const int32_t o = table[at];
                            if (o < 0) {
                                table[at] = g;
                                first[(size_t)g] = g;
                                break;
                            }
                            if (hv[(size_t)o] == h) {
                                if (!s) s = name_ptr(g, len);
                                size_t olen;
                                const char *os = name_ptr(o, olen);
                                if (olen == len && memcmp(os, s, len) == 0) {
                                    first[(size_t)g] = o;
                                    break;
                                }
                            }
                            at = (at + 1) & (cap - 1);
                        }
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    lap("join");
    // number the first occurrences in file order and emit the global table; then the repeats
    int64_t n_global = 0, w = 0;
    out_off[0] = 0;
    for (int32_t p = 0; p < n_pieces; ++p)
        for (int64_t l = 0; l < n[p]; ++l) {
            const int64_t g = base[(size_t)p] + l;
            if (first[(size_t)g] == g) {
                const int64_t len = off[p][l + 1] - off[p][l];
                memcpy(out_blob + w, blob[p] + off[p][l], (size_t)len);
                w += len;
                lut[p][l] = (int32_t)n_global;
                out_off[++n_global] = w;
            }
        }
    for (int32_t p = 0; p < n_pieces; ++p)
        for (int64_t l = 0; l < n[p]; ++l) {
            const int64_t g = base[(size_t)p] + l, f = first[(size_t)g];
            if (f != g) {
                const int32_t fp = (int32_t)(std::upper_bound(base.begin(), base.end(), f) - base.begin()) - 1;
                lut[p][l] = lut[fp][f - base[(size_t)fp]];
            }
        }
    lap("number");
    return n_global;
}

}  // namespace coral_names
