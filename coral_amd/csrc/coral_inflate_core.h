// coral_inflate_core.h — RFC 1951 (DEFLATE) decoder core of the GPU BGZF path (coral_bamgpu.hip).
//
// One 64-lane wave inflates one BGZF block (<= 64 KiB of output, one gzip member).  Symbol decoding is serial by nature and
// runs as WAVE-UNIFORM code (every lane computes the same values; what feeds a branch is made scalar with readfirstlane, the
// rest stays on the vector pipe), the lanes do what is parallel: building the Huffman lookup tables (one table entry per lane
// and step), holding the next 256 input bytes in a register (the bit buffer is refilled with v_readlane, never from memory),
// and copying LZ77 matches 64 bytes per step through a ring of recent output in LDS.
//
// The core is written against a small backend interface (`Wave`) so that the very same decode logic also compiles for the
// host, where tests/test_inflate_core.py drives it against zlib on random, text-like, stored and fixed-Huffman streams (the
// host backend runs "lanes" as loops).  The product only ever uses the device backend; nothing here is a CPU fallback.
//
// Replaces (with coral_bamgpu.hip) what the reference gets from pysam / htslib's bgzf_read + inflate when it opens
// the BAM (/root/reference/src/infer_breakpoint_graph.py:65, :140).
#pragma once
#include <stdint.h>

#define CORAL_LIKELY(x) __builtin_expect(!!(x), 1)
#define CORAL_UNLIKELY(x) __builtin_expect(!!(x), 0)
#if defined(__HIPCC__)
#define CORAL_HD __host__ __device__ __forceinline__
#define CORAL_NOUNROLL _Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
#else
#define CORAL_HD inline
#define CORAL_NOUNROLL
#endif

namespace coral_inflate {

enum { LL_BITS = 10, D_BITS = 8, WAVE_LANES = 64 };
enum { OK = 0, ERR_BTYPE = 1, ERR_STORED = 2, ERR_CODELENS = 3, ERR_OVERSUBSCRIBED = 4, ERR_BAD_CODE = 5, ERR_DISTANCE = 6,
       ERR_OVERFLOW = 7, ERR_INPUT = 8, ERR_SHORT = 9 };

// Literal / length table entry (16 bits):  bits 0..3 code length (0 = not in the primary table: a longer code, an unused
// pattern, or the end-of-block code, which is left to the canonical decode: once per block), bit 4 = "not a literal",
// bits 5..7 extra bits of a length code (7 = end of block, only out of ll_entry), bits 8..15 the literal byte or
// (length base - 3).  A literal is recognised by ONE bit test; a long / unused pattern is 0x0010.
// Distance table entry (32 bits):  bits 0..3 code length (0 as above), bits 8..11 extra bits, bits 16..31 distance base.
enum { LL_NOT_LITERAL = 16, LL_EOB_EXTRA = 7, LL_LONG = 0x0010 };
CORAL_HD uint32_t make_entry(uint32_t nbits, uint32_t kind, uint32_t extra, uint32_t base) {
    return nbits | (kind << 4) | (extra << 8) | (base << 16);
}

struct Tables {                      // per wave, in LDS on the device
    uint16_t ll[1 << LL_BITS];       // literal / length codes of up to LL_BITS bits, indexed by the next LL_BITS stream bits
    union {
        uint32_t dt[1 << D_BITS];    // distance codes of up to D_BITS bits
        uint8_t lens[320 + 4];       // code lengths while a dynamic header is read: dead once the distance code's counts and
    };                               // sorted symbols exist, which is before `dt` is filled (Inflater::build) — same memory
    uint32_t ll_count[16], d_count[16];      // codes per length (canonical decode of the long codes; table build)
    uint16_t ll_sym[288 + 32];       // symbols sorted by (length, symbol); [288..320) = the same for the distance code
};
static_assert(sizeof(Tables) == 2048 + 1024 + 128 + 640, "Tables: 3840 bytes of LDS per wave");

// length symbol 257 + i -> (base, extra bits);  distance symbol d -> (base, extra bits)      (RFC 1951 §3.2.5, computed)
CORAL_HD uint32_t ll_entry(uint32_t sym, uint32_t nbits) {
    if (sym < 256) return nbits | (sym << 8);
    if (sym == 256) return nbits | LL_NOT_LITERAL | (LL_EOB_EXTRA << 5);
    const uint32_t i = sym - 257;
    if (i > 28) return LL_LONG;                             // 286, 287: not valid in a stream (decoded as a bad code)
    if (i < 8) return nbits | LL_NOT_LITERAL | (i << 8);
    if (i == 28) return nbits | LL_NOT_LITERAL | (255u << 8);
    const uint32_t e = (i >> 2) - 1;
    return nbits | LL_NOT_LITERAL | (e << 5) | (((4 + (i & 3)) << e) << 8);
}
CORAL_HD uint32_t dist_entry(uint32_t d, uint32_t nbits) {
    if (d > 29) return 0;
    if (d < 4) return make_entry(nbits, 0, 0, 1 + d);
    const uint32_t e = (d >> 1) - 1;
    return make_entry(nbits, 0, e, 1 + ((2 + (d & 1)) << e));
}

// Backend interface (`W`; W::vector_loop selects Inflater::codes_vector, which needs a few more operations, listed there):
//   per-lane execution is spelled W_FOR_LANES(w, lane) below
//   uint32_t uni(uint32_t)            make a value read from table memory wave-uniform (device: readfirstlane)
//   uint32_t next_dword()             the next 32 input bits
//   void     put_literal(uint32_t b)
//   bool     copy_match(int len, int dist)     false: distance reaches in front of the output
//   bool     copy_stored(long long dwords, uint32_t n)   n raw bytes, starting `dwords` dwords into the input window, to the output
//   uint32_t reset_input_after_stored(long long dwords, uint32_t n)   restart the window behind those bytes; returns the bits
//                                     to drop from the first dword of the new window (device windows are dword-aligned)
//   bool     input_exhausted()        more input pulled than the stream holds (corrupt data)
//   int      produced()               output bytes so far;   int capacity()
//   void     add_count(uint32_t *c)   atomic / plain increment of a table counter
//   void     fence()                  order table writes of all lanes before later reads
// On the device W_FOR_LANES runs its body once with the wave's own lane id; on the host it loops over 64 lanes.
#if defined(__HIP_DEVICE_COMPILE__)
#define W_FOR_LANES(w, lane) for (int lane = (w).lane, _once = 1; _once; _once = 0)
#else
#define W_FOR_LANES(w, lane) for (int lane = 0; lane < WAVE_LANES; ++lane)
#endif

template <class W>
struct Inflater {
    W &w;
    Tables *T;
    uint64_t bb = 0;          // bit buffer (LSB first), wave-uniform
    int bc = 0;               // valid bits in it
    long long dwords = 0;     // input dwords pulled into the bit buffer since the last reset_input

    CORAL_HD Inflater(W &w_, Tables *t) : w(w_), T(t) {}

    CORAL_HD void need() {                       // afterwards bc > 32
        if (bc <= 32) {
            bb |= (uint64_t)w.next_dword() << bc;
            bc += 32;
            ++dwords;
        }
    }
    CORAL_HD uint32_t bits(int n) {              // n <= 32 and n <= bc
        const uint32_t v = n >= 32 ? (uint32_t)bb : ((uint32_t)bb & ~(~0u << n));
        bb >>= n;
        bc -= n;
        return v;
    }

    // Canonical Huffman tables from code lengths lens[0 .. n): count[], sorted symbols, and the primary lookup table.
    // Returns OK, or ERR_OVERSUBSCRIBED.  (Incomplete codes are accepted; their unused patterns decode to "bad code".)
    CORAL_HD int build(const uint8_t *lens, int n, uint32_t *count, uint16_t *sym, uint16_t *table16, uint32_t *table32, int prim_bits) {
        const bool is_dist = table32 != nullptr;
        W_FOR_LANES(w, lane) { if (lane < 16) count[lane] = 0; }
        w.fence();
        W_FOR_LANES(w, lane) { for (int s = lane; s < n; s += WAVE_LANES) w.add_count(&count[lens[s]]); }
        w.fence();
        int left = 1;
        CORAL_NOUNROLL
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= (int)w.uni(count[l]);
            if (left < 0) return ERR_OVERSUBSCRIBED;
        }
        // symbols of equal length in increasing symbol order: lane l places the symbols of length l
        W_FOR_LANES(w, lane) {
            if (lane >= 1 && lane <= 15) {
                uint32_t at = 0;
                CORAL_NOUNROLL
                for (int l = 1; l < lane; ++l) at += count[l];
                CORAL_NOUNROLL
                for (int s = 0; s < n; ++s)
                    if (lens[s] == lane) sym[at++] = (uint16_t)s;
            }
        }
        w.fence();
        // every table index is decoded canonically, bit by bit (entries are independent: one per lane and step)
        const int size = 1 << prim_bits;
        W_FOR_LANES(w, lane) {
            for (int i = lane; i < size; i += WAVE_LANES) {
                uint32_t code = 0, first = 0, index = 0, entry = is_dist ? 0u : (uint32_t)LL_LONG;
                CORAL_NOUNROLL
                for (int l = 1; l <= prim_bits; ++l) {
                    code |= ((uint32_t)i >> (l - 1)) & 1u;
                    const uint32_t cnt = count[l];
                    if (code - first < cnt) {                     // (unsigned: code >= first always holds here)
                        const uint32_t s = sym[index + (code - first)];
                        entry = is_dist ? dist_entry(s, (uint32_t)l) : (s == 256 ? (uint32_t)LL_LONG : ll_entry(s, (uint32_t)l));
                        break;
                    }
                    index += cnt;
                    first = (first + cnt) << 1;
                    code <<= 1;
                }
                if (is_dist) table32[i] = entry;
                else table16[i] = (uint16_t)entry;
            }
        }
        w.fence();
        return OK;
    }

    // A code longer than the primary table (or an unused pattern): canonical decode from the first bit.  Returns the symbol
    // or -1.  Needs up to 15 bits in the buffer (callers have > 32).
    CORAL_HD int decode_long(const uint32_t *count, const uint16_t *sym) {
        uint32_t code = 0, first = 0, index = 0;
        CORAL_NOUNROLL
        for (int l = 1; l <= 15; ++l) {
            code |= (uint32_t)(bb >> (l - 1)) & 1u;
            const uint32_t cnt = w.uni(count[l]);
            if (code - first < cnt) {
                bb >>= l;
                bc -= l;
                return (int)w.uni(sym[index + (code - first)]);
            }
            index += cnt;
            first = (first + cnt) << 1;
            code <<= 1;
        }
        return -1;
    }

    CORAL_HD int read_dynamic_header(int *n_ll, int *n_dist) {
        need();
        const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
        if (hlit > 286 || hdist > 30) return ERR_CODELENS;
        uint8_t *lens = T->lens;
        // code-length code: 19 symbols, 3 bits each, in the permuted order of RFC 1951 §3.2.7
        W_FOR_LANES(w, lane) { if (lane < 19) lens[lane] = 0; }
        w.fence();
        for (int i = 0; i < hclen; ++i) {
            need();
            const uint32_t v = bits(3);
            int pos;                                       // 16 17 18 0 | 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            if (i < 3) pos = 16 + i;
            else if (i == 3) pos = 0;
            else {
                const int k = i - 4;
                pos = (k & 1) ? 7 - (k >> 1) : 8 + (k >> 1);
            }
            W_FOR_LANES(w, lane) { if (lane == 0) lens[pos] = (uint8_t)v; }
        }
        w.fence();
        // the code-length code is decoded canonically (7-bit codes, at most 316 symbols): count / sym reuse the distance slots
        uint32_t *cl_count = T->d_count;
        uint16_t *cl_sym = T->ll_sym + 288;
        {
            W_FOR_LANES(w, lane) { if (lane < 16) cl_count[lane] = 0; }
            w.fence();
            W_FOR_LANES(w, lane) { if (lane < 19) w.add_count(&cl_count[lens[lane]]); }
            w.fence();
            int left = 1;
            for (int l = 1; l <= 7; ++l) {
                left <<= 1;
                left -= (int)w.uni(cl_count[l]);
                if (left < 0) return ERR_OVERSUBSCRIBED;
            }
            W_FOR_LANES(w, lane) {
                if (lane >= 1 && lane <= 7) {
                    uint32_t at = 0;
                    for (int l = 1; l < lane; ++l) at += cl_count[l];
                    for (int s = 0; s < 19; ++s)
                        if (lens[s] == lane) cl_sym[at++] = (uint16_t)s;
                }
            }
            w.fence();
        }
        // the hlit + hdist code lengths (they may run across the boundary between the two alphabets)
        const int total = hlit + hdist;
        // lens[] is reused as the output: the 19 code-length lengths are no longer needed once count / sym are built
        int i = 0;
        uint32_t prev = 0;
        while (i < total) {
            need();
            const int s = decode_long(cl_count, cl_sym);
            if (s < 0) return ERR_CODELENS;
            if (s < 16) {
                W_FOR_LANES(w, lane) { if (lane == 0) lens[i] = (uint8_t)s; }
                prev = (uint32_t)s;
                ++i;
            } else {
                uint32_t rep, val;
                if (s == 16) {
                    if (i == 0) return ERR_CODELENS;
                    rep = 3 + bits(2);
                    val = prev;
                } else if (s == 17) {
                    rep = 3 + bits(3);
                    val = 0;
                } else {
                    rep = 11 + bits(7);
                    val = 0;
                }
                if (i + (int)rep > total) return ERR_CODELENS;
                W_FOR_LANES(w, lane) { for (int k = lane; k < (int)rep; k += WAVE_LANES) lens[i + k] = (uint8_t)val; }
                prev = val;
                i += (int)rep;
            }
        }
        w.fence();
        if (w.uni(lens[256]) == 0) return ERR_CODELENS;                 // no end-of-block code
        *n_ll = hlit;
        *n_dist = hdist;
        return OK;
    }

    CORAL_HD int fixed_tables() {
        uint8_t *lens = T->lens;
        W_FOR_LANES(w, lane) {
            for (int s = lane; s < 320; s += WAVE_LANES)
                lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
        }
        w.fence();
        return OK;
    }

    // Stored block: LEN / NLEN after the next byte boundary, then LEN raw bytes.
    CORAL_HD int stored() {
        bits(bc & 7);                                   // to the byte boundary (bc stays a multiple of 8)
        need();
        const uint32_t len = bits(16), nlen = bits(16);
        if ((len ^ 0xffffu) != nlen) return ERR_STORED;
        if (w.produced() + (int)len > w.capacity()) return ERR_OVERFLOW;
        // bytes still in the bit buffer belong to the block; the rest is copied straight from the input
        uint32_t left = len;
        while (left && bc >= 8) {
            w.put_literal(bits(8));
            --left;
        }
        if (left) {
            // bit buffer is empty here: the input position is exactly `dwords` dwords into the window
            if (!w.copy_stored(dwords, left)) return ERR_INPUT;
            const uint32_t skip = w.reset_input_after_stored(dwords, left);      // bits to drop in the first dword of the new window
            bb = 0;
            bc = 0;
            dwords = 0;
            need();
            bits((int)skip);
        }
        return OK;
    }

    // The symbol loop the device uses (W::vector_loop).  A CU issues one scalar and (over its four SIMDs) one vector instruction
    // per cycle, and symbol decoding is scalar by nature — so everything that does not feed a branch runs on the vector pipe,
    // redundantly in all lanes: table index, field extraction (code length, extra bits, base), ring addresses.  Scalar work per
    // symbol is the bit-buffer shift, the position and the branch conditions.  Two table symbols are decoded per refill check;
    // nothing in the hot path tests for errors: the backend clamps bad lengths / distances, collects them (`bad()`), bounds the
    // output (`over`, looked at once per round), and the canonical loop handles codes longer than the table, unused patterns
    // and the end of the block.  Values named *v live in vector registers on the device (plain integers on the host).
    // Needs from the backend: vec(x), bfe(x, offset, width), lit(bytev), match(lenv, distv), needs_attention() / attention()
    // (one compare per round: output to be written back, or the loop is to stop), fail(code) / failed() / error_code(), and
    // `static constexpr bool has_fast` (+ fast(bb, bc, dwords, tables, lenv, distv) when true, see the loop).
    CORAL_HD int codes_vector() {
        const uint16_t *ll = T->ll;
        const uint32_t *dt = T->dt;
        // ONE way out besides the end-of-block code: every rare problem raises the backend's flag (fail), and the round's
        // single check sees it — several `return`s inside the loop cost a scalar state machine in every round
        for (;;) {
            // A backend with a hand-written fast path (W::has_fast: the device, DevWaveT::fast) decodes as many symbols as it
            // can there — table hits, short near matches, input still in the window — and comes back where it cannot go on:
            //   0  at a symbol boundary (attention due, a code outside the tables, the input window to be switched, ...)
            //   1  a match length has been read (lenv), its distance code is next
            //   2  length and distance have been read (lenv, distv), the copy is not done
            // and this loop does that ONE step the general way.  Without a fast path every round starts at 0.
            uint32_t lenv = 0, distv = 0;
            int at = 0;
            if constexpr (W::has_fast) at = w.fast(bb, bc, dwords, T, lenv, distv);
            if (at == 0) {
                if (CORAL_UNLIKELY(w.needs_attention()) && !w.attention()) break;      // a line of output is complete, or stop
                need();                                                   // more than 32 bits
                uint32_t bbv = w.vec((uint32_t)bb);
                uint32_t ev = ll[bbv & ((1u << LL_BITS) - 1u)];
                uint32_t e = w.uni(ev);
                if (!(e & LL_NOT_LITERAL)) {
                    const uint32_t nb1 = e & 15u;
                    bb >>= nb1;
                    bc -= (int)nb1;
                    w.lit(ev >> 8);
                    if constexpr (W::has_fast) continue;                  // (the fast path takes the literals that follow)
                    bbv = w.vec((uint32_t)bb);                            // at least 23 bits left: enough for any table entry
                    ev = ll[bbv & ((1u << LL_BITS) - 1u)];
                    e = w.uni(ev);
                    if (!(e & LL_NOT_LITERAL)) {
                        const uint32_t nb2 = e & 15u;
                        bb >>= nb2;
                        bc -= (int)nb2;
                        w.lit(ev >> 8);
                        continue;
                    }
                }
                if (CORAL_UNLIKELY((e & 15u) == 0)) {                     // not in the table: canonical decode (up to 15 of >= 23 bits)
                    const int s = decode_long(T->ll_count, T->ll_sym);
                    if (s == 256) return w.failed() ? w.error_code() : OK;
                    const uint32_t x = s < 0 ? (uint32_t)LL_LONG : ll_entry((uint32_t)s, 1);     // (a length of 1 tells a length-3 code from LL_LONG)
                    if (s >= 0 && s < 256) {
                        w.lit((uint32_t)s);
                        continue;
                    }
                    if (x == LL_LONG) {
                        w.fail(ERR_BAD_CODE);
                        continue;
                    }
                    lenv = 3u + (x >> 8) + bits((int)((x >> 5) & 7u));
                } else {                                                  // a length: code + extra bits leave the buffer in one shift
                    const uint32_t nbv = ev & 15u, xbv = (ev >> 5) & 7u;
                    lenv = 3u + (ev >> 8) + w.bfe(bbv, nbv, xbv);
                    const uint32_t tot = w.uni(nbv + xbv);                // <= 10 + 5
                    bb >>= tot;
                    bc -= (int)tot;
                }
            }
            if (at <= 1) {
                need();
                const uint32_t bbv = w.vec((uint32_t)bb);
                const uint32_t dv = dt[bbv & ((1u << D_BITS) - 1u)];
                const uint32_t d = w.uni(dv);
                if (CORAL_UNLIKELY((d & 15u) == 0)) {
                    const int s = decode_long(T->d_count, T->ll_sym + 288);
                    const uint32_t x = s < 0 ? 0u : dist_entry((uint32_t)s, 1);
                    if (x == 0) {
                        w.fail(ERR_BAD_CODE);
                        continue;
                    }
                    distv = (x >> 16) + bits((int)((x >> 8) & 15u));
                } else {
                    const uint32_t nbv = dv & 15u, xbv = (dv >> 8) & 15u;
                    distv = (dv >> 16) + w.bfe(bbv, nbv, xbv);            // <= 15 + 13 of the 32 low bits
                    const uint32_t tot = w.uni(nbv + xbv);
                    bb >>= tot;
                    bc -= (int)tot;
                }
            }
            w.match(lenv, distv);
        }
        return w.error_code();
    }

    // The plain symbol loop (one symbol per round, capacity tested per literal).
    CORAL_HD int codes() {
        if constexpr (W::vector_loop) return codes_vector();
        const uint16_t *ll = T->ll;
        const uint32_t *dt = T->dt;
        for (;;) {
            need();
            uint32_t e = w.uni(ll[(uint32_t)bb & ((1u << LL_BITS) - 1u)]);
            uint32_t nb = e & 15u;
            if (nb == 0) {                                           // long code
                const int s = decode_long(T->ll_count, T->ll_sym);
                if (s < 0) return ERR_BAD_CODE;
                e = ll_entry((uint32_t)s, 1);
                if (e == LL_LONG) return ERR_BAD_CODE;
            } else {
                bb >>= nb;
                bc -= (int)nb;
            }
            if (!(e & LL_NOT_LITERAL)) {
                if (w.produced() >= w.capacity()) return ERR_OVERFLOW;
                w.put_literal(e >> 8);
                continue;
            }
            const uint32_t xb = (e >> 5) & 7u;
            if (xb == LL_EOB_EXTRA) return OK;
            const int len = (int)(3u + (e >> 8) + bits((int)xb));
            need();
            uint32_t d = w.uni(dt[(uint32_t)bb & ((1u << D_BITS) - 1u)]);
            nb = d & 15u;
            if (nb == 0) {
                const int s = decode_long(T->d_count, T->ll_sym + 288);
                if (s < 0) return ERR_BAD_CODE;
                d = dist_entry((uint32_t)s, 1);
                if (d == 0) return ERR_BAD_CODE;
            } else {
                bb >>= nb;
                bc -= (int)nb;
            }
            const uint32_t dxb = (d >> 8) & 15u;
            const int dist = (int)((d >> 16) + bits((int)dxb));
            if (w.produced() + len > w.capacity()) return ERR_OVERFLOW;
            if (!w.copy_match(len, dist)) return ERR_DISTANCE;
            if (w.input_exhausted()) return ERR_INPUT;
        }
    }

    // One raw DEFLATE stream (all its blocks).  `skip_bits`: bits of the window's first dword in front of the stream (device
    // windows start on a dword boundary).  Returns OK when the final block ended and `capacity` bytes were produced.
    CORAL_HD int run(int skip_bits = 0) {
        if (skip_bits) {
            need();
            bits(skip_bits);
        }
        for (;;) {
            need();
            const uint32_t last = bits(1), type = bits(2);
            int rc;
            if (type == 0) {
                rc = stored();
            } else if (type == 3) {
                rc = ERR_BTYPE;
            } else {                                        // one build site and one decode loop for both Huffman block types
                int n_ll = 288, n_dist = 30;
                rc = type == 1 ? fixed_tables() : read_dynamic_header(&n_ll, &n_dist);
                if (rc == OK) rc = build(T->lens, n_ll, T->ll_count, T->ll_sym, T->ll, nullptr, LL_BITS);
                if (rc == OK) rc = build(T->lens + n_ll, n_dist, T->d_count, T->ll_sym + 288, nullptr, T->dt, D_BITS);
                if (rc == OK) rc = codes();
            }
            if (rc != OK) return rc;
            if (w.input_exhausted()) return ERR_INPUT;
            if (last) break;
        }
        return w.produced() == w.capacity() ? OK : ERR_SHORT;
    }
};

}  // namespace coral_inflate
