// coral_bam.cpp — host-side BAM (BGZF) codec of libcoral_hip.so: file -> structure-of-arrays records (and back, for tests
// and benchmarks).
//
// Replaces what the reference gets from pysam.AlignmentFile(path, 'rb') + the whole-file fetch()
// (/root/reference/src/infer_breakpoint_graph.py:65, :140-158): every record is decoded ONCE into the SoA layout of
// include/coral_hip.h (CIGAR padded to 16 bytes with op 15), the SA tag is tokenised into numeric rows, NM is extracted,
// and aligned non-ACGT bases are listed (pysam count_coverage counts only A/C/G/T).
//
// Decoder pipeline (one pass, bounded memory — SEQ / QUAL bytes are dropped as soon as their chunk is parsed):
//   file (mmap) -> BGZF block table of the byte range -> chunks of consecutive blocks
//     stage 1 (worker pool)   inflate a chunk (zlib raw inflate, one z_stream per thread, inflateReset per block)
//     stage 2 (caller, cheap) hop along the record lengths: record starts of the chunk, hand-over of the record that
//                             straddles into the next chunk
//     stage 3 (worker pool)   parse the chunk's records into a per-chunk partial (CIGAR copy + padding, SA tokens, NM,
//                             non-ACGT scan)
//     stage 4 (caller)        append the partials in file order; read names -> ids
// A byte range [rank, world) of the file can be decoded on its own (one process per GPU): the range
// starts at the first BGZF block at or after its first byte (blocks are found by their magic + BC subfield and a chained
// check), its first record is the first offset at or after that block's first uncompressed byte from which a chain of
// plausible records starts, and it ends with the record that straddles into the next range — the same rule seen from
// both sides, so consecutive ranges neither drop nor repeat a record.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/coral_hip.h"

#include "coral_bam_common.h"

using namespace coral_bam;

namespace {

// ---------------------------------------------------------------------------------------------
// the decoder
// ---------------------------------------------------------------------------------------------
const size_t CHUNK_BLOCKS = 48;               // ~3 MiB of inflated bytes per chunk
const size_t HEADROOM = 1 << 20;              // room in front of a chunk's bytes for the head of a straddling record
const size_t OVERHANG_BLOCKS = 1024;          // blocks after the byte range that its last record may straddle into (64 MiB)

struct Chunk {
    size_t b0 = 0, b1 = 0;                    // blocks [b0, b1)
    std::vector<uint8_t> buf;                 // [HEADROOM | inflated blocks]
    size_t own = 0;                           // inflated bytes
    size_t begin = HEADROOM;                  // first byte that matters (HEADROOM - carried bytes)
    std::vector<size_t> starts;               // record starts (offsets into buf)
    Partial part;
    Flag inflated, parsed;
    std::atomic<int> bad{0};
    bool submitted_parse = false;
};

bool decode_file(const char *path, int n_threads, int rank, int world, Decoded &D) {
    const auto t_start = std::chrono::steady_clock::now();
    MappedFile f;
    if (!f.open(path, D.error)) return false;
    n_threads = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    if (world < 1 || rank < 0 || rank >= world) { D.error = "bad rank / world"; return false; }
    // ---- header (every rank): inflate from block 0 until the reference list is complete
    size_t hdr_bytes = 0;                      // header length in the uncompressed stream
    RefIds ref_id;
    if (!read_bam_header(f, D, ref_id, &hdr_bytes)) return false;
    const int32_t n_ref = (int32_t)D.ref_names.size();
    // ---- block table: the blocks that START inside this rank's byte range, plus an overhang for the last record
    const uint64_t byte_lo = rank == 0 ? 0 : f.size / (uint64_t)world * (uint64_t)rank;
    const uint64_t byte_hi = rank == world - 1 ? f.size : f.size / (uint64_t)world * (uint64_t)(rank + 1);
    uint64_t first = 0;
    if (rank > 0 && !find_block(f, byte_lo, &first)) first = f.size;
    std::vector<Block> blocks;
    size_t n_own = 0;
    uint64_t own_bytes = 0;                    // uncompressed offset (from this rank's first block) of the next rank's first block
    for (uint64_t at = first; at < f.size && blocks.size() < n_own + OVERHANG_BLOCKS;) {
        Block b;
        if (!bgzf_header(f.data + at, f.size - at, b)) { D.error = "not a BGZF block"; return false; }
        b.off = at;
        if (at < byte_hi) { ++n_own; own_bytes += b.isize; }
        blocks.push_back(b);
        at += b.csize;
    }
    std::vector<std::unique_ptr<Chunk>> chunks;
    for (size_t b = 0; b < blocks.size(); b += CHUNK_BLOCKS) {
        chunks.emplace_back(new Chunk());
        chunks.back()->b0 = b;
        chunks.back()->b1 = std::min(blocks.size(), b + CHUNK_BLOCKS);
    }
    const size_t n_chunks = chunks.size();
    const size_t own_chunks = (n_own + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;      // chunks holding at least one owned block
    const bool last_rank = rank == world - 1;

    Pool pool(n_threads);
    const size_t window = (size_t)std::max(4, 3 * n_threads);
    auto submit_inflate = [&](size_t k) {
        Chunk *c = chunks[k].get();
        pool.submit([&, c]() {
            thread_local ZStream z;
            size_t total = 0;
            for (size_t b = c->b0; b < c->b1; ++b) total += blocks[b].isize;
            c->buf.resize(HEADROOM + total);
            c->own = total;
            size_t o = HEADROOM;
            for (size_t b = c->b0; b < c->b1; ++b) {
                if (!z.ok || !inflate_block(z.zs, f, blocks[b], c->buf.data() + o)) { c->bad = 1; break; }
                o += blocks[b].isize;
            }
            c->inflated.set();
        });
    };
    auto submit_parse = [&](size_t k) {
        Chunk *c = chunks[k].get();
        c->submitted_parse = true;
        pool.submit([&, c]() {
            Partial &pt = c->part;
            pt.cigar.reserve(c->own / 6 + 64);
            for (size_t s : c->starts) {
                const uint8_t *q = c->buf.data() + s;
                if (!decode_record(q + 4, rd32(q), ref_id, pt, pt.error)) break;
            }
            std::vector<uint8_t>().swap(c->buf);            // SEQ / QUAL bytes are gone from here on
            c->parsed.set();
        });
    };
    auto wait = [&](Flag &fl) { while (!fl.get()) if (!pool.help_one()) std::this_thread::yield(); };

    D.names.grow(1 << 21);
    auto merge = [&](Chunk &c) -> bool {
        Partial &pt = c.part;
        if (!pt.error.empty()) { D.error = pt.error; return false; }
        const int64_t base = (int64_t)D.tid.size();
        auto app = [](std::vector<int32_t> &d, const std::vector<int32_t> &s) { d.insert(d.end(), s.begin(), s.end()); };
        app(D.tid, pt.tid); app(D.pos, pt.pos); app(D.end, pt.end); app(D.flag, pt.flag); app(D.mapq, pt.mapq);
        app(D.qlen, pt.qlen); app(D.has_seq, pt.has_seq); app(D.nm, pt.nm); app(D.n_cigar, pt.n_cigar);
        D.cigar.insert(D.cigar.end(), pt.cigar.begin(), pt.cigar.end());
        for (int64_t l : pt.cigar_len) D.cigar_off.push_back(D.cigar_off.back() + l);
        app(D.sa, pt.sa); app(D.sa_nm, pt.sa_nm);
        for (int32_t cnt : pt.sa_cnt) D.sa_off.push_back(D.sa_off.back() + cnt);
        for (int64_t l : pt.na_rec_local) D.na_rec.push_back(base + l);
        app(D.na_pos, pt.na_pos);
        for (const char *s = pt.names.data(), *e = s + pt.names.size(); s < e;) {
            const size_t len = strlen(s);
            D.name_id.push_back(D.names.intern(s, len));
            s += len + 1;
        }
        c.part = Partial();
        return true;
    };

    // ---- the ordered walk: stage 2 for chunk k, stage 4 for the chunks whose parse is done (in order)
    size_t next_inflate = 0, merged = 0;
    auto top_up = [&](size_t k) {               // beyond the owned chunks only one chunk ahead (the overhang is rarely needed)
        const size_t upto = std::min(n_chunks, std::max(k + 1, std::min(k + window, own_chunks + 1)));
        while (next_inflate < upto) submit_inflate(next_inflate++);
    };
    bool searching = rank > 0;                 // still looking for the first record of the range
    uint64_t pos = rank == 0 ? hdr_bytes : 0;  // offset (in this rank's uncompressed stream) of the next record start
    std::vector<uint8_t> carry;                // bytes [pos, end of the previous chunk): head of a straddling record
    uint64_t ubase = 0;                        // stream offset of the current chunk's first inflated byte
    bool done = false;
    for (size_t k = 0; k < n_chunks && !done; ++k) {
        top_up(k);
        Chunk &c = *chunks[k];
        wait(c.inflated);
        if (c.bad) { D.error = "zlib inflate failed or CRC32 mismatch (corrupt BGZF block)"; return false; }
        if (carry.size() > HEADROOM) {         // (a record of more than 1 MiB straddles) make room
            c.buf.insert(c.buf.begin(), carry.size() - HEADROOM, 0);
            c.begin = 0;
            memcpy(c.buf.data(), carry.data(), carry.size());
        } else {
            c.begin = HEADROOM - carry.size();
            if (!carry.empty()) memcpy(c.buf.data() + c.begin, carry.data(), carry.size());
        }
        const size_t carried = carry.size();
        carry.clear();
        const uint8_t *base = c.buf.data() + c.begin;
        const size_t nbytes = c.buf.size() - c.begin;
        const uint64_t u0 = ubase - carried;     // stream offset of base[0]
        size_t p;
        if (searching) {
            p = nbytes;
            for (size_t cand = 0; cand + 36 <= nbytes; ++cand) {
                size_t q = cand;
                int chain = 0;
                bool ok = true;
                while (chain < 8 && q + 36 <= nbytes) {
                    uint64_t len;
                    if (!plausible_record(base + q, nbytes - q, n_ref, &len)) { ok = false; break; }
                    q += len;
                    ++chain;
                }
                if (ok && chain >= 3) { p = cand; break; }
            }
            if (p == nbytes) {                   // nothing yet (e.g. inside one enormous record): search on with these bytes kept
                if (k + 1 == n_chunks) break;    // no record starts in this range at all
                carry.assign(base, base + nbytes);
                ubase += c.own;
                continue;
            }
            searching = false;
        } else {
            if (pos - u0 >= nbytes) {            // the BAM header is longer than this chunk (rank 0 only)
                ubase += c.own;
                continue;
            }
            p = (size_t)(pos - u0);
        }
        while (p < nbytes) {
            if (!last_rank && u0 + p >= own_bytes) { done = true; break; }     // starts in the next rank's range: theirs
            if (nbytes - p < 4) break;
            const uint64_t len = 4ull + rd32(base + p);
            if (len < 36) { D.error = "record shorter than its fixed fields"; return false; }
            if (p + len > nbytes) break;                                       // straddles into the next chunk
            c.starts.push_back(c.begin + p);
            p += (size_t)len;
        }
        if (!done) {
            if (!last_rank && u0 + p >= own_bytes) done = true;
            else if (p < nbytes) carry.assign(base + p, base + nbytes);
        }
        pos = u0 + p;
        ubase += c.own;
        submit_parse(k);
        while (merged <= k && (chunks[merged]->parsed.get() || !chunks[merged]->submitted_parse || k + 1 - merged > window)) {
            Chunk &m = *chunks[merged];
            if (m.submitted_parse) {
                wait(m.parsed);
                if (!merge(m)) return false;
            }
            ++merged;
        }
    }
    if (searching && rank > 0 && n_own > 0) { /* the whole range lies inside one record of the previous range */ }
    if (!carry.empty() && !done && !searching) { D.error = last_rank ? "truncated record at the end of the file" : "a record straddles further than the supported overhang"; return false; }
    for (; merged < n_chunks; ++merged) {
        Chunk &m = *chunks[merged];
        if (!m.submitted_parse) continue;
        wait(m.parsed);
        if (!merge(m)) return false;
    }
    // inflate tasks that were submitted ahead but never used must finish before the pool (and the chunks) go away
    for (size_t k = 0; k < next_inflate; ++k) wait(chunks[k]->inflated);
    for (size_t b = 0; b < n_own; ++b) { D.compressed_bytes += blocks[b].csize; D.uncompressed_bytes += blocks[b].isize; }
    D.n_blocks = (int64_t)n_own;
    D.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return true;
}

// ---------------------------------------------------------------------------------------------
// writer (tests, benchmarks): SoA records -> coordinate-sorted BAM with deterministic SEQ
// ---------------------------------------------------------------------------------------------
inline void put32(std::vector<uint8_t> &o, uint32_t v) { const size_t n = o.size(); o.resize(n + 4); memcpy(o.data() + n, &v, 4); }
inline void put16(std::vector<uint8_t> &o, uint16_t v) { const size_t n = o.size(); o.resize(n + 2); memcpy(o.data() + n, &v, 2); }

inline uint32_t mix32(uint32_t x) {            // the finaliser of MurmurHash3
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct WriteJob {
    int64_t n_rec;
    const int32_t *tid, *pos, *flag, *mapq, *qlen, *has_seq, *nm, *name_id, *n_cigar;
    const int64_t *cigar_off;
    const uint32_t *cigar;
    const int64_t *sa_off;
    const int32_t *sa, *sa_nm;
    int64_t n_na;
    const int64_t *na_rec;
    const int32_t *na_pos;
    const char *const *names;
    int32_t n_ref;
    const char *const *ref_names;
    uint32_t seed;
};

void sa_text(const WriteJob &J, int64_t row, std::string &out) {
    const int32_t *r = J.sa + 8 * row;
    char buf[160];
    std::string cg;
    if (r[3] > 0) { snprintf(buf, sizeof(buf), "%dS", r[3]); cg += buf; }
    snprintf(buf, sizeof(buf), "%dM", r[4]); cg += buf;
    if (r[5] > 0) { snprintf(buf, sizeof(buf), "%dI", r[5]); cg += buf; }
    if (r[5] < 0) { snprintf(buf, sizeof(buf), "%dD", -r[5]); cg += buf; }
    if (r[6] > 0) { snprintf(buf, sizeof(buf), "%dS", r[6]); cg += buf; }
    snprintf(buf, sizeof(buf), "%s,%d,%c,%s,%d,%d;", (r[0] >= 0 && r[0] < J.n_ref) ? J.ref_names[r[0]] : "*", r[1], r[2] ? '-' : '+',
             cg.c_str(), r[7], J.sa_nm[row]);
    out += buf;
}

void write_record(const WriteJob &J, int64_t i, std::vector<uint8_t> &o, std::vector<uint8_t> &seq) {
    const uint32_t *ops = J.cigar + J.cigar_off[i];
    const uint32_t n_ops = (uint32_t)J.n_cigar[i];
    const uint32_t l_seq = J.has_seq[i] ? (uint32_t)J.qlen[i] : 0;
    int64_t rlen = 0;
    for (uint32_t k = 0; k < n_ops; ++k) rlen += REF_ADV[ops[k] & 15] ? (ops[k] >> 4) : 0;
    if (J.flag[i] & 4) rlen = 0;
    // SEQ: hash-made ACGT, N at the listed aligned positions of this record
    seq.assign(((size_t)l_seq + 1) / 2, 0);
    if (l_seq) {
        static const uint8_t CODE[4] = {1, 2, 4, 8};
        uint32_t state = mix32(J.seed ^ (uint32_t)(i * 2654435761u));
        for (uint32_t q = 0; q < l_seq; q += 16) {
            state = mix32(state + 0x9e3779b9u);
            uint32_t bits = state;
            for (uint32_t j = q; j < q + 16 && j < l_seq; ++j, bits >>= 2)
                seq[j >> 1] |= (uint8_t)(CODE[bits & 3] << ((j & 1) ? 0 : 4));
        }
        const int64_t *lo = std::lower_bound(J.na_rec, J.na_rec + J.n_na, i), *hi = std::upper_bound(J.na_rec, J.na_rec + J.n_na, i);
        for (const int64_t *it = lo; it < hi; ++it) {
            const int64_t want = J.na_pos[it - J.na_rec];
            int64_t q = 0, r = J.pos[i];
            for (uint32_t k = 0; k < n_ops; ++k) {
                const uint32_t op = ops[k] & 15, len = ops[k] >> 4;
                if (IS_ALN[op] && want >= r && want < r + (int64_t)len) {
                    const uint64_t qi = (uint64_t)(q + (want - r));
                    if (qi < l_seq) {
                        seq[qi >> 1] &= (uint8_t)((qi & 1) ? 0xf0 : 0x0f);
                        seq[qi >> 1] |= (uint8_t)(15 << ((qi & 1) ? 0 : 4));
                    }
                    break;
                }
                if (QRY_ADV[op]) q += len;
                if (REF_ADV[op]) r += len;
            }
        }
    }
    // tags
    std::vector<uint8_t> tags;
    tags.push_back('N'); tags.push_back('M'); tags.push_back('i'); put32(tags, (uint32_t)J.nm[i]);
    if (J.sa_off[i + 1] > J.sa_off[i]) {
        std::string s;
        for (int64_t r = J.sa_off[i]; r < J.sa_off[i + 1]; ++r) sa_text(J, r, s);
        tags.push_back('S'); tags.push_back('A'); tags.push_back('Z');
        tags.insert(tags.end(), s.begin(), s.end());
        tags.push_back(0);
    }
    uint32_t cig_field_n = n_ops;
    uint32_t placeholder[2];
    const uint32_t *cig_field = ops;
    if (n_ops > 65535) {                       // long CIGAR -> CG:B,I tag + the <l_seq>S<rlen>N placeholder (SAM spec §4.2.2)
        tags.push_back('C'); tags.push_back('G'); tags.push_back('B'); tags.push_back('I'); put32(tags, n_ops);
        const size_t n = tags.size();
        tags.resize(n + 4ull * n_ops);
        memcpy(tags.data() + n, ops, 4ull * n_ops);
        placeholder[0] = (l_seq << 4) | 4;
        placeholder[1] = ((uint32_t)rlen << 4) | 3;
        cig_field = placeholder;
        cig_field_n = 2;
    }
    const char *name = J.names[J.name_id[i]];
    const uint32_t l_name = (uint32_t)strlen(name) + 1;
    const uint32_t block_size = 32 + l_name + 4 * cig_field_n + (uint32_t)seq.size() + l_seq + (uint32_t)tags.size();
    put32(o, block_size);
    put32(o, (uint32_t)J.tid[i]);
    put32(o, (uint32_t)J.pos[i]);
    o.push_back((uint8_t)l_name);
    o.push_back((uint8_t)J.mapq[i]);
    put16(o, (uint16_t)reg2bin(J.pos[i], J.pos[i] + (rlen > 0 ? rlen : 1)));
    put16(o, (uint16_t)cig_field_n);
    put16(o, (uint16_t)J.flag[i]);
    put32(o, l_seq);
    put32(o, (uint32_t)-1);
    put32(o, (uint32_t)-1);
    put32(o, 0);
    o.insert(o.end(), (const uint8_t *)name, (const uint8_t *)name + l_name);
    { const size_t n = o.size(); o.resize(n + 4ull * cig_field_n); if (cig_field_n) memcpy(o.data() + n, cig_field, 4ull * cig_field_n); }
    o.insert(o.end(), seq.begin(), seq.end());
    o.insert(o.end(), (size_t)l_seq, (uint8_t)0xff);
    o.insert(o.end(), tags.begin(), tags.end());
}

// deflate `n` bytes into BGZF blocks of at most 0xff00 input bytes, appended to `out`
bool bgzf_compress(const uint8_t *in, size_t n, int level, std::vector<uint8_t> &out) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    std::vector<uint8_t> comp(70000);
    for (size_t at = 0; at < n; at += 0xff00) {
        const size_t len = std::min<size_t>(0xff00, n - at);
        deflateReset(&zs);
        zs.next_in = (Bytef *)(in + at);
        zs.avail_in = (uInt)len;
        zs.next_out = comp.data();
        zs.avail_out = (uInt)comp.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return false; }
        const size_t clen = comp.size() - zs.avail_out;
        static const uint8_t HDR[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
        out.insert(out.end(), HDR, HDR + 16);
        put16(out, (uint16_t)(clen + 25));
        out.insert(out.end(), comp.begin(), comp.begin() + (long)clen);
        put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), in + at, (uInt)len));
        put32(out, (uint32_t)len);
    }
    deflateEnd(&zs);
    return true;
}

thread_local std::string g_bam_err;

}  // namespace

void coral_bam::set_error(const std::string &msg) { g_bam_err = msg; }

extern "C" const char *coral_bam_last_error(void) { return g_bam_err.c_str(); }

extern "C" int coral_bam_decode_range(const char *path, int32_t n_threads, int32_t rank, int32_t world, void **handle) {
    if (!path || !handle) return CORAL_ERR_ARG;
    Decoded *D = new Decoded();
    bool ok = false;
    try {
        ok = decode_file(path, n_threads, rank, world, *D);
    } catch (const std::exception &e) {          // e.g. bad_alloc on a corrupt size field: never across the C boundary
        D->error = std::string("decoder failed: ") + e.what();
    }
    if (!ok) {
        g_bam_err = D->error;
        delete D;
        return CORAL_ERR_FORMAT;
    }
    *handle = D;
    return CORAL_OK;
}

extern "C" int coral_bam_decode_open(const char *path, int32_t n_threads, void **handle) {
    return coral_bam_decode_range(path, n_threads, 0, 1, handle);
}

extern "C" int coral_bam_decode_sizes(void *handle, int64_t sizes[8]) {
    if (!handle || !sizes) return CORAL_ERR_ARG;
    Decoded *D = (Decoded *)handle;
    int64_t nb = (int64_t)D->names.blob.size(), rb = 0;
    for (auto &s : D->ref_names) rb += (int64_t)s.size() + 1;
    sizes[0] = (int64_t)D->tid.size();
    sizes[1] = (int64_t)D->cigar.size();
    sizes[2] = (int64_t)D->sa_nm.size();
    sizes[3] = (int64_t)D->na_pos.size();
    sizes[4] = (int64_t)D->names.size();
    sizes[5] = nb;
    sizes[6] = (int64_t)D->ref_names.size();
    sizes[7] = rb;
    return CORAL_OK;
}

extern "C" int coral_bam_decode_stats(void *handle, int64_t stats[3], double *seconds) {
    if (!handle || !stats || !seconds) return CORAL_ERR_ARG;
    Decoded *D = (Decoded *)handle;
    stats[0] = D->compressed_bytes;
    stats[1] = D->uncompressed_bytes;
    stats[2] = D->n_blocks;
    *seconds = D->seconds;
    return CORAL_OK;
}

extern "C" int coral_bam_decode_fill(void *handle, int32_t *tid, int32_t *pos, int32_t *end, int32_t *flag, int32_t *mapq,
                                     int32_t *qlen, int32_t *has_seq, int32_t *nm, int32_t *name_id, int32_t *n_cigar,
                                     int64_t *cigar_off, uint32_t *cigar, int64_t *sa_off, int32_t *sa, int32_t *sa_nm,
                                     int64_t *na_rec, int32_t *na_pos, char *names, int64_t *name_off, char *ref_names,
                                     int32_t *ref_lens) {
    if (!handle) return CORAL_ERR_ARG;
    Decoded *D = (Decoded *)handle;
    auto cp = [](auto *dst, const auto &v) { if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(tid, D->tid); cp(pos, D->pos); cp(end, D->end); cp(flag, D->flag); cp(mapq, D->mapq); cp(qlen, D->qlen);
    cp(has_seq, D->has_seq); cp(nm, D->nm); cp(name_id, D->name_id); cp(n_cigar, D->n_cigar);
    cp(cigar_off, D->cigar_off); cp(cigar, D->cigar); cp(sa_off, D->sa_off); cp(sa, D->sa); cp(sa_nm, D->sa_nm);
    cp(na_rec, D->na_rec); cp(na_pos, D->na_pos); cp(ref_lens, D->ref_lens);
    cp(names, D->names.blob); cp(name_off, D->names.off);
    char *w = ref_names;
    for (auto &s : D->ref_names) { memcpy(w, s.c_str(), s.size() + 1); w += s.size() + 1; }
    return CORAL_OK;
}

extern "C" int coral_bam_decode_close(void *handle) {
    delete (Decoded *)handle;
    return CORAL_OK;
}

// SoA records -> BAM file (tests / benchmarks; the product only reads BAM).  SEQ is deterministic ACGT (hash of seed and
// record ordinal) with N at the listed aligned non-ACGT positions, QUAL absent, tags NM:i, SA:Z and CG:B,I for long CIGARs.
extern "C" int coral_bam_write(const char *path, int64_t n_rec, const int32_t *tid, const int32_t *pos, const int32_t *flag,
                               const int32_t *mapq, const int32_t *qlen, const int32_t *has_seq, const int32_t *nm,
                               const int32_t *name_id, const int32_t *n_cigar, const int64_t *cigar_off, const uint32_t *cigar,
                               const int64_t *sa_off, const int32_t *sa, const int32_t *sa_nm, int64_t n_na, const int64_t *na_rec,
                               const int32_t *na_pos, const char *const *names, int32_t n_ref, const char *const *ref_names,
                               const int32_t *ref_lens, uint32_t seed, int32_t level, int32_t n_threads) {
    if (!path || n_rec < 0 || n_ref < 0 || (n_ref > 0 && (!ref_names || !ref_lens))) return CORAL_ERR_ARG;
    if (n_rec > 0 && (!tid || !pos || !flag || !mapq || !qlen || !has_seq || !nm || !name_id || !n_cigar || !cigar_off || !sa_off || !names))
        return CORAL_ERR_ARG;
    FILE *fp = fopen(path, "wb");
    if (!fp) { g_bam_err = std::string("cannot create ") + path; return CORAL_ERR_ARG; }
    WriteJob J{n_rec, tid, pos, flag, mapq, qlen, has_seq, nm, name_id, n_cigar, cigar_off, cigar, sa_off, sa, sa_nm, n_na, na_rec, na_pos,
               names, n_ref, ref_names, seed};
    bool ok = true;
    {   // header
        std::vector<uint8_t> h;
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int32_t r = 0; r < n_ref; ++r) text += std::string("@SQ\tSN:") + ref_names[r] + "\tLN:" + std::to_string(ref_lens[r]) + "\n";
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, (uint32_t)text.size());
        h.insert(h.end(), text.begin(), text.end());
        put32(h, (uint32_t)n_ref);
        for (int32_t r = 0; r < n_ref; ++r) {
            const size_t ln = strlen(ref_names[r]) + 1;
            put32(h, (uint32_t)ln);
            h.insert(h.end(), (const uint8_t *)ref_names[r], (const uint8_t *)ref_names[r] + ln);
            put32(h, (uint32_t)ref_lens[r]);
        }
        std::vector<uint8_t> out;
        ok = bgzf_compress(h.data(), h.size(), level, out) && fwrite(out.data(), 1, out.size(), fp) == out.size();
    }
    // records: tasks of consecutive records, compressed independently (their last block is short), written in order
    const int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    const int64_t per_task = 256;
    const int64_t n_tasks = (n_rec + per_task - 1) / per_task;
    for (int64_t wave = 0; wave < n_tasks && ok; wave += 4 * nt) {
        const int64_t w1 = std::min<int64_t>(n_tasks, wave + 4 * nt);
        std::vector<std::vector<uint8_t>> outs((size_t)(w1 - wave));
        std::atomic<int64_t> next{wave};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            std::vector<uint8_t> raw, seq;
            for (;;) {
                const int64_t t = next.fetch_add(1);
                if (t >= w1) return;
                raw.clear();
                for (int64_t i = t * per_task; i < std::min(n_rec, (t + 1) * per_task); ++i) write_record(J, i, raw, seq);
                if (!bgzf_compress(raw.data(), raw.size(), level, outs[(size_t)(t - wave)])) bad = true;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        ok = !bad;
        for (auto &o : outs) ok = ok && fwrite(o.data(), 1, o.size(), fp) == o.size();
    }
    static const uint8_t EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(EOF_BLOCK, 1, 28, fp) == 28;
    ok = (fclose(fp) == 0) && ok;
    if (!ok) { g_bam_err = "writing the BAM file failed"; return CORAL_ERR_FORMAT; }
    return CORAL_OK;
}
