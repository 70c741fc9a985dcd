// coral_bam_common.h — host-side pieces shared by the two BAM decoders of libcoral_hip.so: the CPU pipeline (coral_bam.cpp)
// and the GPU pipeline (coral_bamgpu.hip: BGZF inflate and record parsing on the device, these helpers for the file layout,
// the SA-tag tokeniser, the read-name table and the rare non-ACGT records).
#pragma once
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

#include "coral_names.h"

namespace coral_bam {


struct Decoded {
    std::vector<int32_t> tid, pos, end, flag, mapq, qlen, has_seq, nm, name_id, n_cigar;
    std::vector<int64_t> cigar_off{0}, sa_off{0};
    std::vector<uint32_t> cigar;
    std::vector<int32_t> sa;      // 8 per row: tid, pos1, strand, c5, m, x, c3, mapq   (c5 = -2: unparseable shape)
    std::vector<int32_t> sa_nm;
    std::vector<int64_t> na_rec;
    std::vector<int32_t> na_pos;
    coral_names::NameIndex names;           // read names: blob + offsets, ids in first-seen order (coral_names.h)
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    std::string error;
    // statistics of the decode (coral_bam_decode_stats)
    int64_t compressed_bytes = 0, uncompressed_bytes = 0, n_blocks = 0;
    double seconds = 0.0;
};

struct Partial {   // what stage 3 produces for one chunk
    std::vector<int32_t> tid, pos, end, flag, mapq, qlen, has_seq, nm, n_cigar;
    std::vector<uint32_t> cigar;            // padded per record
    std::vector<int64_t> cigar_len;         // padded op count per record
    std::vector<int32_t> sa, sa_nm, sa_cnt;
    std::vector<int64_t> na_rec_local;
    std::vector<int32_t> na_pos;
    std::vector<char> names;                // NUL-separated
    std::string error;
};

inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

static const int REF_ADV[16] = {1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const int IS_ALN[16] = {1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const int QRY_ADV[16] = {1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};

typedef std::unordered_map<std::string, int> RefIds;

// Tokenise one SA entry "rname,pos,strand,CIGAR,mapQ,NM" into 8 ints + nm.  The CIGAR must be
// [c5 S] m M [x I | x D] [c3 S]; anything else containing S and M is marked c5 = -2 (the reference raises
// KeyError for it, cigar_parsing.py:255); a CIGAR without S or without M gets c5 = c3 = 0 / m = 0 as parsed.
inline bool parse_sa_entry(const char *s, const char *e, const RefIds &ref_id, int32_t out[8], int32_t *nm) {
    const char *f[6];
    const char *fe[6];
    int nf = 0;
    const char *p = s;
    f[0] = s;
    for (; p < e && nf < 6; ++p)
        if (*p == ',') {
            fe[nf++] = p;
            if (nf < 6) f[nf] = p + 1;
        }
    if (nf == 5) fe[nf++] = e;
    if (nf != 6) return false;
    auto it = ref_id.find(std::string(f[0], fe[0]));
    out[0] = (it == ref_id.end()) ? -1 : it->second;
    auto to_int = [](const char *a, const char *b) {
        bool neg = a < b && *a == '-';
        if (neg || (a < b && *a == '+')) ++a;
        int64_t v = 0;
        for (; a < b && *a >= '0' && *a <= '9'; ++a) v = v * 10 + (*a - '0');
        return (int32_t)(neg ? -v : v);
    };
    out[1] = to_int(f[1], fe[1]);
    out[2] = (*f[2] == '-') ? 1 : 0;
    out[7] = to_int(f[4], fe[4]);
    *nm = to_int(f[5], fe[5]);
    // CIGAR
    int64_t nums[8];
    char ops[8];
    int n = 0;
    int64_t cur = 0;
    bool overflow = false;
    for (const char *c = f[3]; c < fe[3]; ++c) {
        if (*c >= '0' && *c <= '9') cur = cur * 10 + (*c - '0');
        else {
            if (n < 8) { nums[n] = cur; ops[n] = *c; ++n; } else overflow = true;
            cur = 0;
        }
    }
    bool hasS = false, hasM = false;
    for (int i = 0; i < n; ++i) { hasS |= ops[i] == 'S'; hasM |= ops[i] == 'M'; }
    out[3] = out[4] = out[5] = out[6] = 0;
    if (!hasS || !hasM) {      // reference: the whole read becomes ([], [], []) (cigar_parsing.py:248-253)
        out[4] = 0;
        return true;
    }
    int i = 0;
    bool ok = !overflow;
    if (ok && i < n && ops[i] == 'S') out[3] = (int32_t)nums[i++];
    if (ok && i < n && ops[i] == 'M') out[4] = (int32_t)nums[i++]; else ok = false;
    if (ok && i < n && (ops[i] == 'I' || ops[i] == 'D')) { out[5] = (ops[i] == 'I') ? (int32_t)nums[i] : -(int32_t)nums[i]; ++i; }
    if (ok && i < n && ops[i] == 'S') out[6] = (int32_t)nums[i++];
    if (!ok || i != n || (out[3] == 0 && out[6] == 0)) { out[3] = -2; }
    return true;
}

// true when every 4-bit base code of the packed sequence is A, C, G or T (1, 2, 4, 8); 8 bytes at a time: a nibble x is a
// power of two iff x != 0 and (x & (x - 1)) == 0
inline bool all_acgt(const uint8_t *seq, uint32_t l_seq) {
    const uint32_t full = l_seq / 2;
    uint32_t k = 0;
    const uint64_t LO = 0x0f0f0f0f0f0f0f0full, ONE = 0x0101010101010101ull;
    for (; k + 8 <= full; k += 8) {
        uint64_t w;
        memcpy(&w, seq + k, 8);
        const uint64_t a = w & LO, b = (w >> 4) & LO;
        // per byte (values 0..15): bad if zero or not a power of two
        const uint64_t a1 = (a - ONE) & LO & a, b1 = (b - ONE) & LO & b;          // x & (x - 1) per byte (no borrow across bytes for x >= 1;
        const uint64_t az = ((a | 0x1010101010101010ull) - ONE) & 0x1010101010101010ull;      // for x == 0 the borrow is caught by the zero test)
        const uint64_t bz = ((b | 0x1010101010101010ull) - ONE) & 0x1010101010101010ull;
        // az / bz have bit 4 set in every byte where x >= 1; a zero byte clears it
        if (a1 | b1 | (az ^ 0x1010101010101010ull) | (bz ^ 0x1010101010101010ull)) {
            // the fast test is conservative around borrows: confirm byte by byte
            for (uint32_t j = k; j < k + 8; ++j) {
                const uint8_t hi = seq[j] >> 4, lo = seq[j] & 15;
                if (!((hi == 1 || hi == 2 || hi == 4 || hi == 8) && (lo == 1 || lo == 2 || lo == 4 || lo == 8))) return false;
            }
        }
    }
    for (; k < full; ++k) {
        const uint8_t hi = seq[k] >> 4, lo = seq[k] & 15;
        if (!((hi == 1 || hi == 2 || hi == 4 || hi == 8) && (lo == 1 || lo == 2 || lo == 4 || lo == 8))) return false;
    }
    if (l_seq & 1) {
        const uint8_t hi = seq[full] >> 4;
        if (!(hi == 1 || hi == 2 || hi == 4 || hi == 8)) return false;
    }
    return true;
}

// Decode one BAM record (p points at refID, i.e. after block_size) into the partial.
inline bool decode_record(const uint8_t *p, uint32_t block_size, const RefIds &ref_id, Partial &o, std::string &err) {
    if (block_size < 32) { err = "record shorter than its fixed fields"; return false; }
    const int32_t refID = (int32_t)rd32(p), pos = (int32_t)rd32(p + 4);
    const uint32_t l_read_name = p[8], mapq = p[9];
    uint32_t n_cigar_op = rd16(p + 12);
    const uint32_t flag = rd16(p + 14), l_seq = rd32(p + 16);
    const uint8_t *name = p + 32;
    const uint8_t *cig = name + l_read_name;
    const uint8_t *seq = cig + 4ull * n_cigar_op;
    const uint8_t *qual = seq + ((uint64_t)l_seq + 1) / 2;
    const uint8_t *tags = qual + l_seq;
    const uint8_t *endp = p + block_size;
    if (tags > endp || l_read_name == 0) { err = "record fields overrun the record"; return false; }
    // tags: NM, SA, CG
    int32_t nm = 0;
    const char *sa = nullptr;
    const uint8_t *cg = nullptr;
    uint32_t cg_n = 0;
    for (const uint8_t *t = tags; t + 3 <= endp;) {
        const char a = (char)t[0], b = (char)t[1], ty = (char)t[2];
        const uint8_t *v = t + 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t *z = (const uint8_t *)memchr(v, 0, (size_t)(endp - v)); sz = z ? (size_t)(z - v) + 1 : (size_t)(endp - v) + 1; break; }
            case 'B': {
                if (v + 5 > endp) { err = "truncated B tag"; return false; }
                const char sub = (char)v[0];
                const uint32_t cnt = rd32(v + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if (a == 'C' && b == 'G' && sub == 'I') { cg = v + 5; cg_n = cnt; }
                sz = 5 + es * (size_t)cnt;
                break;
            }
            default: err = "unknown tag type"; return false;
        }
        if (v + sz > endp) { err = "tag overruns the record"; return false; }
        if (a == 'N' && b == 'M') {
            switch (ty) {
                case 'c': nm = (int8_t)v[0]; break;
                case 'C': nm = v[0]; break;
                case 's': nm = (int16_t)rd16(v); break;
                case 'S': nm = rd16(v); break;
                case 'i': case 'I': nm = (int32_t)rd32(v); break;
                default: break;
            }
        } else if (a == 'S' && b == 'A' && ty == 'Z') {
            sa = (const char *)v;
        }
        t = v + sz;
    }
    // long CIGARs live in the CG tag (SAM spec §4.2.2): placeholder is <l_seq>S<rlen>N
    const uint8_t *cig_src = cig;
    if (cg && n_cigar_op == 2 && (rd32(cig) & 15u) == 4 && (rd32(cig) >> 4) == l_seq && (rd32(cig + 4) & 15u) == 3) {
        cig_src = cg;
        n_cigar_op = cg_n;
    }
    int64_t rlen = 0, qinf = 0;
    const size_t c0 = o.cigar.size();
    const size_t padded = ((size_t)n_cigar_op + 3) & ~(size_t)3;
    o.cigar.resize(c0 + padded);
    uint32_t *dst = o.cigar.data() + c0;
    if (n_cigar_op) memcpy(dst, cig_src, 4ull * n_cigar_op);
    for (size_t k = n_cigar_op; k < padded; ++k) dst[k] = 15u;
    for (uint32_t k = 0; k < n_cigar_op; ++k) {
        const uint32_t v = dst[k];
        rlen += REF_ADV[v & 15] ? (v >> 4) : 0;
        qinf += QRY_ADV[v & 15] ? (v >> 4) : 0;
    }
    o.cigar_len.push_back((int64_t)padded);
    if ((flag & 4) || n_cigar_op == 0) rlen = 0;                 // htslib bam_endpos
    o.tid.push_back(refID);
    o.pos.push_back(pos);
    o.end.push_back(pos + (int32_t)(rlen > 0 ? rlen : 1));
    o.flag.push_back((int32_t)flag);
    o.mapq.push_back((int32_t)mapq);
    o.has_seq.push_back(l_seq > 0 ? 1 : 0);
    o.qlen.push_back(l_seq > 0 ? (int32_t)l_seq : (int32_t)qinf);
    o.nm.push_back(nm);
    o.n_cigar.push_back((int32_t)n_cigar_op);
    o.names.insert(o.names.end(), (const char *)name, (const char *)name + l_read_name - 1);
    o.names.push_back('\0');
    // SA rows
    int32_t cnt = 0;
    if (sa) {
        const char *s = sa;
        while (*s) {
            const char *e = s;
            while (*e && *e != ';') ++e;
            if (e > s) {
                int32_t row[8], snm = 0;
                if (!parse_sa_entry(s, e, ref_id, row, &snm)) { err = "malformed SA entry"; return false; }
                o.sa.insert(o.sa.end(), row, row + 8);
                o.sa_nm.push_back(snm);
                ++cnt;
            }
            s = (*e == ';') ? e + 1 : e;
        }
    }
    o.sa_cnt.push_back(cnt);
    // aligned non-ACGT bases
    if (l_seq > 0 && !(flag & 4) && n_cigar_op > 0 && !all_acgt(seq, l_seq)) {
        int64_t q = 0, r = pos;
        const int64_t local = (int64_t)o.tid.size() - 1;
        for (uint32_t k = 0; k < n_cigar_op; ++k) {
            const uint32_t v = dst[k], op = v & 15, len = v >> 4;
            if (IS_ALN[op]) {
                for (uint32_t j = 0; j < len && q + j < l_seq; ++j) {
                    const uint64_t qi = (uint64_t)(q + j);
                    const uint8_t code = (qi & 1) ? (seq[qi >> 1] & 15) : (seq[qi >> 1] >> 4);
                    if (!(code == 1 || code == 2 || code == 4 || code == 8)) {
                        o.na_rec_local.push_back(local);
                        o.na_pos.push_back((int32_t)(r + j));
                    }
                }
            }
            if (QRY_ADV[op]) q += len;
            if (REF_ADV[op]) r += len;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// a small pool: tasks run on n - 1 threads and, while waiting, on the thread that waits
// ---------------------------------------------------------------------------------------------
class Pool {
public:
    explicit Pool(int n) {
        for (int i = 1; i < n; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void submit(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }
    bool help_one() {                          // run one queued task on the calling thread, if any
        std::function<void()> f;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (q_.empty()) return false;
            f = std::move(q_.front());
            q_.pop_front();
        }
        f();
        return true;
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
};

struct Flag {                                  // one-shot completion flag
    std::atomic<int> v{0};
    void set() { v.store(1, std::memory_order_release); }
    bool get() const { return v.load(std::memory_order_acquire) != 0; }
};

// ---------------------------------------------------------------------------------------------
// BGZF
// ---------------------------------------------------------------------------------------------
struct Block {
    uint64_t off;        // file offset of the block
    uint32_t hdr;        // header bytes (12 + xlen)
    uint32_t csize;      // whole block (BSIZE + 1)
    uint32_t isize;      // uncompressed bytes
};

// Parse a BGZF block header at `p` (n bytes available).  Returns false when it is not one.
inline bool bgzf_header(const uint8_t *p, uint64_t n, Block &b) {
    if (n < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return false;
    const uint32_t xlen = rd16(p + 10);
    if (n < 12ull + xlen) return false;
    int bsize = -1;
    for (uint32_t i = 0; i + 4 <= xlen;) {
        const uint32_t slen = rd16(p + 12 + i + 2);
        if (p[12 + i] == 'B' && p[12 + i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = rd16(p + 12 + i + 4);
        i += 4 + slen;
    }
    if (bsize < 0) return false;
    const uint64_t csize = (uint64_t)bsize + 1;
    if (csize < 12ull + xlen + 8 || csize > n) return false;
    b.hdr = 12 + xlen;
    b.csize = (uint32_t)csize;
    b.isize = rd32(p + csize - 4);
    return b.isize <= 65536;
}

struct MappedFile {
    const uint8_t *data = nullptr;
    uint64_t size = 0;
    int fd = -1;
    bool open(const char *path, std::string &err) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0) { err = "cannot stat the file"; return false; }
        size = (uint64_t)st.st_size;
        if (size == 0) { err = "empty file"; return false; }
        void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) { err = "cannot map the file"; return false; }
        data = (const uint8_t *)p;
        (void)madvise(p, size, MADV_SEQUENTIAL);
        return true;
    }
    ~MappedFile() {
        if (data) munmap((void *)data, size);
        if (fd >= 0) ::close(fd);
    }
};

// First BGZF block starting at or after `from`: magic + BC subfield, and the two blocks that follow must parse as well.
inline bool find_block(const MappedFile &f, uint64_t from, uint64_t *at) {
    for (uint64_t p = from; p + 18 <= f.size; ++p) {
        if (f.data[p] != 31 || f.data[p + 1] != 139) continue;
        uint64_t q = p;
        bool ok = true;
        for (int k = 0; k < 3 && ok && q < f.size; ++k) {
            Block b;
            ok = bgzf_header(f.data + q, f.size - q, b);
            if (ok) q += b.csize;
        }
        if (ok) { *at = p; return true; }
    }
    return false;
}

inline bool inflate_block(z_stream &zs, const MappedFile &f, const Block &b, uint8_t *out) {
    if (b.isize == 0) return true;
    if (inflateReset(&zs) != Z_OK) return false;
    zs.next_in = (Bytef *)(f.data + b.off + b.hdr);
    zs.avail_in = (uInt)(b.csize - b.hdr - 8);
    zs.next_out = out;
    zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return false;
    // the trailer's CRC-32 of the inflated bytes (what htslib's bgzf_read_block checks; the GPU pipeline does the same)
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, b.isize) == rd32(f.data + b.off + b.csize - 8);
}

struct ZStream {
    z_stream zs;
    bool ok;
    ZStream() { memset(&zs, 0, sizeof(zs)); ok = inflateInit2(&zs, -15) == Z_OK; }
    ~ZStream() { if (ok) inflateEnd(&zs); }
};

// ---------------------------------------------------------------------------------------------
// plausibility of a BAM record at `p` (n bytes available): used to find the first record of a byte range
// ---------------------------------------------------------------------------------------------
inline bool plausible_record(const uint8_t *p, uint64_t n, int32_t n_ref, uint64_t *len) {
    if (n < 36) return false;
    const uint32_t bs = rd32(p);
    if (bs < 34 || bs > (1u << 29)) return false;
    const int32_t refID = (int32_t)rd32(p + 4), pos = (int32_t)rd32(p + 8);
    const uint32_t l_name = p[12], n_cig = rd16(p + 16), l_seq = rd32(p + 20);
    const int32_t mate = (int32_t)rd32(p + 24), mpos = (int32_t)rd32(p + 28);
    if (refID < -1 || refID >= n_ref || mate < -1 || mate >= n_ref || pos < -1 || mpos < -1) return false;
    if (l_name < 2 || l_seq > (1u << 29)) return false;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (fixed > bs) return false;
    if (n >= 36ull + l_name) {                                   // read name: printable, NUL-terminated
        const uint8_t *nm = p + 36;
        if (nm[l_name - 1] != 0) return false;
        for (uint32_t k = 0; k + 1 < l_name; ++k)
            if (nm[k] < 33 || nm[k] > 126) return false;
    }
    *len = 4ull + bs;
    return true;
}

// BAM header: magic, text, reference list (inflated from block 0 on, as many blocks as it takes).  Fills D.ref_names /
// D.ref_lens / ref_id and the header's length in the uncompressed stream.
inline bool read_bam_header(const MappedFile &f, Decoded &D, RefIds &ref_id, size_t *hdr_bytes) {
    std::vector<uint8_t> head;
    ZStream z;
    if (!z.ok) { D.error = "zlib init failed"; return false; }
    uint64_t at = 0;
    auto more = [&]() -> bool {
        Block b;
        if (at >= f.size || !bgzf_header(f.data + at, f.size - at, b)) return false;
        b.off = at;
        const size_t o = head.size();
        head.resize(o + b.isize);
        if (!inflate_block(z.zs, f, b, head.data() + o)) return false;
        at += b.csize;
        return true;
    };
    auto need = [&](size_t n) { while (head.size() < n) if (!more()) return false; return true; };
    if (!need(12) || memcmp(head.data(), "BAM\1", 4) != 0) { D.error = "not a BAM file"; return false; }
    const uint32_t l_text = rd32(head.data() + 4);
    if (!need(12 + (size_t)l_text)) { D.error = "truncated BAM header"; return false; }
    size_t cur = 8 + l_text;
    const uint32_t n_ref = rd32(head.data() + cur);
    cur += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!need(cur + 4)) { D.error = "truncated reference list"; return false; }
        const uint32_t l_name = rd32(head.data() + cur);
        if (!need(cur + 8 + (size_t)l_name)) { D.error = "truncated reference list"; return false; }
        std::string nm((const char *)head.data() + cur + 4, l_name ? l_name - 1 : 0);
        D.ref_lens.push_back((int32_t)rd32(head.data() + cur + 4 + l_name));
        ref_id[nm] = (int)i;
        D.ref_names.push_back(nm);
        cur += 8 + l_name;
    }
    *hdr_bytes = cur;
    return true;
}

void set_error(const std::string &msg);         // coral_bam_last_error() of the calling thread (defined in coral_bam.cpp)

}  // namespace coral_bam
