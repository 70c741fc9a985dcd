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

namespace {

struct Decoded {
    std::vector<int32_t> tid, pos, end, flag, mapq, qlen, has_seq, nm, name_id, n_cigar;
    std::vector<int64_t> cigar_off{0}, sa_off{0};
    std::vector<uint32_t> cigar;
    std::vector<int32_t> sa;      // 8 per row: tid, pos1, strand, c5, m, x, c3, mapq   (c5 = -2: unparseable shape)
    std::vector<int32_t> sa_nm;
    std::vector<int64_t> na_rec;
    std::vector<int32_t> na_pos;
    std::vector<std::string> names;
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

const int REF_ADV[16] = {1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const int IS_ALN[16] = {1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const int QRY_ADV[16] = {1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};

typedef std::unordered_map<std::string, int> RefIds;

// Tokenise one SA entry "rname,pos,strand,CIGAR,mapQ,NM" into 8 ints + nm.  The CIGAR must be
// [c5 S] m M [x I | x D] [c3 S]; anything else containing S and M is marked c5 = -2 (the reference raises
// KeyError for it, cigar_parsing.py:255); a CIGAR without S or without M gets c5 = c3 = 0 / m = 0 as parsed.
bool parse_sa_entry(const char *s, const char *e, const RefIds &ref_id, int32_t out[8], int32_t *nm) {
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
bool decode_record(const uint8_t *p, uint32_t block_size, const RefIds &ref_id, Partial &o, std::string &err) {
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
bool bgzf_header(const uint8_t *p, uint64_t n, Block &b) {
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
bool find_block(const MappedFile &f, uint64_t from, uint64_t *at) {
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

bool inflate_block(z_stream &zs, const MappedFile &f, const Block &b, uint8_t *out) {
    if (b.isize == 0) return true;
    if (inflateReset(&zs) != Z_OK) return false;
    zs.next_in = (Bytef *)(f.data + b.off + b.hdr);
    zs.avail_in = (uInt)(b.csize - b.hdr - 8);
    zs.next_out = out;
    zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    return rc == Z_STREAM_END && zs.avail_out == 0;
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
bool plausible_record(const uint8_t *p, uint64_t n, int32_t n_ref, uint64_t *len) {
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
    {
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
        hdr_bytes = cur;
    }
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

    std::unordered_map<std::string, int32_t> name_id;
    name_id.reserve(1 << 20);
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
            std::string nm(s, len);
            auto it = name_id.find(nm);
            if (it == name_id.end()) {
                it = name_id.emplace(nm, (int32_t)D.names.size()).first;
                D.names.push_back(nm);
            }
            D.name_id.push_back(it->second);
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
        if (c.bad) { D.error = "zlib inflate failed (corrupt BGZF block)"; return false; }
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
    int64_t nb = 0, rb = 0;
    for (auto &s : D->names) nb += (int64_t)s.size() + 1;
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
                                     int64_t *na_rec, int32_t *na_pos, char *names, char *ref_names, int32_t *ref_lens) {
    if (!handle) return CORAL_ERR_ARG;
    Decoded *D = (Decoded *)handle;
    auto cp = [](auto *dst, const auto &v) { if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(tid, D->tid); cp(pos, D->pos); cp(end, D->end); cp(flag, D->flag); cp(mapq, D->mapq); cp(qlen, D->qlen);
    cp(has_seq, D->has_seq); cp(nm, D->nm); cp(name_id, D->name_id); cp(n_cigar, D->n_cigar);
    cp(cigar_off, D->cigar_off); cp(cigar, D->cigar); cp(sa_off, D->sa_off); cp(sa, D->sa); cp(sa_nm, D->sa_nm);
    cp(na_rec, D->na_rec); cp(na_pos, D->na_pos); cp(ref_lens, D->ref_lens);
    char *w = names;
    for (auto &s : D->names) { memcpy(w, s.c_str(), s.size() + 1); w += s.size() + 1; }
    w = ref_names;
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
