// coral_bam.cpp — host-side BAM (BGZF) decoder of libcoral_hip.so: file -> structure-of-arrays records.
//
// Replaces what the reference gets from pysam.AlignmentFile(path, 'rb') + the whole-file fetch()
// (/root/reference/src/infer_breakpoint_graph.py:65, :140-158): every mapped-or-unmapped record is decoded ONCE
// into the SoA layout of include/coral_hip.h (CIGAR padded to 16 bytes with op 15), the SA tag is tokenised
// into numeric rows, NM is extracted, and aligned non-ACGT bases are listed (pysam count_coverage counts only
// A/C/G/T).  BGZF blocks are inflated by a pool of threads (zlib raw inflate), a batch at a time, so the
// uncompressed SEQ/QUAL bytes never accumulate in memory.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
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
};

struct Partial {   // what one worker thread produces for a contiguous run of records
    std::vector<int32_t> tid, pos, end, flag, mapq, qlen, has_seq, nm, n_cigar;
    std::vector<uint32_t> cigar;            // padded per record
    std::vector<int64_t> cigar_len;         // padded op count per record
    std::vector<int32_t> sa, sa_nm, sa_cnt;
    std::vector<int64_t> na_rec_local;
    std::vector<int32_t> na_pos;
    std::vector<std::string> names;
};

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

const int REF_ADV[16] = {1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const int IS_ALN[16] = {1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const int QRY_ADV[16] = {1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};

// Tokenise one SA entry "rname,pos,strand,CIGAR,mapQ,NM" into 8 ints + nm.  The CIGAR must be
// [c5 S] m M [x I | x D] [c3 S]; anything else containing S and M is marked c5 = -2 (the reference raises
// KeyError for it, cigar_parsing.py:255); a CIGAR without S or without M gets c5 = c3 = 0 / m = 0 as parsed.
bool parse_sa_entry(const char *s, const char *e, const std::unordered_map<std::string, int> &ref_id, int32_t out[8], int32_t *nm) {
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
    out[1] = (int32_t)strtol(std::string(f[1], fe[1]).c_str(), nullptr, 10);
    out[2] = (*f[2] == '-') ? 1 : 0;
    out[7] = (int32_t)strtol(std::string(f[4], fe[4]).c_str(), nullptr, 10);
    *nm = (int32_t)strtol(std::string(f[5], fe[5]).c_str(), nullptr, 10);
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

// Decode one BAM record (p points at refID, i.e. after block_size) into the partial.
bool decode_record(const uint8_t *p, uint32_t block_size, const std::unordered_map<std::string, int> &ref_id, Partial &o, std::string &err) {
    if (block_size < 32) { err = "record shorter than its fixed fields"; return false; }
    const int32_t refID = (int32_t)rd32(p), pos = (int32_t)rd32(p + 4);
    const uint32_t l_read_name = p[8], mapq = p[9];
    uint32_t n_cigar_op = rd16(p + 12);
    const uint32_t flag = rd16(p + 14), l_seq = rd32(p + 16);
    const uint8_t *name = p + 32;
    const uint8_t *cig = name + l_read_name;
    const uint8_t *seq = cig + 4ull * n_cigar_op;
    const uint8_t *qual = seq + (l_seq + 1) / 2;
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
            case 'Z': case 'H': { const uint8_t *z = v; while (z < endp && *z) ++z; sz = (size_t)(z - v) + 1; break; }
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
    for (uint32_t k = 0; k < n_cigar_op; ++k) {
        const uint32_t v = rd32(cig_src + 4ull * k);
        o.cigar.push_back(v);
        rlen += REF_ADV[v & 15] ? (v >> 4) : 0;
        qinf += QRY_ADV[v & 15] ? (v >> 4) : 0;
    }
    while ((o.cigar.size() - c0) & 3) o.cigar.push_back(15u);
    o.cigar_len.push_back((int64_t)(o.cigar.size() - c0));
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
    o.names.emplace_back((const char *)name, l_read_name - 1);
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
    if (l_seq > 0 && !(flag & 4) && n_cigar_op > 0) {
        bool any = false;
        for (uint32_t k = 0; k < (l_seq + 1) / 2 && !any; ++k) {
            const uint8_t hi = seq[k] >> 4, lo = seq[k] & 15;
            const bool hi_ok = hi == 1 || hi == 2 || hi == 4 || hi == 8;
            const bool lo_ok = lo == 1 || lo == 2 || lo == 4 || lo == 8 || (2 * k + 1 >= l_seq);
            any = !(hi_ok && lo_ok);
        }
        if (any) {
            int64_t q = 0, r = pos;
            const int64_t local = (int64_t)o.tid.size() - 1;
            for (uint32_t k = 0; k < n_cigar_op; ++k) {
                const uint32_t v = rd32(cig_src + 4ull * k), op = v & 15, len = v >> 4;
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
    }
    return true;
}

struct Reader {
    FILE *fp = nullptr;
    int n_threads = 1;
    std::vector<uint8_t> carry;     // undecoded tail of the uncompressed stream
    bool eof = false;

    // read a batch of BGZF blocks and inflate them in parallel; append to `out`
    bool next_batch(std::vector<uint8_t> &out, size_t max_blocks, std::string &err) {
        struct Blk { std::vector<uint8_t> comp; uint32_t isize; size_t off; };
        std::vector<Blk> blks;
        size_t total = 0;
        while (blks.size() < max_blocks) {
            uint8_t h[18];
            size_t got = fread(h, 1, 18, fp);
            if (got == 0) { eof = true; break; }
            if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
            const uint32_t xlen = rd16(h + 10);
            // locate the BC subfield (it is first in every htslib-written file, but be general)
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), h + 12, xlen < 6 ? xlen : 6);
            if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != xlen - 6) { err = "truncated BGZF header"; return false; }
            int bsize = -1;
            for (uint32_t i = 0; i + 4 <= xlen;) {
                const uint32_t slen = rd16(extra.data() + i + 2);
                if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = rd16(extra.data() + i + 4);
                i += 4 + slen;
            }
            if (bsize < 0) { err = "BGZF block without BC field"; return false; }
            const size_t remain = (size_t)bsize + 1 - 12 - xlen;     // compressed data + crc32 + isize
            if (remain < 8) { err = "bad BGZF block size"; return false; }
            Blk b;
            b.comp.resize(remain);
            if (fread(b.comp.data(), 1, remain, fp) != remain) { err = "truncated BGZF block"; return false; }
            b.isize = rd32(b.comp.data() + remain - 4);
            b.off = total;
            total += b.isize;
            blks.push_back(std::move(b));
        }
        const size_t base = out.size();
        out.resize(base + total);
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blks.size()) return;
                Blk &b = blks[i];
                if (b.isize == 0) continue;
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
                zs.next_in = b.comp.data();
                zs.avail_in = (uInt)(b.comp.size() - 8);
                zs.next_out = out.data() + base + b.off;
                zs.avail_out = b.isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = true; return; }
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)n_threads, blks.size() ? blks.size() : 1);
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        if (bad) { err = "zlib inflate failed (corrupt BGZF block)"; return false; }
        return true;
    }
};

bool decode_file(const char *path, int n_threads, Decoded &D) {
    Reader R;
    R.fp = fopen(path, "rb");
    if (!R.fp) { D.error = std::string("cannot open ") + path; return false; }
    R.n_threads = n_threads < 1 ? 1 : n_threads;
    std::vector<uint8_t> buf;
    size_t cur = 0;
    auto need = [&](size_t n) -> bool {      // make sure buf[cur .. cur+n) is available
        while (buf.size() - cur < n) {
            if (R.eof) return false;
            if (cur > 0) { buf.erase(buf.begin(), buf.begin() + (long)cur); cur = 0; }
            if (!R.next_batch(buf, 2048, D.error)) return false;
            if (R.eof && buf.size() - cur < n) return false;
        }
        return true;
    };
    // ---- header
    if (!need(12) || memcmp(buf.data() + cur, "BAM\1", 4) != 0) { if (D.error.empty()) D.error = "not a BAM file"; fclose(R.fp); return false; }
    const uint32_t l_text = rd32(buf.data() + cur + 4);
    if (!need(12 + (size_t)l_text)) { D.error = "truncated BAM header"; fclose(R.fp); return false; }
    cur += 8 + l_text;
    const uint32_t n_ref = rd32(buf.data() + cur);
    cur += 4;
    std::unordered_map<std::string, int> ref_id;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!need(4)) { D.error = "truncated reference list"; fclose(R.fp); return false; }
        const uint32_t l_name = rd32(buf.data() + cur);
        if (!need(8 + (size_t)l_name)) { D.error = "truncated reference list"; fclose(R.fp); return false; }
        std::string nm((const char *)buf.data() + cur + 4, l_name ? l_name - 1 : 0);
        D.ref_lens.push_back((int32_t)rd32(buf.data() + cur + 4 + l_name));
        ref_id[nm] = (int)i;
        D.ref_names.push_back(nm);
        cur += 8 + l_name;
    }
    // ---- records, a batch of inflated bytes at a time
    std::unordered_map<std::string, int32_t> name_id;
    for (;;) {
        if (!need(4)) {
            if (!D.error.empty()) { fclose(R.fp); return false; }
            if (buf.size() - cur != 0) { D.error = "trailing bytes after the last record"; fclose(R.fp); return false; }
            break;
        }
        // record boundaries available in the current buffer
        std::vector<size_t> starts;
        size_t p = cur;
        while (buf.size() - p >= 4) {
            const uint32_t bs = rd32(buf.data() + p);
            if (buf.size() - p - 4 < bs) break;
            starts.push_back(p);
            p += 4 + (size_t)bs;
        }
        if (starts.empty()) {            // one record larger than what is buffered: pull more
            const uint32_t bs = rd32(buf.data() + cur);
            if (!need(4 + (size_t)bs)) { if (D.error.empty()) D.error = "truncated record"; fclose(R.fp); return false; }
            continue;
        }
        const size_t nrec = starts.size();
        const int nt = (int)std::min<size_t>((size_t)R.n_threads, (nrec + 255) / 256);
        std::vector<Partial> parts((size_t)nt);
        std::vector<std::string> errs((size_t)nt);
        auto work = [&](int t) {
            const size_t a = nrec * (size_t)t / (size_t)nt, b = nrec * (size_t)(t + 1) / (size_t)nt;
            for (size_t i = a; i < b; ++i) {
                const uint8_t *q = buf.data() + starts[i];
                if (!decode_record(q + 4, rd32(q), ref_id, parts[(size_t)t], errs[(size_t)t])) return;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
        for (auto &e : errs) if (!e.empty()) { D.error = e; fclose(R.fp); return false; }
        for (auto &pt : parts) {
            const int64_t base = (int64_t)D.tid.size();
            auto app = [](std::vector<int32_t> &d, const std::vector<int32_t> &s) { d.insert(d.end(), s.begin(), s.end()); };
            app(D.tid, pt.tid); app(D.pos, pt.pos); app(D.end, pt.end); app(D.flag, pt.flag); app(D.mapq, pt.mapq);
            app(D.qlen, pt.qlen); app(D.has_seq, pt.has_seq); app(D.nm, pt.nm); app(D.n_cigar, pt.n_cigar);
            D.cigar.insert(D.cigar.end(), pt.cigar.begin(), pt.cigar.end());
            for (int64_t l : pt.cigar_len) D.cigar_off.push_back(D.cigar_off.back() + l);
            app(D.sa, pt.sa); app(D.sa_nm, pt.sa_nm);
            for (int32_t c : pt.sa_cnt) D.sa_off.push_back(D.sa_off.back() + c);
            for (int64_t l : pt.na_rec_local) D.na_rec.push_back(base + l);
            app(D.na_pos, pt.na_pos);
            for (auto &nm : pt.names) {
                auto it = name_id.find(nm);
                if (it == name_id.end()) {
                    it = name_id.emplace(nm, (int32_t)D.names.size()).first;
                    D.names.push_back(nm);
                }
                D.name_id.push_back(it->second);
            }
        }
        cur = p;
    }
    fclose(R.fp);
    return true;
}

thread_local std::string g_bam_err;

}  // namespace

extern "C" const char *coral_bam_last_error(void) { return g_bam_err.c_str(); }

extern "C" int coral_bam_decode_open(const char *path, int32_t n_threads, void **handle) {
    if (!path || !handle) return CORAL_ERR_ARG;
    Decoded *D = new Decoded();
    if (!decode_file(path, n_threads, *D)) {
        g_bam_err = D->error;
        delete D;
        return CORAL_ERR_FORMAT;
    }
    *handle = D;
    return CORAL_OK;
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
