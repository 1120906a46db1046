// pv_io.cpp — native BAM/BAI and FASTA/FAI readers for the image-builder input.
//
// Replaces, without htslib (not available offline; the reference fetches htslib 1.9 at configure time,
// pepper/modules/htslib.cmake:8-10):
//   BAM_handler::get_reads            pepper_variant/modules/cpp/bam_handler.cpp:115-451
//   FASTA_handler::get_reference_sequence / get_chromosome_sequence_length / names
//                                     pepper_variant/modules/cpp/fasta_handler.cpp:18-56
// Output is the flat SoA read layout of include/pepper_hip.h (pv_batch_in) for ONE region, i.e. exactly the
// fields of type_read the image builder consumes, after the reference's region clipping:
//   * records with QC-fail / duplicate / secondary / unmapped flags are dropped, supplementary unless asked
//     for, MAPQ < min_mapq dropped (:137-150);
//   * the CIGAR walk stops at the first op that starts beyond `stop` (:186-188);
//   * M/=/X: the part left of `start` is skipped, bases with pos <= stop are kept, the op is re-emitted with
//     the kept length and its ORIGINAL code (:191-239);
//   * I and S: kept (with their bases) only when start <= pos <= stop and a base has already been kept,
//     otherwise only the query index advances (:240-275);
//   * D and N: kept (length clipped at stop) under the same condition, else the position advances (:276-301);
//   * H: ignored; P/B: no state change; a read is returned only if it kept at least one base (:432-445).
// BGZF/BAM/BAI/FAI are implemented from the SAM/BAM format specification (hts-specs SAMv1 section 4, 5).
// Parity note: htslib is absent, so this reader is pinned by this repository's own writer-based tests only.
#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pepper_io.h"

static thread_local char g_ioerr[512] = "";
static void io_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ioerr, sizeof(g_ioerr), fmt, ap);
    va_end(ap);
}
extern "C" const char* pvio_last_error(void) { return g_ioerr; }

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// ---- BGZF ---------------------------------------------------------------------------------------------------
struct Bgzf {
    FILE* f = nullptr;
    int64_t block_coffset = -1;  // compressed offset of the block held in `buf`
    int64_t next_coffset = 0;
    std::vector<uint8_t> buf;    // uncompressed block
    size_t pos = 0;
    std::vector<uint8_t> cbuf;

    bool load_block(int64_t coffset) {
        if (fseeko(f, coffset, SEEK_SET) != 0) return false;
        uint8_t h[18];
        if (fread(h, 1, 18, f) != 18) { buf.clear(); pos = 0; block_coffset = coffset; next_coffset = coffset; return false; }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { io_err("not a BGZF block at offset %lld", (long long)coffset); return false; }
        const int xlen = rd16(h + 10);
        // find the BC subfield (it is the first one in practice; scan to be safe)
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), h + 12, std::min(xlen, 6));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != (size_t)(xlen - 6)) return false;
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = rd16(&extra[i + 2]);
            if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = rd16(&extra[i + 4]);
            i += 4 + slen;
        }
        if (bsize < 0) { io_err("BGZF block without BC field"); return false; }
        const int clen = bsize + 1 - 12 - xlen - 8;
        if (clen < 0) { io_err("corrupt BGZF block size"); return false; }
        cbuf.resize(clen + 8);
        if (fread(cbuf.data(), 1, clen + 8, f) != (size_t)(clen + 8)) { io_err("truncated BGZF block"); return false; }
        const uint32_t isize = rd32(&cbuf[clen + 4]);
        buf.resize(isize);
        if (isize) {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { io_err("inflateInit2 failed"); return false; }
            zs.next_in = cbuf.data(); zs.avail_in = clen;
            zs.next_out = buf.data(); zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) { io_err("inflate failed (%d)", rc); return false; }
        }
        block_coffset = coffset;
        next_coffset = coffset + bsize + 1;
        pos = 0;
        return true;
    }
    bool seek(uint64_t voffset) {
        const int64_t co = (int64_t)(voffset >> 16);
        if (co != block_coffset && !load_block(co)) return false;
        pos = voffset & 0xFFFF;
        return pos <= buf.size();
    }
    uint64_t tell() const { return ((uint64_t)block_coffset << 16) | (uint64_t)pos; }
    // read exactly n bytes; false at EOF / error
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos >= buf.size()) {
                if (!load_block(next_coffset)) return false;
                if (buf.empty()) {  // empty block (EOF marker) — try the next one
                    if (feof(f)) return false;
                    continue;
                }
            }
            const size_t k = std::min(n, buf.size() - pos);
            memcpy(d, buf.data() + pos, k);
            d += k; pos += k; n -= k;
        }
        return true;
    }
};

// ---- BAM + BAI -------------------------------------------------------------------------------------------------
struct Chunk { uint64_t beg, end; };
struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;
};

struct pv_bam {
    Bgzf z;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::vector<RefIndex> index;
    // last query result (owned here, pointers handed to the caller)
    std::vector<int64_t> pos, pos_end, base_off, cigar_off, name_off;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> is_rev, mapq, bases, quals;
    std::vector<int32_t> hp;
    std::vector<uint32_t> cigar;
    std::vector<char> names;
};

static bool load_bai(pv_bam* b, const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n);
    if (fread(d.data(), 1, n, f) != (size_t)n) { fclose(f); return false; }
    fclose(f);
    if (n < 8 || memcmp(d.data(), "BAI\1", 4) != 0) { io_err("%s is not a BAI index", path.c_str()); return false; }
    size_t p = 4;
    const int n_ref = (int)rd32(&d[p]); p += 4;
    b->index.assign(n_ref, RefIndex());
    for (int r = 0; r < n_ref; r++) {
        if (p + 4 > d.size()) return false;
        const int n_bin = (int)rd32(&d[p]); p += 4;
        for (int k = 0; k < n_bin; k++) {
            const uint32_t bin = rd32(&d[p]); p += 4;
            const int n_chunk = (int)rd32(&d[p]); p += 4;
            std::vector<Chunk>& v = b->index[r].bins[bin];
            for (int c = 0; c < n_chunk; c++) {
                Chunk ch; ch.beg = rd64(&d[p]); ch.end = rd64(&d[p + 8]); p += 16;
                v.push_back(ch);
            }
        }
        const int n_intv = (int)rd32(&d[p]); p += 4;
        b->index[r].linear.resize(n_intv);
        for (int k = 0; k < n_intv; k++) { b->index[r].linear[k] = rd64(&d[p]); p += 8; }
    }
    return true;
}

extern "C" pv_bam* pvio_bam_open(const char* path) {
    pv_bam* b = new pv_bam();
    b->z.f = fopen(path, "rb");
    if (!b->z.f) { io_err("cannot open %s", path); delete b; return nullptr; }
    uint8_t h[12];
    if (!b->z.seek(0) || !b->z.read(h, 8) || memcmp(h, "BAM\1", 4) != 0) { io_err("%s is not a BAM file", path); fclose(b->z.f); delete b; return nullptr; }
    const uint32_t l_text = rd32(h + 4);
    std::vector<uint8_t> text(l_text);
    if (l_text && !b->z.read(text.data(), l_text)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
    if (!b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
    const int n_ref = (int)rd32(h);
    for (int i = 0; i < n_ref; i++) {
        if (!b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
        const uint32_t l_name = rd32(h);
        std::vector<char> nm(l_name);
        if (!b->z.read(nm.data(), l_name) || !b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
        b->ref_names.push_back(std::string(nm.data()));
        b->ref_lens.push_back((int64_t)rd32(h));
    }
    std::string p1 = std::string(path) + ".bai", p2 = path;
    if (p2.size() > 4 && p2.substr(p2.size() - 4) == ".bam") p2 = p2.substr(0, p2.size() - 4) + ".bai";
    if (!load_bai(b, p1) && !load_bai(b, p2)) {
        io_err("no BAI index next to %s", path);
        fclose(b->z.f); delete b; return nullptr;
    }
    return b;
}

extern "C" void pvio_bam_close(pv_bam* b) {
    if (!b) return;
    if (b->z.f) fclose(b->z.f);
    delete b;
}
extern "C" int pvio_bam_nref(pv_bam* b) { return b ? (int)b->ref_names.size() : 0; }
extern "C" const char* pvio_bam_ref_name(pv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[i].c_str() : nullptr; }
extern "C" int64_t pvio_bam_ref_len(pv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_lens.size()) ? b->ref_lens[i] : -1; }

// SAMv1 5.3: bins overlapping [beg, end)
static void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& out) {
    out.clear();
    if (beg >= end) return;
    --end;
    out.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) out.push_back((uint32_t)k);
}

static const char NT16[] = "=ACMGRSVTWYHKDBN";

extern "C" int pvio_bam_get_reads(pv_bam* b, const char* contig, int64_t start, int64_t stop, int include_supplementary,
                                  int min_mapq, int min_baseq, pvio_reads* out) {
    (void)min_baseq;  // only feeds bad_indicies in the reference (:216-222), which the image builder never reads
    if (!b || !contig || !out) { io_err("null argument"); return -1; }
    memset(out, 0, sizeof(*out));
    b->pos.clear(); b->pos_end.clear(); b->flag.clear(); b->is_rev.clear(); b->mapq.clear(); b->hp.clear();
    b->bases.clear(); b->quals.clear(); b->cigar.clear(); b->names.clear();
    b->base_off.assign(1, 0); b->cigar_off.assign(1, 0); b->name_off.assign(1, 0);
    int tid = -1;
    for (size_t i = 0; i < b->ref_names.size(); i++)
        if (b->ref_names[i] == contig) { tid = (int)i; break; }
    if (tid < 0) { io_err("contig %s not in the BAM header", contig); return -1; }
    const int64_t qbeg = start < 0 ? 0 : start, qend = stop;  // sam_itr_queryi(idx, tid, start, stop): [start, stop)
    std::vector<Chunk> chunks;
    if (tid < (int)b->index.size() && qend > qbeg) {
        const RefIndex& ri = b->index[tid];
        uint64_t min_off = 0;
        const size_t li = (size_t)(qbeg >> 14);
        if (!ri.linear.empty()) min_off = ri.linear[std::min(li, ri.linear.size() - 1)];
        std::vector<uint32_t> bins;
        reg2bins(qbeg, qend, bins);
        for (uint32_t bin : bins) {
            auto it = ri.bins.find(bin);
            if (it == ri.bins.end()) continue;
            for (const Chunk& c : it->second)
                if (c.end > min_off) chunks.push_back(c);
        }
        std::sort(chunks.begin(), chunks.end(), [](const Chunk& x, const Chunk& y) { return x.beg < y.beg; });
        // merge overlapping / adjacent chunks
        std::vector<Chunk> m;
        for (const Chunk& c : chunks) {
            if (!m.empty() && c.beg <= m.back().end) m.back().end = std::max(m.back().end, c.end);
            else m.push_back(c);
        }
        chunks.swap(m);
    }
    std::vector<uint8_t> rec;
    bool done = false;
    for (size_t ci = 0; ci < chunks.size() && !done; ci++) {
        if (!b->z.seek(chunks[ci].beg)) { io_err("seek failed in BAM"); return -1; }
        while (b->z.tell() < chunks[ci].end) {
            uint8_t h4[4];
            if (!b->z.read(h4, 4)) break;
            const uint32_t bs = rd32(h4);
            rec.resize(bs);
            if (!b->z.read(rec.data(), bs)) { io_err("truncated BAM record"); return -1; }
            const uint8_t* r = rec.data();
            const int32_t rtid = (int32_t)rd32(r);
            const int64_t rpos = (int32_t)rd32(r + 4);
            const int l_name = r[8];
            const int rmapq = r[9];
            const int n_cig = rd16(r + 12);
            const int fl = rd16(r + 14);
            const int64_t l_seq = (int32_t)rd32(r + 16);
            if (rtid != tid) { if (rtid > tid) { done = true; break; } continue; }
            if (rpos >= qend) { done = true; break; }
            const uint8_t* name = r + 32;
            const uint8_t* cig = name + l_name;
            const uint8_t* seq = cig + 4 * (size_t)n_cig;
            const uint8_t* qual = seq + (l_seq + 1) / 2;
            const uint8_t* aux = qual + l_seq;
            const uint8_t* aux_end = r + bs;
            // reference end (bam_endpos)
            int64_t rlen = 0;
            for (int k = 0; k < n_cig; k++) {
                const uint32_t c = rd32(cig + 4 * k);
                const int op = c & 0xF;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
            }
            const int64_t rend = rpos + (rlen > 0 ? rlen : 1);
            if (!(rend > qbeg && qend > rpos)) continue;
            // flag / mapq filters (:137-150)
            if ((fl & 0x200) || (fl & 0x400) || (fl & 0x100) || (fl & 0x4)) continue;
            if (!include_supplementary && (fl & 0x800)) continue;
            if (rmapq < min_mapq) continue;
            // clip to [start, stop] and rebuild the CIGAR (:180-306)
            const size_t base0 = b->bases.size(), cig0 = b->cigar.size();
            int64_t pos_start = -1, pos_endv = -1, cur_pos = rpos, cur_idx = 0;
            for (int k = 0; k < n_cig; k++) {
                const uint32_t c = rd32(cig + 4 * k);
                const int op = c & 0xF;
                const int64_t len = c >> 4;
                if (cur_pos > stop) break;
                int64_t kept = 0;
                switch (op) {
                    case 0: case 7: case 8: {
                        int64_t i0 = 0;
                        if (cur_pos < start) {
                            i0 = std::min(start - cur_pos, len);
                            cur_idx += i0;
                            cur_pos += i0;
                        }
                        for (int64_t i = i0; i < len; i++) {
                            if (cur_pos <= stop) {
                                if (pos_start == -1) { pos_start = cur_pos; pos_endv = pos_start; }
                                if (cur_idx >= l_seq) { io_err("CIGAR longer than SEQ in a BAM record"); return -1; }
                                b->quals.push_back(qual[cur_idx]);
                                b->bases.push_back((uint8_t)NT16[(seq[cur_idx >> 1] >> ((~cur_idx & 1) << 2)) & 0xF]);
                                kept++;
                                pos_endv++;
                            } else break;
                            cur_idx++;
                            cur_pos++;
                        }
                        break;
                    }
                    case 4: case 1:
                        if (cur_pos >= start && cur_pos <= stop && pos_start != -1) {
                            for (int64_t i = 0; i < len; i++) {
                                if (cur_idx >= l_seq) { io_err("CIGAR longer than SEQ in a BAM record"); return -1; }
                                b->quals.push_back(qual[cur_idx]);
                                b->bases.push_back((uint8_t)NT16[(seq[cur_idx >> 1] >> ((~cur_idx & 1) << 2)) & 0xF]);
                                kept++;
                                cur_idx++;
                            }
                        } else {
                            cur_idx += len;
                        }
                        break;
                    case 3: case 2:
                        if (cur_pos >= start && cur_pos <= stop && pos_start != -1) {
                            for (int64_t i = 0; i < len; i++) {
                                if (cur_pos <= stop) { kept++; pos_endv++; } else break;
                                cur_pos++;
                            }
                        } else {
                            cur_pos += len;
                        }
                        break;
                    default:  // H: ignored; P, B and unknown codes fall out of the switch without state change
                        break;
                }
                if (kept > 0) b->cigar.push_back((uint32_t)((kept << 4) | (uint32_t)op));
            }
            if (b->bases.size() == base0) {  // nothing kept: the read is not returned (:432)
                b->cigar.resize(cig0);
                continue;
            }
            // HP aux tag (:313-428): integer types c C s S i I
            int32_t hp = 0;
            for (const uint8_t* s = aux; aux_end - s >= 4;) {
                const char t0 = (char)s[0], t1 = (char)s[1];
                const char ty = (char)s[2];
                s += 3;
                int sz = 0;
                bool ok = true;
                switch (ty) {
                    case 'A': case 'c': case 'C': sz = 1; break;
                    case 's': case 'S': sz = 2; break;
                    case 'i': case 'I': case 'f': sz = 4; break;
                    case 'Z': case 'H': while (s < aux_end && *s) s++; s++; sz = 0; break;
                    case 'B': {
                        if (aux_end - s < 5) { ok = false; break; }
                        const char st = (char)s[0];
                        const uint32_t ne = rd32(s + 1);
                        const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                        s += 5 + (size_t)ne * es;
                        break;
                    }
                    default: ok = false; break;
                }
                if (!ok || aux_end - s < sz) break;
                if (t0 == 'H' && t1 == 'P') {
                    switch (ty) {
                        case 'c': hp = (int8_t)s[0]; break;
                        case 'C': hp = s[0]; break;
                        case 's': hp = (int16_t)rd16(s); break;
                        case 'S': hp = rd16(s); break;
                        case 'i': case 'I': hp = (int32_t)rd32(s); break;
                        default: break;
                    }
                }
                s += sz;
            }
            b->pos.push_back(pos_start);
            b->pos_end.push_back(pos_endv);
            b->flag.push_back((uint16_t)fl);
            b->is_rev.push_back((fl & 0x10) ? 1 : 0);
            b->mapq.push_back((uint8_t)rmapq);
            b->hp.push_back(hp);
            b->base_off.push_back((int64_t)b->bases.size());
            b->cigar_off.push_back((int64_t)b->cigar.size());
            b->names.insert(b->names.end(), (const char*)name, (const char*)name + (l_name > 0 ? l_name - 1 : 0));
            b->name_off.push_back((int64_t)b->names.size());
        }
    }
    out->n_reads = (int64_t)b->pos.size();
    out->n_bases = (int64_t)b->bases.size();
    out->n_cigar = (int64_t)b->cigar.size();
    out->pos = b->pos.data(); out->pos_end = b->pos_end.data(); out->flag = b->flag.data();
    out->is_reverse = b->is_rev.data(); out->mapq = b->mapq.data(); out->hp_tag = b->hp.data();
    out->base_off = b->base_off.data(); out->bases = b->bases.data(); out->quals = b->quals.data();
    out->cigar_off = b->cigar_off.data(); out->cigar = b->cigar.data();
    out->name_off = b->name_off.data(); out->names = b->names.data();
    return 0;
}

// ---- FASTA + FAI -------------------------------------------------------------------------------------------------
struct FaiEntry { int64_t len, offset, linebases, linewidth; };
struct pv_fasta {
    FILE* f = nullptr;
    std::vector<std::string> names;
    std::map<std::string, FaiEntry> idx;
};

extern "C" pv_fasta* pvio_fasta_open(const char* path) {
    pv_fasta* fa = new pv_fasta();
    fa->f = fopen(path, "rb");
    if (!fa->f) { io_err("cannot open %s", path); delete fa; return nullptr; }
    const std::string fai = std::string(path) + ".fai";
    FILE* fi = fopen(fai.c_str(), "r");
    if (!fi) { io_err("FASTA index %s not found", fai.c_str()); fclose(fa->f); delete fa; return nullptr; }
    char line[4096];
    while (fgets(line, sizeof(line), fi)) {
        char name[2048];
        long long len, off, lb, lw;
        if (sscanf(line, "%2047s\t%lld\t%lld\t%lld\t%lld", name, &len, &off, &lb, &lw) == 5) {
            FaiEntry e; e.len = len; e.offset = off; e.linebases = lb; e.linewidth = lw;
            fa->idx[name] = e;
            fa->names.push_back(name);
        }
    }
    fclose(fi);
    return fa;
}
extern "C" void pvio_fasta_close(pv_fasta* fa) {
    if (!fa) return;
    if (fa->f) fclose(fa->f);
    delete fa;
}
extern "C" int pvio_fasta_nseq(pv_fasta* fa) { return fa ? (int)fa->names.size() : 0; }
extern "C" const char* pvio_fasta_name(pv_fasta* fa, int i) { return (fa && i >= 0 && i < (int)fa->names.size()) ? fa->names[i].c_str() : nullptr; }
extern "C" int64_t pvio_fasta_len(pv_fasta* fa, const char* contig) {
    if (!fa || !contig) return -1;
    auto it = fa->idx.find(contig);
    return it == fa->idx.end() ? -2 : it->second.len;
}
// get_reference_sequence(contig, start, stop) = faidx_fetch_seq(start, stop-1), upper-cased (fasta_handler.cpp:31-51):
// bases [start, stop-1], clamped to the sequence; returns the number of bases written (<= stop-start), -2 unknown contig
extern "C" int64_t pvio_fasta_fetch(pv_fasta* fa, const char* contig, int64_t start, int64_t stop, char* out) {
    if (!fa || !contig || !out) { io_err("null argument"); return -1; }
    auto it = fa->idx.find(contig);
    if (it == fa->idx.end()) { io_err("contig %s not in the FASTA index", contig); return -2; }
    const FaiEntry& e = it->second;
    int64_t beg = start < 0 ? 0 : start, end = stop - 1;  // inclusive
    if (end >= e.len) end = e.len - 1;
    if (beg > end) return 0;
    int64_t n = 0;
    int64_t p = beg;
    std::vector<char> buf;
    while (p <= end) {
        const int64_t line = p / e.linebases, col = p % e.linebases;
        const int64_t take = std::min(end - p + 1, e.linebases - col);
        if (fseeko(fa->f, e.offset + line * e.linewidth + col, SEEK_SET) != 0) { io_err("seek failed in FASTA"); return -1; }
        buf.resize(take);
        if (fread(buf.data(), 1, take, fa->f) != (size_t)take) { io_err("truncated FASTA"); return -1; }
        for (int64_t i = 0; i < take; i++) {
            char c = buf[i];
            if (c >= 'a' && c <= 'z') c -= 32;
            out[n++] = c;
        }
        p += take;
    }
    return n;
}
