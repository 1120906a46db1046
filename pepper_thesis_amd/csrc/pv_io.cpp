// pv_io.cpp — native BAM/BAI and FASTA/FAI readers for the image-builder input, plus BGZF writers
// (BAM + BAI for synthetic inputs, bgzip + tabix for the VCF outputs).
//
// Replaces, without htslib (not available offline; the reference fetches htslib 1.9 at configure time,
// pepper/modules/htslib.cmake:8-10):
//   BAM_handler::get_reads            pepper_variant/modules/cpp/bam_handler.cpp:115-451
//   FASTA_handler::get_reference_sequence / get_chromosome_sequence_length / names
//                                     pepper_variant/modules/cpp/fasta_handler.cpp:18-56
//   the per-interval read/reference fetch of AlignmentSummarizer.create_summary (AlignmentSummarizer.py:180-218):
//   pvio_fill_batch writes a whole batch of intervals straight into the flat SoA layout of pv_batch_in
//   (include/pepper_hip.h), so no per-read host objects exist between the BAM and the GPU.
// The reference's region clipping:
//   * records with QC-fail / duplicate / secondary / unmapped flags are dropped, supplementary unless asked
//     for, MAPQ < min_mapq dropped (:137-150);
//   * the CIGAR walk stops at the first op that starts beyond `stop` (:186-188);
//   * M/=/X: the part left of `start` is skipped, bases with pos <= stop are kept, the op is re-emitted with
//     the kept length and its ORIGINAL code (:191-239);
//   * I and S: kept (with their bases) only when start <= pos <= stop and a base has already been kept,
//     otherwise only the query index advances (:240-275);
//   * D and N: kept (length clipped at stop) under the same condition, else the position advances (:276-301);
//   * H: ignored; P/B: no state change; a read is returned only if it kept at least one base (:432-445).
// Reads with more than 65535 CIGAR operations carry a `<l_seq>S<rlen>N` placeholder and their real CIGAR in the
// CG:B,I aux tag (SAMv1 4.2.2); htslib resolves that inside bam_read1, so the reference never sees the placeholder:
// the real CIGAR is taken from the tag here as well.
// BGZF/BAM/BAI/FAI/TBI are implemented from the SAM/BAM/tabix format specifications (hts-specs).
// Every size taken from a file is validated before it is used as an offset.
// Parity note: htslib is absent, so this reader is pinned by this repository's own writer-based tests only.
#include <zlib.h>
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <map>
#include <string>
#include <vector>

#include "../../include/pepper_io.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
static inline void wr16(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 0xFF); v.push_back((x >> 8) & 0xFF); }
static inline void wr32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((x >> (8 * i)) & 0xFF); }
static inline void wr64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((x >> (8 * i)) & 0xFF); }
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- raw-deflate backend ----------------------------------------------------------------------------------------
// A BGZF block is an independent raw-deflate stream of <= 64 KiB: exactly libdeflate's whole-buffer use case (about twice
// zlib's inflate rate). libdeflate.so.0 is resolved at run time with dlopen (its four entry points declared here, no header
// needed); zlib stays the fallback and the two are byte-identical (tests/test_bamio.py). PEPPER_INFLATE=zlib forces zlib.
struct Deflate {
    void* h = nullptr;
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
};
static Deflate* libdeflate() {
    static Deflate d;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"libdeflate.so.0", "libdeflate.so", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0", "/opt/conda/lib/libdeflate.so.0"};
        for (const char* n : names)
            if ((d.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!d.h) return;
        d.alloc = (void* (*)())dlsym(d.h, "libdeflate_alloc_decompressor");
        d.decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(d.h, "libdeflate_deflate_decompress");
        d.release = (void (*)(void*))dlsym(d.h, "libdeflate_free_decompressor");
        d.crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(d.h, "libdeflate_crc32");
        if (!d.alloc || !d.decompress || !d.release || !d.crc) { dlclose(d.h); d.h = nullptr; }
    });
    return d.h ? &d : nullptr;
}
static std::atomic<int> g_inflate_choice{-1};   // -1: from the environment; 0 zlib; 1 libdeflate when present
static bool want_libdeflate() {
    int c = g_inflate_choice.load();
    if (c < 0) {
        const char* e = getenv("PEPPER_INFLATE");
        c = (e && !strcmp(e, "zlib")) ? 0 : 1;
        g_inflate_choice.store(c);
    }
    return c == 1 && libdeflate() != nullptr;
}
extern "C" const char* pvio_inflate_backend(void) { return want_libdeflate() ? "libdeflate" : "zlib"; }
extern "C" int pvio_set_inflate_backend(int use_libdeflate) {   // tests; returns 1 if libdeflate is now in use
    g_inflate_choice.store(use_libdeflate ? 1 : 0);
    return want_libdeflate() ? 1 : 0;
}

// ---- BGZF reader ------------------------------------------------------------------------------------------------
// The consuming thread reads the blocks of the file in order (raw bytes); they are inflated either by that thread itself, one
// block at a time with nothing read ahead (the default), or - after set_helpers(n) - by n helper threads of the handle working
// on a ring of blocks AHEAD of the consumer (what hts_set_threads does for htslib): the consumer then mostly parses records
// while the next blocks are being inflated, so ONE interval is read at several cores' rate. Short jobs use it: with one thread
// per interval all intervals of a wave complete at the same moment and the device has nothing to do until then.
// Nothing is read ahead past `ahead_limit` (the end of the index chunk being walked), so at most the ring's depth in blocks is
// inflated without being used when a query ends early.
struct Bgzf {
    FILE* f = nullptr;
    int64_t block_coffset = -1;  // compressed offset of the block held in `buf`
    int64_t next_coffset = 0;
    std::vector<uint8_t> buf;    // uncompressed block
    size_t pos = 0;
    double t_inflate = 0.0;      // consumer thread: seconds in load_block (file reads, its own inflates, waiting for helpers)
    double t_helpers = 0.0;      // helper threads: seconds spent inflating (summed; guarded by m)
    int64_t bytes_inflated = 0;
    bool failed = false;         // a block could not be read for a reason OTHER than a clean end of file (message set)
    int64_t fpos = -1;           // file position (-1 unknown): consecutive blocks need no seek, and an fseeko would throw the
                                 // stdio buffer away every 20-64 KB
    std::vector<char> iobuf;     // 1 MB stdio buffer (set on the first block)
    void* ld = nullptr;          // libdeflate decompressor of the consuming thread
    int64_t ahead_limit = INT64_MAX;   // blocks STARTING beyond this compressed offset are not read ahead

    struct Slot {
        int64_t coffset = -1, next = -1;
        int clen = 0;
        uint32_t isize = 0;
        std::vector<uint8_t> cbuf, buf;
        int state = 0;           // 0 free, 1 raw bytes loaded, 2 being inflated, 3 inflated, 4 failed (msg)
        char msg[200];
    };
    std::vector<Slot> ring;      // slots head .. head + count - 1 (mod size) hold consecutive blocks of the file
    size_t head = 0, count = 0;
    int64_t file_next = 0;       // compressed offset of the block after the newest ring slot
    bool ahead_eof = false;      // the file ended (or a raw read failed) at file_next
    std::mutex m;                // guards slot states, head / count as seen by helpers, t_helpers, stop
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> helpers;
    bool stop = false;
    int sleepers = 0;            // helpers waiting for work (guarded by m)

    ~Bgzf() {
        stop_helpers();
        if (ld) libdeflate()->release(ld);
    }
    bool fail() { failed = true; return false; }

    // raw bytes of a slot -> its uncompressed bytes, CRC32 checked against the gzip trailer; `dec` = the calling thread's
    // libdeflate decompressor (created on first use). On failure the message is left in the slot.
    static bool inflate_slot(Slot& s, void*& dec) {
        s.buf.resize(s.isize);
        uint32_t crc = 0;
        if (want_libdeflate()) {
            Deflate* L = libdeflate();
            if (!dec && !(dec = L->alloc())) { snprintf(s.msg, sizeof(s.msg), "libdeflate_alloc_decompressor failed"); return false; }
            if (s.isize) {
                size_t actual = 0;
                const int rc = L->decompress(dec, s.cbuf.data(), (size_t)s.clen, s.buf.data(), s.isize, &actual);
                if (rc != 0 || actual != s.isize) { snprintf(s.msg, sizeof(s.msg), "inflate failed (libdeflate %d) at offset %lld", rc, (long long)s.coffset); return false; }
            }
            crc = L->crc(0, s.buf.data(), s.isize);
        } else {
            if (s.isize) {
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { snprintf(s.msg, sizeof(s.msg), "inflateInit2 failed"); return false; }
                zs.next_in = s.cbuf.data(); zs.avail_in = (uInt)s.clen;
                zs.next_out = s.buf.data(); zs.avail_out = s.isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END) { snprintf(s.msg, sizeof(s.msg), "inflate failed (%d) at offset %lld", rc, (long long)s.coffset); return false; }
            }
            crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), s.buf.data(), (uInt)s.isize);
        }
        if (crc != rd32(&s.cbuf[s.clen])) {   // cheap next to the inflate
            snprintf(s.msg, sizeof(s.msg), "BGZF block at offset %lld: CRC32 mismatch (corrupt file)", (long long)s.coffset);
            return false;
        }
        return true;
    }

    // header + payload of the block at the current file position (= coffset) -> slot. 1 ok, 0 clean end of file (nothing left at
    // a block boundary), -1 error (message set: a partial header or payload = truncated file)
    int read_raw(int64_t coffset, Slot& s) {
        uint8_t h[18];
        const size_t got = fread(h, 1, 18, f);
        fpos = -1;
        if (got != 18) {
            if (got != 0) { io_err("truncated BGZF block header at offset %lld", (long long)coffset); return -1; }
            return 0;
        }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { io_err("not a BGZF block at offset %lld", (long long)coffset); return -1; }
        const int xlen = rd16(h + 10);
        if (xlen < 6) { io_err("BGZF block without BC field"); return -1; }
        // find the BC subfield (it is the first one in practice; scan to be safe, never past the extra field)
        uint8_t extra_small[64];
        std::vector<uint8_t> extra_big;
        uint8_t* extra = extra_small;
        if (xlen > (int)sizeof(extra_small)) { extra_big.resize(xlen); extra = extra_big.data(); }
        memcpy(extra, h + 12, 6);
        if (xlen > 6 && fread(extra + 6, 1, xlen - 6, f) != (size_t)(xlen - 6)) { io_err("truncated BGZF header"); return -1; }
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = rd16(&extra[i + 2]);
            if (i + 4 + slen > xlen) break;
            if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = rd16(&extra[i + 4]);
            i += 4 + slen;
        }
        if (bsize < 0) { io_err("BGZF block without BC field"); return -1; }
        const int clen = bsize + 1 - 12 - xlen - 8;
        if (clen < 0) { io_err("corrupt BGZF block size"); return -1; }
        s.cbuf.resize((size_t)clen + 8);
        if (fread(s.cbuf.data(), 1, (size_t)clen + 8, f) != (size_t)(clen + 8)) { io_err("truncated BGZF block"); return -1; }
        const uint32_t isize = rd32(&s.cbuf[clen + 4]);
        if (isize > 65536) { io_err("corrupt BGZF block (ISIZE %u > 64 KiB)", isize); return -1; }
        s.coffset = coffset; s.clen = clen; s.isize = isize;
        s.next = coffset + bsize + 1;
        fpos = s.next;
        return 1;
    }

    Slot& at(size_t i) { return ring[(head + i) % ring.size()]; }
    // raw blocks into the free slots of the ring (consumer thread). `must_one`: the block at file_next is wanted itself and is
    // read whatever the read-ahead limit says.
    void refill(bool must_one) {
        size_t added = 0;
        while (count < ring.size() && !ahead_eof) {
            if (!(must_one && count == 0) && file_next > ahead_limit) break;
            Slot& s = at(count);   // free: no helper looks at slots beyond `count`
            const int rc = read_raw(file_next, s);
            if (rc == 0) { ahead_eof = true; break; }
            std::lock_guard<std::mutex> lk(m);
            if (rc < 0) {          // reported when (if) the consumer gets there
                s.coffset = file_next; s.state = 4; ahead_eof = true;
                snprintf(s.msg, sizeof(s.msg), "%.*s", (int)sizeof(s.msg) - 1, g_ioerr);
            } else {
                s.state = 1; file_next = s.next;
            }
            count++;
            added++;
        }
        if (added && !helpers.empty()) {   // one wake-up per batch of blocks, and only if somebody sleeps: a futex call per block
                                           // costs the consumer more than the block's share of the inflate
            bool wake;
            { std::lock_guard<std::mutex> lk(m); wake = sleepers > 0; }
            if (wake) cv_work.notify_all();
        }
    }
    void pop_locked() { at(0).state = 0; head = (head + 1) % ring.size(); count--; }
    // forget everything buffered (helpers finish the blocks they are working on first)
    void drain() {
        std::unique_lock<std::mutex> lk(m);
        for (size_t i = 0; i < count; i++) if (at(i).state == 1) at(i).state = 0;
        for (size_t i = 0; i < count; i++) { while (at(i).state == 2) cv_done.wait(lk); at(i).state = 0; }
        head = 0; count = 0;
    }
    void helper_main() {
        void* dec = nullptr;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            Slot* job = nullptr;
            for (;;) {
                if (stop) break;
                for (size_t i = 0; i < count && !job; i++) if (at(i).state == 1) job = &at(i);   // the oldest waiting block
                if (job) break;
                sleepers++;
                cv_work.wait(lk);
                sleepers--;
            }
            if (stop) break;
            job->state = 2;
            lk.unlock();
            const double t0 = now_s();
            const bool ok = inflate_slot(*job, dec);
            const double dt = now_s() - t0;
            lk.lock();
            job->state = ok ? 3 : 4;
            t_helpers += dt;
            cv_done.notify_all();
        }
        lk.unlock();
        if (dec) libdeflate()->release(dec);
    }
    void stop_helpers() {
        if (helpers.empty()) return;
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        for (std::thread& t : helpers) t.join();
        helpers.clear();
        stop = false;
    }
    // n helper threads (0 = the consumer inflates every block itself and nothing is read ahead)
    void set_helpers(int n) {
        stop_helpers();
        drain();
        fpos = -1;
        n = std::max(0, std::min(n, 64));
        ring.clear();
        ring.resize(n ? (size_t)(4 * n + 4) : 1);
        head = count = 0;
        for (int i = 0; i < n; i++) helpers.emplace_back([this] { helper_main(); });
    }

    bool load_block(int64_t coffset) {
        const double t0 = now_s();
        const bool ok = load_block_(coffset);
        t_inflate += now_s() - t0;
        return ok;
    }
    bool load_block_(int64_t coffset) {
        if (ring.empty()) ring.resize(1);
        if (iobuf.empty()) { iobuf.resize(1 << 20); setvbuf(f, iobuf.data(), _IOFBF, iobuf.size()); fpos = -1; }
        if (count > 0 && at(0).coffset < coffset && coffset < file_next) {   // a forward skip inside what is buffered
            std::unique_lock<std::mutex> lk(m);
            while (count > 0 && at(0).coffset < coffset) {
                while (at(0).state == 2) cv_done.wait(lk);
                pop_locked();
            }
        }
        if (!(count > 0 && at(0).coffset == coffset)) {   // not the block the ring continues with: start again there
            drain();
            if (coffset != fpos && fseeko(f, coffset, SEEK_SET) != 0) { io_err("seek to BGZF block at %lld failed", (long long)coffset); return fail(); }
            fpos = coffset;
            file_next = coffset; ahead_eof = false;
        }
        if (count == 0) refill(true);   // (otherwise the ring is topped up in batches after a block has been taken, below)
        if (count == 0) {   // clean end of file at a block boundary
            buf.clear(); pos = 0; block_coffset = coffset; next_coffset = coffset;
            return false;
        }
        Slot& s = at(0);
        {
            std::unique_lock<std::mutex> lk(m);
            if (s.state == 1) {   // no helper has picked it up: the consumer inflates it
                s.state = 2;
                lk.unlock();
                const bool ok = inflate_slot(s, ld);
                lk.lock();
                s.state = ok ? 3 : 4;
            } else {
                while (s.state == 2) cv_done.wait(lk);
            }
            if (s.state == 4) {
                io_err("%s", s.msg);
                pop_locked();
                return fail();
            }
            buf.swap(s.buf);
            block_coffset = s.coffset; next_coffset = s.next; pos = 0;
            bytes_inflated += s.isize;
            pop_locked();
        }
        if (!helpers.empty() && count * 2 <= ring.size()) refill(false);   // top up in batches, half a ring at a time: the helpers
                                                                           // work on them while the consumer parses
        return true;
    }
    bool seek(uint64_t voffset) {
        const int64_t co = (int64_t)(voffset >> 16);
        if (co != block_coffset && !load_block(co)) return false;
        pos = voffset & 0xFFFF;
        return pos <= buf.size();
    }
    // virtual offset of the next byte. A block that has been consumed to its end (in particular a full 64 KiB block,
    // whose end offset 65536 does not fit the 16 offset bits) is reported as offset 0 of the following block, as
    // htslib's bgzf_tell does after bgzf_read moves on.
    uint64_t tell() const {
        if (block_coffset >= 0 && !buf.empty() && pos >= buf.size()) return (uint64_t)next_coffset << 16;
        return ((uint64_t)block_coffset << 16) | (uint64_t)(pos & 0xFFFF);
    }
    // read exactly n bytes; false at EOF / error
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos >= buf.size()) {
                if (!load_block(next_coffset)) return false;
                if (buf.empty()) continue;   // empty block (the EOF marker): the next load reports the end of the file
            }
            const size_t k = std::min(n, buf.size() - pos);
            memcpy(d, buf.data() + pos, k);
            d += k; pos += k; n -= k;
        }
        return true;
    }
};

// ---- BGZF writer ------------------------------------------------------------------------------------------------
static const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct BgzfWriter {
    FILE* f = nullptr;
    std::vector<uint8_t> cur;
    int64_t coff = 0;  // compressed bytes written so far
    int level = 6;
    static const size_t BLOCK = 0xFF00;

    uint64_t tell() const { return ((uint64_t)coff << 16) | (uint64_t)cur.size(); }
    bool flush_block() {
        if (cur.empty()) return true;
        std::vector<uint8_t> comp(compressBound((uLong)cur.size()) + 64);
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { io_err("deflateInit2 failed"); return false; }
        zs.next_in = cur.data(); zs.avail_in = (uInt)cur.size();
        zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || clen + 26 > 65536) { io_err("deflate failed (%d)", rc); return false; }
        std::vector<uint8_t> blk = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0};
        wr16(blk, (uint32_t)(clen + 25));
        blk.insert(blk.end(), comp.begin(), comp.begin() + clen);
        wr32(blk, (uint32_t)crc32(crc32(0L, Z_NULL, 0), cur.data(), (uInt)cur.size()));
        wr32(blk, (uint32_t)cur.size());
        if (fwrite(blk.data(), 1, blk.size(), f) != blk.size()) { io_err("short write"); return false; }
        coff += (int64_t)blk.size();
        cur.clear();
        return true;
    }
    bool write(const void* p, size_t n) {
        const uint8_t* s = (const uint8_t*)p;
        while (n) {
            const size_t k = std::min(n, BLOCK - cur.size());
            cur.insert(cur.end(), s, s + k);
            s += k; n -= k;
            if (cur.size() >= BLOCK && !flush_block()) return false;
        }
        return true;
    }
    bool close() {
        bool ok = flush_block();
        if (f) {
            ok = ok && fwrite(BGZF_EOF, 1, 28, f) == 28;
            ok = (fclose(f) == 0) && ok;
            f = nullptr;
        }
        return ok;
    }
};

// SAMv1 5.3
static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
// bins overlapping [beg, end)
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

// binning + linear index under construction (BAI and TBI share the layout)
struct Chunk { uint64_t beg, end; };
struct IndexBuilder {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::map<int64_t, uint64_t> linear;
    void add(int64_t beg, int64_t end, uint64_t vbeg, uint64_t vend) {
        std::vector<Chunk>& v = bins[(uint32_t)reg2bin(beg, end)];
        if (!v.empty() && vbeg <= v.back().end) v.back().end = std::max(v.back().end, vend);
        else v.push_back(Chunk{vbeg, vend});
        for (int64_t w = beg >> 14; w <= (end - 1) >> 14; w++) {
            auto it = linear.find(w);
            if (it == linear.end() || vbeg < it->second) linear[w] = vbeg;
        }
    }
    void serialize(std::vector<uint8_t>& out) const {
        wr32(out, (uint32_t)bins.size());
        for (const auto& kv : bins) {
            wr32(out, kv.first);
            wr32(out, (uint32_t)kv.second.size());
            for (const Chunk& c : kv.second) { wr64(out, c.beg); wr64(out, c.end); }
        }
        const int64_t n_intv = linear.empty() ? 0 : linear.rbegin()->first + 1;
        wr32(out, (uint32_t)n_intv);
        uint64_t last = 0;
        for (int64_t w = 0; w < n_intv; w++) {
            auto it = linear.find(w);
            if (it != linear.end()) last = it->second;
            wr64(out, last);
        }
    }
};

// ---- BAM + BAI -------------------------------------------------------------------------------------------------
struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;
};

// flat read storage shared by get_reads (one region) and fill_batch (many regions)
// Growable byte array whose resize() leaves new bytes uninitialised and grows through realloc(): the base / quality arrays
// (12 MB per 100 kb interval at 60x) are overwritten right after they grow. (std::vector spent more time in resize() than the
// record decode took: value-initialisation, or with a no-op construct() an element loop the compiler keeps, plus a copy per
// doubling; realloc() of a block this size is an mremap.)
struct ByteVec {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    ByteVec() {}
    ByteVec(const ByteVec&) = delete;
    ByteVec& operator=(const ByteVec&) = delete;
    ~ByteVec() { free(p); }
    size_t size() const { return n; }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    uint8_t& operator[](size_t i) { return p[i]; }
    void clear() { n = 0; }
    void reserve(size_t c) {
        if (c <= cap) return;
        size_t nc = cap ? cap : 4096;
        while (nc < c) nc *= 2;
        uint8_t* q = (uint8_t*)realloc(p, nc);
        if (!q) throw std::bad_alloc();
        p = q; cap = nc;
    }
    void resize(size_t m) { reserve(m); n = m; }
    void append(const uint8_t* src, size_t k) { const size_t at = n; resize(n + k); if (k) memcpy(p + at, src, k); }
};

struct ReadSink {
    std::vector<int64_t> pos, pos_end, base_off, cigar_off, name_off;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> is_rev, mapq;
    ByteVec bases, quals;
    std::vector<int32_t> hp;
    std::vector<uint32_t> cigar;
    std::vector<char> names;
    bool want_names = true;
    void clear() {
        pos.clear(); pos_end.clear(); flag.clear(); is_rev.clear(); mapq.clear(); hp.clear();
        bases.clear(); quals.clear(); cigar.clear(); names.clear();
        base_off.assign(1, 0); cigar_off.assign(1, 0); name_off.assign(1, 0);
    }
    size_t n_reads() const { return pos.size(); }
    // drop the reads after the first n (their bases / cigar ops are at the end of the flat arrays)
    void truncate(size_t n) {
        pos.resize(n); pos_end.resize(n); flag.resize(n); is_rev.resize(n); mapq.resize(n); hp.resize(n);
        base_off.resize(n + 1); cigar_off.resize(n + 1); name_off.resize(n + 1);
        bases.resize((size_t)base_off[n]); quals.resize((size_t)base_off[n]); cigar.resize((size_t)cigar_off[n]);
        names.resize((size_t)name_off[n]);
    }
};

struct pv_bam {
    Bgzf z;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::vector<RefIndex> index;
    ReadSink sink;  // last pvio_bam_get_reads result (owned here, pointers handed to the caller)
    double t_decode = 0.0;
};

static bool load_bai(pv_bam* b, const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n > 0 ? n : 0);
    if (n < 0 || fread(d.data(), 1, d.size(), f) != d.size()) { fclose(f); return false; }
    fclose(f);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) { io_err("%s is not a BAI index", path.c_str()); return false; }
    size_t p = 4;
    // every read is bounded against the file size: a truncated index is an error, never an out-of-bounds read
    auto need = [&](size_t k) { return p + k <= d.size(); };
    const uint32_t n_ref = rd32(&d[p]); p += 4;
    if (n_ref > (1u << 24)) { io_err("%s: implausible reference count", path.c_str()); return false; }
    b->index.assign(n_ref, RefIndex());
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!need(4)) { io_err("%s is truncated", path.c_str()); return false; }
        const uint32_t n_bin = rd32(&d[p]); p += 4;
        for (uint32_t k = 0; k < n_bin; k++) {
            if (!need(8)) { io_err("%s is truncated", path.c_str()); return false; }
            const uint32_t bin = rd32(&d[p]); p += 4;
            const uint32_t n_chunk = rd32(&d[p]); p += 4;
            if ((uint64_t)n_chunk * 16 > d.size() - p) { io_err("%s is truncated", path.c_str()); return false; }
            std::vector<Chunk>& v = b->index[r].bins[bin];
            for (uint32_t c = 0; c < n_chunk; c++) {
                Chunk ch; ch.beg = rd64(&d[p]); ch.end = rd64(&d[p + 8]); p += 16;
                v.push_back(ch);
            }
        }
        if (!need(4)) { io_err("%s is truncated", path.c_str()); return false; }
        const uint32_t n_intv = rd32(&d[p]); p += 4;
        if ((uint64_t)n_intv * 8 > d.size() - p) { io_err("%s is truncated", path.c_str()); return false; }
        b->index[r].linear.resize(n_intv);
        for (uint32_t k = 0; k < n_intv; k++) { b->index[r].linear[k] = rd64(&d[p]); p += 8; }
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
    std::vector<uint8_t> text;
    // the header text is read in bounded pieces: a corrupt l_text cannot trigger a huge allocation before EOF is seen
    for (uint32_t left = l_text; left;) {
        uint8_t tmp[4096];
        const uint32_t k = std::min<uint32_t>(left, sizeof(tmp));
        if (!b->z.read(tmp, k)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
        left -= k;
    }
    if (!b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
    const uint32_t n_ref = rd32(h);
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
        const uint32_t l_name = rd32(h);
        if (l_name == 0 || l_name > 65536) { io_err("corrupt BAM header (reference name length %u)", l_name); fclose(b->z.f); delete b; return nullptr; }
        std::vector<char> nm(l_name + 1, 0);
        if (!b->z.read(nm.data(), l_name) || !b->z.read(h, 4)) { io_err("truncated BAM header"); fclose(b->z.f); delete b; return nullptr; }
        b->ref_names.push_back(std::string(nm.data()));
        b->ref_lens.push_back((int64_t)rd32(h));
    }
    std::string p1 = std::string(path) + ".bai", p2 = path;
    if (p2.size() > 4 && p2.substr(p2.size() - 4) == ".bam") p2 = p2.substr(0, p2.size() - 4) + ".bai";
    g_ioerr[0] = 0;
    if (!load_bai(b, p1) && !load_bai(b, p2)) {
        if (!g_ioerr[0]) io_err("no BAI index next to %s", path);
        fclose(b->z.f); delete b; return nullptr;
    }
    return b;
}

extern "C" void pvio_bam_close(pv_bam* b) {
    if (!b) return;
    if (b->z.f) fclose(b->z.f);
    delete b;
}
// n helper threads inflate BGZF blocks ahead of the reading thread (0, the default: none, nothing read ahead)
extern "C" int pvio_bam_set_threads(pv_bam* b, int n_helpers) {
    if (!b) { io_err("null argument"); return -1; }
    b->z.set_helpers(n_helpers);
    return (int)b->z.helpers.size();
}
extern "C" int pvio_bam_nref(pv_bam* b) { return b ? (int)b->ref_names.size() : 0; }
extern "C" const char* pvio_bam_ref_name(pv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[i].c_str() : nullptr; }
extern "C" int64_t pvio_bam_ref_len(pv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_lens.size()) ? b->ref_lens[i] : -1; }

static const char NT16[] = "=ACMGRSVTWYHKDBN";

// one alignment record, validated: every pointer below lies inside [r, r + bs)
struct RecView {
    int32_t tid;
    int64_t pos;
    int l_name, mapq, flag;
    int64_t l_seq;
    const uint8_t *name, *seq, *qual, *aux, *aux_end;
    const uint32_t* cigar;  // host-endian ops (from the record, or from the CG tag)
    int64_t n_cigar;
};

// advance over one aux field starting at s (tag[2] type[1] value); returns the value pointer and its size through
// the arguments, nullptr when the field does not fit
static const uint8_t* aux_next(const uint8_t* s, const uint8_t* end, char& t0, char& t1, char& ty, const uint8_t*& val, int64_t& vbytes) {
    if (end - s < 3) return nullptr;
    t0 = (char)s[0]; t1 = (char)s[1]; ty = (char)s[2];
    s += 3;
    val = s;
    switch (ty) {
        case 'A': case 'c': case 'C': vbytes = 1; break;
        case 's': case 'S': vbytes = 2; break;
        case 'i': case 'I': case 'f': vbytes = 4; break;
        case 'Z': case 'H': {
            const uint8_t* e = s;
            while (e < end && *e) e++;
            if (e >= end) return nullptr;
            vbytes = (e - s) + 1;
            break;
        }
        case 'B': {
            if (end - s < 5) return nullptr;
            const char st = (char)s[0];
            const uint32_t ne = rd32(s + 1);
            const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
            if (!es) return nullptr;
            vbytes = 5 + (int64_t)ne * es;
            break;
        }
        default: return nullptr;
    }
    if (end - s < vbytes) return nullptr;
    return s + vbytes;
}

// 0 ok, -1 corrupt (message set). cigbuf receives the decoded CIGAR words.
static int parse_record(const uint8_t* r, uint32_t bs, RecView& v, std::vector<uint32_t>& cigbuf) {
    if (bs < 32) { io_err("corrupt BAM record (block_size %u < 32)", bs); return -1; }
    v.tid = (int32_t)rd32(r);
    v.pos = (int32_t)rd32(r + 4);
    v.l_name = r[8];
    v.mapq = r[9];
    const int64_t n_cig = rd16(r + 12);
    v.flag = rd16(r + 14);
    v.l_seq = (int32_t)rd32(r + 16);
    if (v.l_seq < 0) { io_err("corrupt BAM record (negative l_seq)"); return -1; }
    const uint64_t need = 32ull + (uint64_t)v.l_name + 4ull * (uint64_t)n_cig + (uint64_t)((v.l_seq + 1) / 2) + (uint64_t)v.l_seq;
    if (need > bs) { io_err("corrupt BAM record (fields need %llu bytes, block_size %u)", (unsigned long long)need, bs); return -1; }
    v.name = r + 32;
    const uint8_t* cig = v.name + v.l_name;
    v.seq = cig + 4 * (size_t)n_cig;
    v.qual = v.seq + (v.l_seq + 1) / 2;
    v.aux = v.qual + v.l_seq;
    v.aux_end = r + bs;
    cigbuf.resize((size_t)n_cig);
    for (int64_t k = 0; k < n_cig; k++) cigbuf[k] = rd32(cig + 4 * k);
    // long-CIGAR placeholder: first op = <l_seq>S and a CG:B,I tag (htslib bam_tag2cigar)
    if (n_cig >= 1 && v.tid >= 0 && v.pos >= 0 && (cigbuf[0] & 0xF) == 4 && (int64_t)(cigbuf[0] >> 4) == v.l_seq) {
        for (const uint8_t* s = v.aux; s < v.aux_end;) {
            char t0, t1, ty; const uint8_t* val; int64_t vb;
            const uint8_t* nx = aux_next(s, v.aux_end, t0, t1, ty, val, vb);
            if (!nx) break;
            if (t0 == 'C' && t1 == 'G' && ty == 'B' && (char)val[0] == 'I') {
                const uint32_t ne = rd32(val + 1);
                cigbuf.resize(ne);
                for (uint32_t k = 0; k < ne; k++) cigbuf[k] = rd32(val + 5 + 4 * (size_t)k);
                break;
            }
            s = nx;
        }
    }
    v.cigar = cigbuf.data();
    v.n_cigar = (int64_t)cigbuf.size();
    return 0;
}

static int32_t aux_hp(const RecView& v) {
    // HP aux tag (:313-428): integer types c C s S i I
    int32_t hp = 0;
    for (const uint8_t* s = v.aux; s < v.aux_end;) {
        char t0, t1, ty; const uint8_t* val; int64_t vb;
        const uint8_t* nx = aux_next(s, v.aux_end, t0, t1, ty, val, vb);
        if (!nx) break;
        if (t0 == 'H' && t1 == 'P') {
            switch (ty) {
                case 'c': hp = (int8_t)val[0]; break;
                case 'C': hp = val[0]; break;
                case 's': hp = (int16_t)rd16(val); break;
                case 'S': hp = rd16(val); break;
                case 'i': case 'I': hp = (int32_t)rd32(val); break;
                default: break;
            }
        }
        s = nx;
    }
    return hp;
}

// 4-bit codes [q0, q0 + n) of a packed SEQ field -> letters
static const struct PairLut { uint16_t t[256]; PairLut() { for (int b = 0; b < 256; b++) t[b] = (uint16_t)((uint8_t)NT16[b >> 4] | ((uint8_t)NT16[b & 15] << 8)); } } g_pair_lut;
static void unpack_pairs(const uint8_t* seq, int64_t q, int64_t n, uint8_t* dst) {   // q even: two letters per packed byte
    int64_t i = 0;
    for (; i + 2 <= n; i += 2, q += 2) {
        const uint16_t two = g_pair_lut.t[seq[q >> 1]];
        memcpy(dst + i, &two, 2);
    }
    if (i < n) dst[i] = (uint8_t)NT16[seq[q >> 1] >> 4];
}
#if defined(__x86_64__)
__attribute__((target("ssse3"))) static void unpack_pairs_ssse3(const uint8_t* seq, int64_t q, int64_t n, uint8_t* dst) {
    const __m128i lut = _mm_loadu_si128((const __m128i*)NT16), m4 = _mm_set1_epi8(0x0F);
    int64_t i = 0;
    for (; i + 32 <= n; i += 32, q += 32) {   // 16 packed bytes -> 32 letters
        const __m128i p = _mm_loadu_si128((const __m128i*)(seq + (q >> 1)));
        const __m128i hi = _mm_shuffle_epi8(lut, _mm_and_si128(_mm_srli_epi16(p, 4), m4)), lo = _mm_shuffle_epi8(lut, _mm_and_si128(p, m4));
        _mm_storeu_si128((__m128i*)(dst + i), _mm_unpacklo_epi8(hi, lo));
        _mm_storeu_si128((__m128i*)(dst + i + 16), _mm_unpackhi_epi8(hi, lo));
    }
    unpack_pairs(seq, q, n - i, dst + i);
}
static const bool g_have_ssse3 = __builtin_cpu_supports("ssse3");
#endif
static void unpack_seq(const uint8_t* seq, int64_t q, int64_t n, uint8_t* dst) {
    if (n > 0 && (q & 1)) { *dst++ = (uint8_t)NT16[seq[q >> 1] & 0xF]; q++; n--; }
#if defined(__x86_64__)
    if (g_have_ssse3) { unpack_pairs_ssse3(seq, q, n, dst); return; }
#endif
    unpack_pairs(seq, q, n, dst);
}

// clip one record to [start, stop] and append it to the sink (:180-306). 1 = appended, 0 = nothing kept, -1 = corrupt.
// The reference appends the bases of every kept operation as it walks the CIGAR; the kept operations of a read cover ONE
// contiguous range of its SEQ (from the first aligned base inside the window every SEQ-consuming operation is kept until the
// walk leaves the window), so the walk only tracks that range and the bases / qualities are copied once at the end - an ONT
// read has thousands of operations of a few bases each.
static int clip_append(const RecView& v, int64_t start, int64_t stop, ReadSink& o) {
    const size_t cig0 = o.cigar.size();
    int64_t pos_start = -1, pos_endv = -1, cur_pos = v.pos, cur_idx = 0;
    int64_t q_first = 0, q_end = 0;   // kept SEQ range
    const int64_t l_seq = v.l_seq;
    for (int64_t k = 0; k < v.n_cigar; k++) {
        const uint32_t c = v.cigar[k];
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
                // bases with pos <= stop are kept: a closed form of the reference's per-base loop
                const int64_t n = std::max<int64_t>(0, std::min(len - i0, stop - cur_pos + 1));
                if (n > 0) {
                    if (cur_idx + n > l_seq) { io_err("CIGAR longer than SEQ in a BAM record"); o.cigar.resize(cig0); return -1; }
                    if (pos_start == -1) { pos_start = cur_pos; pos_endv = pos_start; q_first = cur_idx; }
                    kept = n;
                    pos_endv += n;
                    cur_idx += n;
                    cur_pos += n;
                    q_end = cur_idx;
                }
                break;
            }
            case 4: case 1:
                if (cur_pos >= start && cur_pos <= stop && pos_start != -1) {
                    if (cur_idx + len > l_seq) { io_err("CIGAR longer than SEQ in a BAM record"); o.cigar.resize(cig0); return -1; }
                    kept = len;
                    q_end = cur_idx + len;
                }
                cur_idx += len;
                break;
            case 3: case 2:
                if (cur_pos >= start && cur_pos <= stop && pos_start != -1) {
                    kept = std::min(len, stop - cur_pos + 1);
                    pos_endv += kept;
                    cur_pos += kept;
                } else {
                    cur_pos += len;
                }
                break;
            default:  // H: ignored; P, B and unknown codes fall out of the switch without state change
                break;
        }
        if (kept > 0) o.cigar.push_back((uint32_t)((kept << 4) | (uint32_t)op));
    }
    if (pos_start == -1) {  // nothing kept: the read is not returned (:432)
        o.cigar.resize(cig0);
        return 0;
    }
    {
        const int64_t n = q_end - q_first;
        const size_t at = o.bases.size();
        o.bases.resize(at + (size_t)n);
        o.quals.resize(at + (size_t)n);
        memcpy(&o.quals[at], v.qual + q_first, (size_t)n);
        unpack_seq(v.seq, q_first, n, &o.bases[at]);
    }
    o.pos.push_back(pos_start);
    o.pos_end.push_back(pos_endv);
    o.flag.push_back((uint16_t)v.flag);
    o.is_rev.push_back((v.flag & 0x10) ? 1 : 0);
    o.mapq.push_back((uint8_t)v.mapq);
    o.hp.push_back(aux_hp(v));
    o.base_off.push_back((int64_t)o.bases.size());
    o.cigar_off.push_back((int64_t)o.cigar.size());
    if (o.want_names) {
        int ln = v.l_name > 0 ? v.l_name - 1 : 0;
        o.names.insert(o.names.end(), (const char*)v.name, (const char*)v.name + ln);
    }
    o.name_off.push_back((int64_t)o.names.size());
    return 1;
}

// BAM_handler::get_reads body: every record of [start, stop) on `tid`, filtered and clipped, appended to the sink
static int query_region(pv_bam* b, int tid, int64_t start, int64_t stop, int include_supplementary, int min_mapq, ReadSink& sink) {
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
        std::vector<Chunk> m;  // merge overlapping / adjacent chunks
        for (const Chunk& c : chunks) {
            if (!m.empty() && c.beg <= m.back().end) m.back().end = std::max(m.back().end, c.end);
            else m.push_back(c);
        }
        chunks.swap(m);
    }
    std::vector<uint8_t> rec;
    std::vector<uint32_t> cigbuf;
    bool done = false;
    size_t run_end = 0;
    for (size_t ci = 0; ci < chunks.size() && !done; ci++) {
        if (ci >= run_end) {   // helpers do not inflate past the run of chunks that follow each other in the file: a writer that
                               // starts a new block rather than split a record (htslib's bgzf_flush_try) ends a chunk at the end of
                               // one block and begins the next at offset 0 of the following block - consecutive, but not mergeable
            run_end = ci + 1;
            while (run_end < chunks.size() && (int64_t)(chunks[run_end].beg >> 16) - (int64_t)(chunks[run_end - 1].end >> 16) <= (1 << 17)) run_end++;
            b->z.ahead_limit = (int64_t)(chunks[run_end - 1].end >> 16);
        }
        if (!b->z.seek(chunks[ci].beg)) { if (!b->z.failed) io_err("seek failed in BAM"); return -1; }   // (a failed block keeps its own message)
        while (b->z.tell() < chunks[ci].end) {
            uint8_t h4[4];
            if (!b->z.read(h4, 4)) {
                if (b->z.failed) return -1;   // a corrupt / truncated block in the middle of a chunk is an error, not "no more reads"
                break;                        // clean end of file
            }
            const uint32_t bs = rd32(h4);
            if (bs < 32 || bs > (1u << 30)) { io_err("corrupt BAM record (block_size %u)", bs); return -1; }
            rec.resize(bs);
            if (!b->z.read(rec.data(), bs)) { io_err("truncated BAM record"); return -1; }
            RecView v;
            if (parse_record(rec.data(), bs, v, cigbuf) != 0) return -1;
            if (v.tid != tid) { if (v.tid > tid) { done = true; break; } continue; }
            if (v.pos >= qend) { done = true; break; }
            int64_t rlen = 0;  // reference end (bam_endpos)
            for (int64_t k = 0; k < v.n_cigar; k++) {
                const int op = v.cigar[k] & 0xF;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += v.cigar[k] >> 4;
            }
            const int64_t rend = v.pos + (rlen > 0 ? rlen : 1);
            if (!(rend > qbeg && qend > v.pos)) continue;
            // flag / mapq filters (:137-150)
            if ((v.flag & 0x200) || (v.flag & 0x400) || (v.flag & 0x100) || (v.flag & 0x4)) continue;
            if (!include_supplementary && (v.flag & 0x800)) continue;
            if (v.mapq < min_mapq) continue;
            if (clip_append(v, start, stop, sink) < 0) return -1;
        }
    }
    return 0;
}

static int find_tid(pv_bam* b, const char* contig) {
    for (size_t i = 0; i < b->ref_names.size(); i++)
        if (b->ref_names[i] == contig) return (int)i;
    return -1;
}

extern "C" int pvio_bam_get_reads(pv_bam* b, const char* contig, int64_t start, int64_t stop, int include_supplementary,
                                  int min_mapq, int min_baseq, pvio_reads* out) {
    (void)min_baseq;  // only feeds bad_indicies in the reference (:216-222), which the image builder never reads
    if (!b || !contig || !out) { io_err("null argument"); return -1; }
    memset(out, 0, sizeof(*out));
    ReadSink& s = b->sink;
    s.want_names = true;
    s.clear();
    const int tid = find_tid(b, contig);
    if (tid < 0) { io_err("contig %s not in the BAM header", contig); return -1; }
    if (query_region(b, tid, start, stop, include_supplementary, min_mapq, s) != 0) return -1;
    out->n_reads = (int64_t)s.pos.size();
    out->n_bases = (int64_t)s.bases.size();
    out->n_cigar = (int64_t)s.cigar.size();
    out->pos = s.pos.data(); out->pos_end = s.pos_end.data(); out->flag = s.flag.data();
    out->is_reverse = s.is_rev.data(); out->mapq = s.mapq.data(); out->hp_tag = s.hp.data();
    out->base_off = s.base_off.data(); out->bases = s.bases.data(); out->quals = s.quals.data();
    out->cigar_off = s.cigar_off.data(); out->cigar = s.cigar.data();
    out->name_off = s.name_off.data(); out->names = s.names.data();
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
        if (sscanf(line, "%2047s\t%lld\t%lld\t%lld\t%lld", name, &len, &off, &lb, &lw) == 5 && lb > 0 && lw >= lb && len >= 0 && off >= 0) {
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
    // one seek, then sequential reads of the span (line ends included) in pieces of up to 4 MB; the line ends are dropped while
    // copying: a seek per line would throw the stdio buffer away ~1800 times per 110 kb
    const int64_t off0 = e.offset + (beg / e.linebases) * e.linewidth + beg % e.linebases;
    const int64_t off1 = e.offset + (end / e.linebases) * e.linewidth + end % e.linebases;
    if (fseeko(fa->f, off0, SEEK_SET) != 0) { io_err("seek failed in FASTA"); return -1; }
    std::vector<char> buf((size_t)std::min<int64_t>(off1 - off0 + 1, 1 << 22));
    int64_t n = 0, col = beg % e.linebases;   // col in [0, linewidth): columns >= linebases are the line terminator
    for (int64_t left = off1 - off0 + 1; left > 0;) {
        const size_t k = (size_t)std::min<int64_t>(left, (int64_t)buf.size());
        if (fread(buf.data(), 1, k, fa->f) != k) { io_err("truncated FASTA"); return -1; }
        left -= (int64_t)k;
        for (size_t i = 0; i < k;) {
            if (col >= e.linebases) {   // inside the terminator
                const size_t skip = (size_t)std::min<int64_t>((int64_t)(k - i), e.linewidth - col);
                i += skip; col += (int64_t)skip;
                if (col == e.linewidth) col = 0;
                continue;
            }
            const size_t take = (size_t)std::min<int64_t>((int64_t)(k - i), e.linebases - col);
            for (size_t j = 0; j < take; j++) {
                char c = buf[i + j];
                if (c >= 'a' && c <= 'z') c -= 32;
                out[n++] = c;
            }
            i += take; col += (int64_t)take;
            if (col == e.linewidth) col = 0;
        }
    }
    return n;
}

// ---- whole batches of intervals -> pv_batch_in arrays -------------------------------------------------------------
// NumPy's legacy RandomState(seed).randint(0, high) stream: MT19937 (init_genrand) + masked rejection on 32-bit draws
// (numpy/random/src/distributions: random_bounded_uint64_fill with use_masked, rng <= 0xFFFFFFFF) — the generator
// AlignmentSummarizer.py:195-205 draws its reservoir indices from.
struct LegacyMT {
    uint32_t mt[624];
    int idx;
    explicit LegacyMT(uint32_t seed) {
        for (int i = 0; i < 624; i++) {
            mt[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    // randint(0, high): uniform on [0, high)
    uint64_t randint(uint64_t high) {
        const uint64_t rng = high - 1;
        if (rng == 0) return 0;
        uint64_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        while ((v = ((uint64_t)next() & mask)) > rng) {}
        return v;
    }
};

extern "C" int64_t pvio_reservoir_indices(int64_t n_reads, double downsample_rate, int64_t max_reads, uint32_t seed, int64_t* out) {
    int64_t limit = (int64_t)std::min((double)max_reads, downsample_rate * (double)n_reads);
    if (limit < 0) limit = 0;
    if (n_reads <= limit) {
        for (int64_t i = 0; i < n_reads; i++) out[i] = i;
        return n_reads;
    }
    LegacyMT rng(seed);
    for (int64_t i = 0; i < limit; i++) out[i] = i;
    for (int64_t i = limit; i < n_reads; i++) {
        // `if len(sample) < total_allowed_reads: append else: j = random.randint(0, i + 1)`: with limit == 0 every read
        // draws and none is kept
        const int64_t j = (int64_t)rng.randint((uint64_t)i + 1);
        if (j < limit) out[j] = i;
    }
    return limit;
}

struct pvio_batch_store {
    ReadSink sink;
    std::vector<int64_t> ref_start, ref_end, cand_start, cand_end, ref_off, read_off, interval_index, reads_seen;
    std::vector<uint8_t> ref;
    pvio_batch view;
};

extern "C" void pvio_batch_free(pvio_batch* bt) {
    if (!bt) return;
    delete (pvio_batch_store*)bt->owner;
}

extern "C" int pvio_fill_batch(pv_bam* b, pv_fasta* fa, int n_intervals, const char* const* contigs, const int64_t* starts,
                               const int64_t* ends, int safe_bases, int include_supplementary, int min_mapq,
                               double downsample_rate, int64_t max_reads, uint32_t seed, pvio_batch** out) {
    if (!b || !fa || !out || (n_intervals > 0 && (!contigs || !starts || !ends))) { io_err("null argument"); return -1; }
    *out = nullptr;
    const double t_begin = now_s();
    const double infl0 = b->z.t_inflate;
    double help0;
    { std::lock_guard<std::mutex> lk(b->z.m); help0 = b->z.t_helpers; }
    const int64_t inflb0 = b->z.bytes_inflated;
    pvio_batch_store* st = new pvio_batch_store();
    ReadSink& s = st->sink;
    s.want_names = false;
    s.clear();
    st->ref_off.push_back(0);
    st->read_off.push_back(0);
    std::vector<int64_t> keep;
    std::vector<char> refbuf;
    int64_t max_len = 0;
    for (int iv = 0; iv < n_intervals; iv++) {
        const int tid = find_tid(b, contigs[iv]);
        if (tid < 0) { io_err("contig %s not in the BAM header", contigs[iv]); delete st; return -1; }
        // AlignmentSummarizer.py:181-189: reads for [start - safe, end + safe]
        const int64_t rs = std::max<int64_t>(0, starts[iv] - safe_bases), re = ends[iv] + safe_bases;
        const size_t r0 = s.n_reads();
        if (query_region(b, tid, rs, re, include_supplementary, min_mapq, s) != 0) { delete st; return -1; }
        const int64_t n_seen = (int64_t)(s.n_reads() - r0);
        // reservoir down-sampling (:191-208). The common case keeps everything in place.
        int64_t limit = (int64_t)std::min((double)max_reads, downsample_rate * (double)n_seen);
        if (limit < 0) limit = 0;
        if (n_seen > limit) {
            keep.resize((size_t)n_seen);
            const int64_t nk = pvio_reservoir_indices(n_seen, downsample_rate, max_reads, seed, keep.data());
            // rebuild the tail of the sink in reservoir order
            ReadSink t;
            t.want_names = false;
            t.clear();
            for (int64_t q = 0; q < nk; q++) {
                const size_t r = r0 + (size_t)keep[q];
                t.pos.push_back(s.pos[r]); t.pos_end.push_back(s.pos_end[r]); t.flag.push_back(s.flag[r]);
                t.is_rev.push_back(s.is_rev[r]); t.mapq.push_back(s.mapq[r]); t.hp.push_back(s.hp[r]);
                t.bases.append(s.bases.data() + s.base_off[r], (size_t)(s.base_off[r + 1] - s.base_off[r]));
                t.quals.append(s.quals.data() + s.base_off[r], (size_t)(s.base_off[r + 1] - s.base_off[r]));
                t.cigar.insert(t.cigar.end(), s.cigar.begin() + s.cigar_off[r], s.cigar.begin() + s.cigar_off[r + 1]);
                t.base_off.push_back((int64_t)t.bases.size());
                t.cigar_off.push_back((int64_t)t.cigar.size());
                t.name_off.push_back(0);
            }
            s.truncate(r0);
            const int64_t b0 = s.base_off.back(), c0 = s.cigar_off.back();
            for (size_t q = 0; q < t.n_reads(); q++) {
                s.pos.push_back(t.pos[q]); s.pos_end.push_back(t.pos_end[q]); s.flag.push_back(t.flag[q]);
                s.is_rev.push_back(t.is_rev[q]); s.mapq.push_back(t.mapq[q]); s.hp.push_back(t.hp[q]);
                s.base_off.push_back(b0 + t.base_off[q + 1]);
                s.cigar_off.push_back(c0 + t.cigar_off[q + 1]);
                s.name_off.push_back(0);
            }
            s.bases.append(t.bases.data(), t.bases.size());
            s.quals.append(t.quals.data(), t.quals.size());
            s.cigar.insert(s.cigar.end(), t.cigar.begin(), t.cigar.end());
        }
        if (s.n_reads() == r0) continue;  // "no group when no reads" (:212-213)
        // ref_seq must contain the region_end position (:216-218); the FASTA clamps at the contig end
        refbuf.resize((size_t)(re + 1 - rs));
        const int64_t got = pvio_fasta_fetch(fa, contigs[iv], rs, re + 1, refbuf.data());
        if (got < 0) { delete st; return -1; }
        if (got == 0) { s.truncate(r0); continue; }
        const int64_t re_c = rs + got - 1;
        st->ref.insert(st->ref.end(), refbuf.begin(), refbuf.begin() + got);
        st->ref_start.push_back(rs);
        st->ref_end.push_back(re_c);
        st->cand_start.push_back(starts[iv]);
        st->cand_end.push_back(std::min(ends[iv], re_c));
        st->ref_off.push_back((int64_t)st->ref.size());
        st->read_off.push_back((int64_t)s.n_reads());
        st->interval_index.push_back(iv);
        st->reads_seen.push_back(n_seen);
        max_len = std::max(max_len, got);
    }
    pvio_batch& v = st->view;
    memset(&v, 0, sizeof(v));
    v.owner = st;
    v.n_regions = (int32_t)st->ref_start.size();
    v.n_reads = (int64_t)s.n_reads();
    v.n_bases = (int64_t)s.bases.size();
    v.n_cigar = (int64_t)s.cigar.size();
    v.n_ref_bytes = (int64_t)st->ref.size();
    v.max_region_len = max_len;
    v.ref_start = st->ref_start.data(); v.ref_end = st->ref_end.data();
    v.cand_start = st->cand_start.data(); v.cand_end = st->cand_end.data();
    v.ref_off = st->ref_off.data(); v.ref = st->ref.data();
    v.read_off = st->read_off.data(); v.read_pos = s.pos.data();
    v.read_flags = s.is_rev.data(); v.read_mapq = s.mapq.data();
    v.base_off = s.base_off.data(); v.bases = s.bases.data(); v.quals = s.quals.data();
    v.cigar_off = s.cigar_off.data(); v.cigar = s.cigar.data();
    v.interval_index = st->interval_index.data(); v.reads_seen = st->reads_seen.data();
    v.t_inflate = b->z.t_inflate - infl0;
    { std::lock_guard<std::mutex> lk(b->z.m); v.t_helpers = b->z.t_helpers - help0; }
    v.t_total = now_s() - t_begin;
    v.bytes_inflated = b->z.bytes_inflated - inflb0;
    v.read_hp = s.hp.data();
    *out = &st->view;
    return 0;
}

// ---- writers ---------------------------------------------------------------------------------------------------------
// A coordinate-sorted BAM + BAI from flat arrays (synthetic inputs of the file-path benchmark and tests). Reads must be
// sorted by (tid, pos); flags: bit0 = reverse strand.
extern "C" int pvio_write_bam(const char* path, int n_ref, const char* const* ref_names, const int64_t* ref_lens, int64_t n_reads,
                              const int32_t* read_tid, const int64_t* read_pos, const uint8_t* read_flags, const uint8_t* read_mapq,
                              const int64_t* base_off, const uint8_t* bases, const uint8_t* quals, const int64_t* cigar_off,
                              const uint32_t* cigar, int level) {
    if (!path || n_ref <= 0 || !ref_names || !ref_lens) { io_err("null argument"); return -1; }
    BgzfWriter w;
    w.level = level;
    w.f = fopen(path, "wb");
    if (!w.f) { io_err("cannot create %s", path); return -1; }
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int i = 0; i < n_ref; i++) text += "@SQ\tSN:" + std::string(ref_names[i]) + "\tLN:" + std::to_string((long long)ref_lens[i]) + "\n";
    std::vector<uint8_t> hdr = {'B', 'A', 'M', 1};
    wr32(hdr, (uint32_t)text.size());
    hdr.insert(hdr.end(), text.begin(), text.end());
    wr32(hdr, (uint32_t)n_ref);
    for (int i = 0; i < n_ref; i++) {
        const size_t ln = strlen(ref_names[i]) + 1;
        wr32(hdr, (uint32_t)ln);
        hdr.insert(hdr.end(), ref_names[i], ref_names[i] + ln);
        wr32(hdr, (uint32_t)ref_lens[i]);
    }
    if (!w.write(hdr.data(), hdr.size()) || !w.flush_block()) { fclose(w.f); return -1; }
    uint8_t code[256];
    memset(code, 15, sizeof(code));
    for (int i = 0; i < 16; i++) code[(uint8_t)NT16[i]] = (uint8_t)i;
    std::vector<IndexBuilder> index((size_t)n_ref);
    std::vector<uint8_t> rec;
    for (int64_t r = 0; r < n_reads; r++) {
        const int64_t b0 = base_off[r], b1 = base_off[r + 1], c0 = cigar_off[r], c1 = cigar_off[r + 1];
        const int64_t l_seq = b1 - b0, n_cig = c1 - c0;
        if (n_cig > 65535) { io_err("pvio_write_bam: more than 65535 CIGAR operations"); fclose(w.f); return -1; }
        int64_t rlen = 0;
        for (int64_t k = c0; k < c1; k++) {
            const int op = cigar[k] & 0xF;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cigar[k] >> 4;
        }
        const int64_t end = read_pos[r] + std::max<int64_t>(rlen, 1);
        char name[32];
        const int l_name = snprintf(name, sizeof(name), "r%lld", (long long)r) + 1;
        rec.clear();
        wr32(rec, 0);  // block_size, patched below
        wr32(rec, (uint32_t)read_tid[r]);
        wr32(rec, (uint32_t)read_pos[r]);
        rec.push_back((uint8_t)l_name);
        rec.push_back(read_mapq[r]);
        wr16(rec, (uint32_t)reg2bin(read_pos[r], end));
        wr16(rec, (uint32_t)n_cig);
        wr16(rec, (read_flags[r] & 1) ? 16u : 0u);
        wr32(rec, (uint32_t)l_seq);
        wr32(rec, 0xFFFFFFFFu); wr32(rec, 0xFFFFFFFFu); wr32(rec, 0);
        rec.insert(rec.end(), name, name + l_name);
        for (int64_t k = c0; k < c1; k++) wr32(rec, cigar[k]);
        const size_t at = rec.size();
        rec.resize(at + (size_t)((l_seq + 1) / 2), 0);
        for (int64_t i = 0; i < l_seq; i++) rec[at + (size_t)(i >> 1)] |= (uint8_t)(code[bases[b0 + i]] << ((~i & 1) << 2));
        rec.insert(rec.end(), quals + b0, quals + b1);
        const uint32_t bs = (uint32_t)(rec.size() - 4);
        for (int i = 0; i < 4; i++) rec[i] = (bs >> (8 * i)) & 0xFF;
        if (w.cur.size() + rec.size() > BgzfWriter::BLOCK && !w.flush_block()) { fclose(w.f); return -1; }
        const uint64_t vbeg = w.tell();
        if (!w.write(rec.data(), rec.size())) { fclose(w.f); return -1; }
        const uint64_t vend = w.tell();
        if (read_tid[r] >= 0 && read_tid[r] < n_ref) index[(size_t)read_tid[r]].add(read_pos[r], end, vbeg, vend);
    }
    if (!w.close()) return -1;
    std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
    wr32(bai, (uint32_t)n_ref);
    for (const IndexBuilder& ix : index) ix.serialize(bai);
    const std::string bp = std::string(path) + ".bai";
    FILE* f = fopen(bp.c_str(), "wb");
    if (!f || fwrite(bai.data(), 1, bai.size(), f) != bai.size()) { io_err("cannot write %s", bp.c_str()); if (f) fclose(f); return -1; }
    fclose(f);
    return 0;
}

// bgzip a VCF text and write its tabix index (<path>.tbi), the two things pysam.VariantFile('w' on *.vcf.gz) +
// pysam.tabix_index do for the reference's outputs (VcfWriter.py:21-46). `text` = whole VCF (header lines start with '#',
// records sorted by contig then position); every record line starts a new index entry.
extern "C" int pvio_write_vcf_gz(const char* path, const char* text, int64_t n_bytes) {
    if (!path || (!text && n_bytes)) { io_err("null argument"); return -1; }
    BgzfWriter w;
    w.f = fopen(path, "wb");
    if (!w.f) { io_err("cannot create %s", path); return -1; }
    std::vector<std::string> names;
    std::vector<IndexBuilder> index;
    const char* p = text;
    const char* endp = text + n_bytes;
    while (p < endp) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(endp - p));
        const char* le = nl ? nl + 1 : endp;
        if (*p != '#') {
            // CHROM \t POS \t ID \t REF ...
            const char* t1 = (const char*)memchr(p, '\t', (size_t)(le - p));
            const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1)) : nullptr;
            const char* t3 = t2 ? (const char*)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1)) : nullptr;
            const char* t4 = t3 ? (const char*)memchr(t3 + 1, '\t', (size_t)(le - t3 - 1)) : nullptr;
            if (!t4) { io_err("malformed VCF record line"); fclose(w.f); return -1; }
            const std::string chrom(p, t1);
            const int64_t pos1 = strtoll(std::string(t1 + 1, t2).c_str(), nullptr, 10);
            const int64_t reflen = t4 - (t3 + 1);
            if (names.empty() || names.back() != chrom) { names.push_back(chrom); index.emplace_back(); }
            // a record must not straddle two BGZF blocks' worth of offset ambiguity: start a new block when it would not fit
            if (w.cur.size() + (size_t)(le - p) > BgzfWriter::BLOCK && !w.flush_block()) { fclose(w.f); return -1; }
            const uint64_t vbeg = w.tell();
            if (!w.write(p, (size_t)(le - p))) { fclose(w.f); return -1; }
            index.back().add(pos1 - 1, pos1 - 1 + std::max<int64_t>(reflen, 1), vbeg, w.tell());
        } else if (!w.write(p, (size_t)(le - p))) { fclose(w.f); return -1; }
        p = le;
    }
    if (!w.close()) return -1;
    // tabix index, VCF preset (tabix spec): format 2, col_seq 1, col_beg 2, col_end 0, meta '#', skip 0
    std::vector<uint8_t> tbi = {'T', 'B', 'I', 1};
    wr32(tbi, (uint32_t)names.size());
    wr32(tbi, 2); wr32(tbi, 1); wr32(tbi, 2); wr32(tbi, 0); wr32(tbi, (uint32_t)'#'); wr32(tbi, 0);
    size_t l_nm = 0;
    for (const std::string& n : names) l_nm += n.size() + 1;
    wr32(tbi, (uint32_t)l_nm);
    for (const std::string& n : names) { tbi.insert(tbi.end(), n.begin(), n.end()); tbi.push_back(0); }
    for (const IndexBuilder& ix : index) ix.serialize(tbi);
    BgzfWriter wi;
    const std::string ip = std::string(path) + ".tbi";
    wi.f = fopen(ip.c_str(), "wb");
    if (!wi.f) { io_err("cannot create %s", ip.c_str()); return -1; }
    if (!wi.write(tbi.data(), tbi.size()) || !wi.close()) return -1;
    return 0;
}

// inflate a whole BGZF file into memory (tests: read back a .vcf.gz / .tbi written above); returns bytes written or -1
extern "C" int64_t pvio_bgzf_read_all(const char* path, char* out, int64_t capacity) {
    Bgzf z;
    z.f = fopen(path, "rb");
    if (!z.f) { io_err("cannot open %s", path); return -1; }
    int64_t n = 0;
    int64_t co = 0;
    while (true) {
        if (!z.load_block(co)) break;
        if (z.buf.empty() && feof(z.f)) break;
        if (out) {
            if (n + (int64_t)z.buf.size() > capacity) { fclose(z.f); io_err("buffer too small"); return -1; }
            memcpy(out + n, z.buf.data(), z.buf.size());
        }
        n += (int64_t)z.buf.size();
        if (z.next_coffset == co) break;
        co = z.next_coffset;
    }
    fclose(z.f);
    return n;
}
