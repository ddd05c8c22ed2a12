// bamio.cpp — liblgmi_io.so (include/lgmi_io.h): streaming BGZF/BAM reader with BAI random access, pysam-default
// pile-up, and a BAI writer.  Written from the SAM/BAM specification (v1: 4.1 BGZF, 4.2 BAM records, 5.2 BAI, 5.3
// reg2bin / reg2bins); zlib does the raw inflate.  Replaces the pysam.AlignmentFile of
// src/giremi/script/giremi.py:21-24 for what the MI path calls on it (footprint.py:6-28, mismatch.py:69-190).
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lgmi_io.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ---------------------------------------------------------------- BGZF (spec 4.1): a series of gzip members, each with a
// 'BC' extra subfield holding its total size - 1; a virtual offset is (file offset of the block << 16) | offset inside it
// One BGZF block's deflate stream -> `out` (isize bytes) and its CRC-32.  libdeflate (whole-buffer decompressor, SIMD
// CRC) when the system has the library — opened with dlopen, its four entry points declared here: the image ships
// libdeflate.so.0 without a header — else zlib: an 8,000-gene run inflates the alignment file twice (footprint scan,
// site extraction), 0.9 of the 3 ms a footprint's record fetch took were zlib's inflate and crc32.
// LGIO_INFLATE=zlib forces the zlib path (tests run both).
struct Deflater {
    void* lib = nullptr;
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    void (*release)(void*) = nullptr;
    Deflater() {
        const char* e = getenv("LGIO_INFLATE");
        if (e && !strcmp(e, "zlib")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        *(void**)(&alloc) = dlsym(h, "libdeflate_alloc_decompressor");
        *(void**)(&decompress) = dlsym(h, "libdeflate_deflate_decompress");
        *(void**)(&crc) = dlsym(h, "libdeflate_crc32");
        *(void**)(&release) = dlsym(h, "libdeflate_free_decompressor");
        if (alloc && decompress && crc && release) lib = h; else dlclose(h);
    }
};
const Deflater& deflater() { static const Deflater d; return d; }
// per-thread decompressor object (libdeflate's is not shareable between threads at once)
struct DeflateCtx {
    void* d = nullptr;
    ~DeflateCtx() { if (d) deflater().release(d); }
    void* get() { if (!d && deflater().lib) d = deflater().alloc(); return d; }
};
// 0 ok, 1 inflate failed, 2 CRC mismatch
int inflate_raw(const uint8_t* comp, size_t clen, uint8_t* out, size_t isize, uint32_t want_crc) {
    static thread_local DeflateCtx ctx;
    if (void* d = ctx.get()) {
        size_t got = 0;
        if (deflater().decompress(d, comp, clen, out, isize, &got) != 0 || got != isize) return 1;
        return deflater().crc(0u, out, isize) == want_crc ? 0 : 2;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return 1;
    zs.next_in = const_cast<uint8_t*>(comp); zs.avail_in = (uInt)clen;
    zs.next_out = out; zs.avail_out = (uInt)isize;
    const int zr = inflate(&zs, Z_FINISH);
    const size_t total = zs.total_out;
    inflateEnd(&zs);
    if (zr != Z_STREAM_END || total != isize) return 1;
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)isize) == want_crc ? 0 : 2;
}

struct Bgzf {
    FILE* f = nullptr;
    uint64_t block_addr = ~0ull;        // file offset of the block in `data`
    uint64_t next_addr = 0;             // file offset of the block after it
    std::vector<uint8_t> data;          // its uncompressed bytes
    size_t upos = 0;
    uint64_t bytes_read = 0;
    std::vector<uint8_t> comp;
    // The last few blocks, inflated (round 5).  A BAI region query starts at the first read of the 16-kb window its start
    // lies in, and the footprints of a run are queried in position order: each query re-read — and re-inflated — the
    // blocks of the one or two footprints before it (10 of the 15 blocks a query of the 8,000-gene run touched).
    struct Kept { uint64_t addr = ~0ull, next = 0; std::vector<uint8_t> bytes; };
    std::vector<Kept> kept = std::vector<Kept>(24);
    size_t kept_next = 0;

    ~Bgzf() { if (f) fclose(f); }

    // 0 ok, 1 clean end of file, negative error
    int load(uint64_t addr) {
        if (addr == block_addr) return 0;
        for (Kept& k : kept)
            if (k.addr == addr) {                     // (the kept one comes out first, then the current block takes a place on the shelf)
                std::vector<uint8_t> bytes;
                bytes.swap(k.bytes);
                const uint64_t nxt = k.next;
                k.addr = ~0ull;
                stash();
                data.swap(bytes); block_addr = addr; next_addr = nxt;
                upos = 0;
                return 0;
            }
        stash();
        if (fseeko(f, (off_t)addr, SEEK_SET) != 0) return fail(LGIO_E_IO, "seek to %llu failed", (unsigned long long)addr);
        uint8_t h[18];
        const size_t got = fread(h, 1, 18, f);
        if (got == 0) return 1;
        if (got != 18) return fail(LGIO_E_FORMAT, "truncated BGZF block header at %llu", (unsigned long long)addr);
        bytes_read += 18;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(LGIO_E_FORMAT, "not a BGZF block at %llu", (unsigned long long)addr);
        const uint32_t xlen = le16(h + 10);
        // the BC subfield is normally the first (and only) one; walk the extra field otherwise
        uint32_t bsize = 0;
        if (xlen == 6 && h[12] == 'B' && h[13] == 'C') {
            bsize = le16(h + 16);
        } else {
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
            if (xlen > 6) {
                if (fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) return fail(LGIO_E_FORMAT, "truncated BGZF extra field");
                bytes_read += xlen - 6;
            }
            bool found = false;
            for (uint32_t p = 0; p + 4 <= xlen;) {
                const uint32_t slen = le16(&extra[p + 2]);
                if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) { bsize = le16(&extra[p + 4]); found = true; break; }
                p += 4 + slen;
            }
            if (!found) return fail(LGIO_E_FORMAT, "gzip member without a BC subfield at %llu (plain gzip, not BGZF?)", (unsigned long long)addr);
            if (fseeko(f, (off_t)(addr + 12 + xlen), SEEK_SET) != 0) return fail(LGIO_E_IO, "seek failed");
        }
        const uint64_t total = (uint64_t)bsize + 1;
        if (total < 12ull + xlen + 8) return fail(LGIO_E_FORMAT, "bad BGZF block size at %llu", (unsigned long long)addr);
        const size_t clen = (size_t)(total - 12 - xlen - 8);
        comp.resize(clen + 8);
        if (fread(comp.data(), 1, clen + 8, f) != clen + 8) return fail(LGIO_E_FORMAT, "truncated BGZF block at %llu", (unsigned long long)addr);
        bytes_read += clen + 8;
        const uint32_t isize = le32(comp.data() + clen + 4);
        if (isize > 65536) return fail(LGIO_E_FORMAT, "BGZF block larger than 64 KiB at %llu", (unsigned long long)addr);
        data.resize(isize);
        if (isize) {
            const int zr = inflate_raw(comp.data(), clen, data.data(), isize, le32(comp.data() + clen));
            if (zr == 1) return fail(LGIO_E_FORMAT, "inflate failed at block %llu", (unsigned long long)addr);
            if (zr == 2) return fail(LGIO_E_FORMAT, "CRC mismatch in BGZF block %llu", (unsigned long long)addr);
        }
        block_addr = addr;
        next_addr = addr + total;
        upos = 0;
        return 0;
    }
    void stash() {                                    // keep the block that is being left
        if (block_addr == ~0ull || data.empty()) return;
        Kept& k = kept[kept_next];
        kept_next = (kept_next + 1) % kept.size();
        k.bytes.swap(data); k.addr = block_addr; k.next = next_addr;
        block_addr = ~0ull;
    }
    int seek(uint64_t voff) {
        const int rc = load(voff >> 16);
        if (rc) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "virtual offset beyond the end of the file");
        upos = (size_t)(voff & 0xFFFF);
        if (upos > data.size()) return fail(LGIO_E_FORMAT, "virtual offset beyond its block");
        return 0;
    }
    // the offset of the NEXT byte to be read: a position at the end of a block belongs to the following block
    uint64_t tell() {
        if (upos >= data.size() && block_addr != ~0ull) return next_addr << 16;
        return (block_addr << 16) | upos;
    }
    // 0 ok, 1 end of file before the first byte, negative error (including EOF in the middle)
    int read(void* dst, size_t n) {
        uint8_t* out = static_cast<uint8_t*>(dst);
        size_t done = 0;
        while (done < n) {
            if (block_addr == ~0ull || upos >= data.size()) {
                const int rc = load(block_addr == ~0ull ? 0 : next_addr);
                if (rc == 1) return done ? fail(LGIO_E_FORMAT, "file ends inside a record") : 1;
                if (rc) return rc;
                continue;               // (an empty block — the EOF marker — simply moves on)
            }
            const size_t take = std::min(n - done, data.size() - upos);
            memcpy(out + done, data.data() + upos, take);
            upos += take;
            done += take;
        }
        return 0;
    }
};

// ---------------------------------------------------------------- BAI (spec 5.2)
int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& out) {
    out.clear();
    --end;
    out.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) out.push_back((uint32_t)k);
}

struct Chunk { uint64_t beg, end; };
struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;       // 16 kb windows: smallest virtual offset of a read overlapping the window
};
const int64_t MAX_POS = (int64_t)1 << 29;   // what a BAI can address

struct Record {     // one alignment, decoded
    int32_t tid; int64_t pos, end; uint16_t flag; uint8_t mapq;
    uint32_t l_name, n_cigar, l_seq;
    std::vector<uint8_t> raw;           // the record without its block_size word
    const uint8_t* name() const { return raw.data() + 32; }
    const uint8_t* cigar() const { return name() + l_name; }
    const uint8_t* seq() const { return cigar() + 4ull * n_cigar; }
    const uint8_t* qual() const { return seq() + (l_seq + 1) / 2; }
    const uint8_t* aux() const { return qual() + l_seq; }
    size_t aux_len() const { return raw.size() - (size_t)(aux() - raw.data()); }
};

// 0 ok, 1 end of file, negative error
int read_record(Bgzf& z, Record& r) {
    uint8_t len4[4];
    int rc = z.read(len4, 4);
    if (rc) return rc;
    const uint32_t bs = le32(len4);
    if (bs < 32 || bs > (1u << 29)) return fail(LGIO_E_FORMAT, "implausible BAM record size %u", bs);
    r.raw.resize(bs);
    rc = z.read(r.raw.data(), bs);
    if (rc) return rc == 1 ? fail(LGIO_E_FORMAT, "file ends inside a record") : rc;
    const uint8_t* p = r.raw.data();
    r.tid = (int32_t)le32(p);
    r.pos = (int32_t)le32(p + 4);
    r.l_name = p[8];
    r.mapq = p[9];
    r.n_cigar = le16(p + 12);
    r.flag = le16(p + 14);
    r.l_seq = le32(p + 16);
    const uint64_t need = 32ull + r.l_name + 4ull * r.n_cigar + (r.l_seq + 1ull) / 2 + r.l_seq;
    if (need > bs || r.l_name == 0) return fail(LGIO_E_FORMAT, "BAM record fields exceed the record");
    int64_t span = 0;
    const uint8_t* c = r.cigar();
    for (uint32_t k = 0; k < r.n_cigar; ++k) {
        const uint32_t v = le32(c + 4 * k), op = v & 0xF;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;      // M D N = X consume the reference
    }
    r.end = r.pos + (span > 0 ? span : 1);       // htslib: a read without reference-consuming operations spans 1
    return 0;
}

}  // namespace

struct lgio_bam {
    Bgzf z;
    std::string path, header_text;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::vector<RefIndex> index;
    uint64_t first_record_voff = 0;
    bool index_from_file = false;
};

namespace {

int read_header(lgio_bam* b) {
    uint8_t magic[4], w[4];
    int rc = b->z.read(magic, 4);
    if (rc) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "empty file");
    if (memcmp(magic, "BAM\1", 4) != 0) return fail(LGIO_E_FORMAT, "%s is not a BAM file", b->path.c_str());
    if ((rc = b->z.read(w, 4))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
    const uint32_t l_text = le32(w);
    if (l_text > (1u << 30)) return fail(LGIO_E_FORMAT, "header text of %u bytes refused", l_text);
    b->header_text.resize(l_text);
    if (l_text && (rc = b->z.read(&b->header_text[0], l_text))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
    while (!b->header_text.empty() && b->header_text.back() == '\0') b->header_text.pop_back();
    if ((rc = b->z.read(w, 4))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
    const uint32_t n_ref = le32(w);
    for (uint32_t k = 0; k < n_ref; ++k) {
        if ((rc = b->z.read(w, 4))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
        const uint32_t ln = le32(w);
        if (ln == 0 || ln > 65536) return fail(LGIO_E_FORMAT, "bad reference name length");
        std::string name(ln, '\0');
        if ((rc = b->z.read(&name[0], ln))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
        name.resize(ln - 1);
        if ((rc = b->z.read(w, 4))) return rc < 0 ? rc : fail(LGIO_E_FORMAT, "truncated header");
        b->ref_names.push_back(name);
        b->ref_lens.push_back((int32_t)le32(w));
    }
    b->first_record_voff = b->z.tell();
    return 0;
}

// one pass over the records: bins, chunks and the linear index of every reference
int scan_index(lgio_bam* b, std::vector<RefIndex>& idx) {
    idx.assign(b->ref_names.size(), RefIndex());
    int rc = b->z.seek(b->first_record_voff);
    if (rc) return rc;
    Record r;
    int32_t last_tid = -1; int64_t last_pos = -1;
    for (;;) {
        const uint64_t v0 = b->z.tell();
        rc = read_record(b->z, r);
        if (rc == 1) break;
        if (rc) return rc;
        const uint64_t v1 = b->z.tell();
        if (r.tid < 0) continue;                            // unplaced reads sit at the end
        if ((size_t)r.tid >= idx.size()) return fail(LGIO_E_FORMAT, "record refers to reference %d of %zu", r.tid, idx.size());
        if (r.tid < last_tid || (r.tid == last_tid && r.pos < last_pos))
            return fail(LGIO_E_FORMAT, "the BAM is not sorted by coordinate (reference %d, position %lld after %lld)", r.tid,
                        (long long)r.pos, (long long)last_pos);
        last_tid = r.tid; last_pos = r.pos;
        if (r.pos < 0 || r.end > MAX_POS) return fail(LGIO_E_FORMAT, "position beyond 2^29: a BAI cannot index it");
        RefIndex& ri = idx[r.tid];
        std::vector<Chunk>& ch = ri.bins[(uint32_t)reg2bin(r.pos, r.end)];
        if (!ch.empty() && ch.back().end == v0) ch.back().end = v1;   // consecutive records of a bin form one chunk
        else ch.push_back(Chunk{v0, v1});
        const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.end - 1) >> 14);
        if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (ri.linear[w] == 0) ri.linear[w] = v0;   // file order: the first is the smallest
    }
    // windows no read starts in or crosses inherit the previous window's offset (what samtools writes)
    for (RefIndex& ri : idx)
        for (size_t w = 1; w < ri.linear.size(); ++w) if (ri.linear[w] == 0) ri.linear[w] = ri.linear[w - 1];
    return 0;
}

int load_bai(const std::string& path, size_t n_ref, std::vector<RefIndex>& idx) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return 1;
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= buf.size(); };
    if (!need(8) || memcmp(buf.data(), "BAI\1", 4) != 0) return fail(LGIO_E_FORMAT, "%s is not a BAI file", path.c_str());
    p = 4;
    const uint32_t nr = le32(&buf[p]); p += 4;
    if (nr != n_ref) return fail(LGIO_E_FORMAT, "%s indexes %u references, the BAM has %zu", path.c_str(), nr, n_ref);
    idx.assign(n_ref, RefIndex());
    for (uint32_t r = 0; r < nr; ++r) {
        if (!need(4)) return fail(LGIO_E_FORMAT, "truncated BAI");
        const uint32_t n_bin = le32(&buf[p]); p += 4;
        for (uint32_t k = 0; k < n_bin; ++k) {
            if (!need(8)) return fail(LGIO_E_FORMAT, "truncated BAI");
            const uint32_t bin = le32(&buf[p]), n_chunk = le32(&buf[p + 4]); p += 8;
            if (!need(16ull * n_chunk)) return fail(LGIO_E_FORMAT, "truncated BAI");
            if (bin != 37450) {                             // 37450 is the metadata pseudo-bin
                std::vector<Chunk>& ch = idx[r].bins[bin];
                for (uint32_t c = 0; c < n_chunk; ++c) ch.push_back(Chunk{le64(&buf[p + 16 * c]), le64(&buf[p + 16 * c + 8])});
            }
            p += 16ull * n_chunk;
        }
        if (!need(4)) return fail(LGIO_E_FORMAT, "truncated BAI");
        const uint32_t n_intv = le32(&buf[p]); p += 4;
        if (!need(8ull * n_intv)) return fail(LGIO_E_FORMAT, "truncated BAI");
        idx[r].linear.resize(n_intv);
        for (uint32_t w = 0; w < n_intv; ++w) idx[r].linear[w] = le64(&buf[p + 8 * w]);
        p += 8ull * n_intv;
    }
    return 0;
}

int write_bai(const std::string& path, const std::vector<RefIndex>& idx) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(LGIO_E_IO, "cannot write %s", path.c_str());
    auto w32 = [&](uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; fwrite(b, 1, 4, f); };
    auto w64 = [&](uint64_t v) { w32((uint32_t)v); w32((uint32_t)(v >> 32)); };
    fwrite("BAI\1", 1, 4, f);
    w32((uint32_t)idx.size());
    for (const RefIndex& ri : idx) {
        w32((uint32_t)ri.bins.size());
        for (const auto& kv : ri.bins) {
            w32(kv.first);
            w32((uint32_t)kv.second.size());
            for (const Chunk& c : kv.second) { w64(c.beg); w64(c.end); }
        }
        w32((uint32_t)ri.linear.size());
        for (uint64_t v : ri.linear) w64(v);
    }
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? 0 : fail(LGIO_E_IO, "write error on %s", path.c_str());
}

// chunks that can hold reads overlapping [beg, end), merged and sorted
void query_chunks(const RefIndex& ri, int64_t beg, int64_t end, std::vector<Chunk>& out) {
    out.clear();
    std::vector<uint32_t> bins;
    reg2bins(beg, end, bins);
    const size_t w = (size_t)(beg >> 14);
    const uint64_t min_off = ri.linear.empty() ? 0 : (w < ri.linear.size() ? ri.linear[w] : ri.linear.back());
    for (uint32_t bn : bins) {
        auto it = ri.bins.find(bn);
        if (it == ri.bins.end()) continue;
        for (const Chunk& c : it->second) if (c.end > min_off) out.push_back(Chunk{std::max(c.beg, min_off), c.end});
    }
    std::sort(out.begin(), out.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
    size_t k = 0;
    for (size_t i = 0; i < out.size(); ++i) {
        if (k && out[i].beg <= out[k - 1].end) out[k - 1].end = std::max(out[k - 1].end, out[i].end);
        else out[k++] = out[i];
    }
    out.resize(k);
}

// the text of the first cs:Z auxiliary field (tag[2] type value ...), if any
bool find_cs(const Record& r, const uint8_t** text, size_t* text_len) {
    const uint8_t* a = r.aux();
    const size_t n = r.aux_len();
    size_t p = 0;
    while (p + 3 <= n) {
        const uint8_t t0 = a[p], t1 = a[p + 1], ty = a[p + 2];
        p += 3;
        size_t len = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
        else if (ty == 's' || ty == 'S') len = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
        else if (ty == 'Z' || ty == 'H') {
            while (p + len < n && a[p + len]) ++len;
            if (t0 == 'c' && t1 == 's' && ty == 'Z') { *text = a + p; *text_len = len; return true; }
            ++len;
        } else if (ty == 'B') {
            if (p + 5 > n) break;
            const uint8_t sub = a[p];
            const uint32_t cnt = le32(a + p + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * cnt;
        } else break;
        p += len;
    }
    return false;
}

struct ReadsOwner {
    std::vector<int32_t> tid; std::vector<int64_t> start, end; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, has_cs;
    std::vector<uint64_t> name_off{0}, cigar_off{0}, seq_off{0}, cs_off{0}, aux_off{0};
    std::vector<char> names, seq, cs; std::vector<uint32_t> cigar; std::vector<uint8_t> qual, aux;
    void add(const Record& r, uint32_t what) {
        tid.push_back(r.tid); start.push_back(r.pos); end.push_back(r.end); flag.push_back(r.flag); mapq.push_back(r.mapq);
        if (what & LGIO_NAMES) names.insert(names.end(), r.name(), r.name() + r.l_name - 1);
        name_off.push_back(names.size());
        if (what & LGIO_CIGAR) for (uint32_t k = 0; k < r.n_cigar; ++k) cigar.push_back(le32(r.cigar() + 4 * k));
        cigar_off.push_back(cigar.size());
        if (what & LGIO_SEQ) {
            static const char code[] = "=ACMGRSVTWYHKDBN";
            const uint8_t* s = r.seq();
            for (uint32_t k = 0; k < r.l_seq; ++k) seq.push_back(code[(s[k >> 1] >> ((~k & 1) << 2)) & 0xF]);
            qual.insert(qual.end(), r.qual(), r.qual() + r.l_seq);
        }
        seq_off.push_back(seq.size());
        uint8_t found = 0;
        if (what & LGIO_CS) {
            const uint8_t* text; size_t len;
            if (find_cs(r, &text, &len)) { cs.insert(cs.end(), text, text + len); found = 1; }
        }
        has_cs.push_back(found);
        cs_off.push_back(cs.size());
        if (what & LGIO_AUX) aux.insert(aux.end(), r.aux(), r.aux() + r.aux_len());
        aux_off.push_back(aux.size());
    }
    void view(lgio_reads* out) {
        out->n = tid.size();
        out->tid = tid.data(); out->start = start.data(); out->end = end.data(); out->flag = flag.data(); out->mapq = mapq.data();
        out->name_off = name_off.data(); out->names = names.data();
        out->cigar_off = cigar_off.data(); out->cigar = cigar.data();
        out->seq_off = seq_off.data(); out->seq = seq.data(); out->qual = qual.data();
        out->cs_off = cs_off.data(); out->cs = cs.data(); out->has_cs = has_cs.data();
        out->aux_off = aux_off.data(); out->aux = aux.data();
        out->owner_ = this;
    }
};

// calls fn(record) for every mapped read of `tid` overlapping [beg, end) (beg < 0: the whole reference)
template <class F>
int for_each_overlap(lgio_bam* b, int tid, int64_t beg, int64_t end, F fn) {
    if (tid < 0 || (size_t)tid >= b->ref_names.size()) return fail(LGIO_E_ARG, "reference id %d out of range", tid);
    if (beg < 0) { beg = 0; end = MAX_POS; }
    if (end > MAX_POS) end = MAX_POS;
    if (beg >= end) return 0;
    std::vector<Chunk> chunks;
    query_chunks(b->index[tid], beg, end, chunks);
    Record r;
    for (const Chunk& c : chunks) {
        int rc = b->z.seek(c.beg);
        if (rc) return rc;
        while (b->z.tell() < c.end) {
            rc = read_record(b->z, r);
            if (rc == 1) break;
            if (rc) return rc;
            if (r.tid != tid || r.pos >= end) { if (r.tid > tid || (r.tid == tid && r.pos >= end)) goto next_chunk; continue; }
            if (r.end <= beg || (r.flag & 4)) continue;
            fn(r);
        }
    next_chunk:;
    }
    return 0;
}

}  // namespace

extern "C" int lgio_abi_version(void) { return LGIO_ABI_VERSION; }
extern "C" const char* lgio_last_error(void) { return g_err.c_str(); }

static int bam_open_impl(const char* path, lgio_bam** out) {
    if (!path || !out) return fail(LGIO_E_ARG, "NULL argument");
    *out = nullptr;
    lgio_bam* b = new lgio_bam();
    struct Drop { lgio_bam* p; ~Drop() { delete p; } } drop{b};     // (also on an exception)
    b->path = path;
    b->z.f = fopen(path, "rb");
    if (!b->z.f) return fail(LGIO_E_IO, "cannot open %s", path);
    int rc = read_header(b);
    if (rc) return rc;
    std::string stem = b->path;
    if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".bam") == 0) stem.resize(stem.size() - 4);
    rc = load_bai(b->path + ".bai", b->ref_names.size(), b->index);
    if (rc == 1) rc = load_bai(stem + ".bai", b->ref_names.size(), b->index);
    if (rc < 0) return rc;
    b->index_from_file = (rc == 0);
    if (rc == 1 && (rc = scan_index(b, b->index))) return rc;
    drop.p = nullptr;
    *out = b;
    return LGIO_OK;
}

// No exception crosses the C boundary (ctypes cannot unwind one): allocation failures of the header text, the record
// buffer, the pile-up's span-sized arrays come back as LGIO_E_OOM, anything else as LGIO_E_FORMAT
template <class F> static int guarded(F f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(LGIO_E_OOM, "out of memory"); }
    catch (const std::length_error&) { return fail(LGIO_E_OOM, "out of memory (length)"); }
    catch (const std::exception& e) { return fail(LGIO_E_FORMAT, "%s", e.what()); }
    catch (...) { return fail(LGIO_E_FORMAT, "unknown error"); }
}
extern "C" int lgio_bam_open(const char* path, lgio_bam** out) { return guarded([&] { return bam_open_impl(path, out); }); }

extern "C" void lgio_bam_close(lgio_bam* b) { delete b; }
extern "C" int lgio_bam_n_refs(const lgio_bam* b) { return b ? (int)b->ref_names.size() : 0; }
extern "C" const char* lgio_bam_ref_name(const lgio_bam* b, int tid) {
    return (b && tid >= 0 && (size_t)tid < b->ref_names.size()) ? b->ref_names[tid].c_str() : nullptr;
}
extern "C" int64_t lgio_bam_ref_length(const lgio_bam* b, int tid) {
    return (b && tid >= 0 && (size_t)tid < b->ref_lens.size()) ? b->ref_lens[tid] : -1;
}
extern "C" const char* lgio_bam_header_text(const lgio_bam* b) { return b ? b->header_text.c_str() : nullptr; }
extern "C" int lgio_bam_has_index_file(const lgio_bam* b) { return b && b->index_from_file ? 1 : 0; }
// ---------------------------------------------------------------- the removed-site table, formatted natively (lgmi_io.h)
static int write_removed_impl(const char* path, int append, int header, uint64_t n, const int32_t* chrom_code, const int8_t* strand,
                              const int64_t* pos, const int8_t* reason_code, const char* const* chrom_names, uint32_t n_chrom,
                              const char* const* reason_names, uint32_t n_reasons, int threads) {
    if (!path || (n && (!chrom_code || !strand || !pos || !reason_code))) return fail(LGIO_E_ARG, "NULL argument");
    auto plain = [](const char* s) { return s && *s && !strpbrk(s, "\t\"\r\n"); };
    std::vector<std::string> cn(n_chrom), rn(n_reasons);
    for (uint32_t k = 0; k < n_chrom; ++k) { if (!plain(chrom_names[k])) return fail(LGIO_E_ARG, "chromosome name %u needs quoting", k); cn[k] = chrom_names[k]; }
    for (uint32_t k = 0; k < n_reasons; ++k) { if (!plain(reason_names[k])) return fail(LGIO_E_ARG, "reason %u needs quoting", k); rn[k] = reason_names[k]; }
    for (uint64_t k = 0; k < n; ++k)
        if (chrom_code[k] < 0 || (uint32_t)chrom_code[k] >= n_chrom || reason_code[k] < 0 || (uint32_t)reason_code[k] >= n_reasons || (strand[k] & ~1))
            return fail(LGIO_E_ARG, "row %llu: code out of range", (unsigned long long)k);
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (n < 200000) threads = 1;
    std::vector<std::string> buf((size_t)threads);
    auto format = [&](int t) {
        const uint64_t k0 = n * (uint64_t)t / (uint64_t)threads, k1 = n * (uint64_t)(t + 1) / (uint64_t)threads;
        std::string& out = buf[(size_t)t];
        out.reserve((size_t)(k1 - k0) * 56);
        char num[24];
        for (uint64_t k = k0; k < k1; ++k) {
            out += cn[(size_t)chrom_code[k]];
            out += strand[k] ? "\t-\t" : "\t+\t";
            // the decimal digits of pos, as str(int) writes them
            int64_t v = pos[k];
            const bool neg = v < 0;
            uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
            int at = 24;
            do { num[--at] = (char)('0' + u % 10); u /= 10; } while (u);
            if (neg) num[--at] = '-';
            out.append(num + at, (size_t)(24 - at));
            out += '\t';
            out += rn[(size_t)reason_code[k]];
            out += '\n';
        }
    };
    {
        std::vector<std::thread> th;
        int started = 1;
        for (; started < threads; ++started) { try { th.emplace_back(format, started); } catch (...) { break; } }
        format(0);
        for (int t = started; t < threads; ++t) format(t);           // (threads the system refused: their ranges here)
        for (auto& x : th) x.join();
    }
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return fail(LGIO_E_IO, "cannot open %s for writing", path);
    bool ok = true;
    if (header) ok = fputs("chromosome\tstrand\tpos\tremoved\n", f) >= 0;
    for (const std::string& b : buf) if (ok && !b.empty()) ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    if (fclose(f) != 0) ok = false;
    return ok ? LGIO_OK : fail(LGIO_E_IO, "writing %s failed", path);
}
extern "C" int lgio_write_removed_table(const char* path, int append, int header, uint64_t n, const int32_t* chrom_code,
                                        const int8_t* strand, const int64_t* pos, const int8_t* reason_code,
                                        const char* const* chrom_names, uint32_t n_chrom, const char* const* reason_names,
                                        uint32_t n_reasons, int threads) {
    return guarded([&] { return write_removed_impl(path, append, header, n, chrom_code, strand, pos, reason_code, chrom_names, n_chrom,
                                                   reason_names, n_reasons, threads); });
}

// ---------------------------------------------------------------- any table of integers, floats and dictionary strings (lgmi_io.h)
// A float64 the way numpy's astype(str) — pandas' to_csv — writes it: the shortest digits that read back as the same
// double (std::to_chars gives them), positional for 1e-4 <= |x| < 1e16 with at least one digit behind the point, otherwise
// d[.ddd]e+XX with at least two exponent digits; "nan" is the caller's empty field.  Returns the length written (<= 32).
static int format_double_repr(double x, char* out) {
    if (std::isnan(x)) return 0;
    char* p = out;
    if (std::signbit(x)) { *p++ = '-'; x = -x; }
    if (std::isinf(x)) { memcpy(p, "inf", 3); return (int)(p + 3 - out); }
    if (x == 0.0) { memcpy(p, "0.0", 3); return (int)(p + 3 - out); }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);      // d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q) if (*q != '.') digits[nd++] = *q;
    int e10 = 0;
    { const char* e = q + 1; const bool neg = *e == '-'; if (*e == '+' || *e == '-') ++e; for (; e < r.ptr; ++e) e10 = e10 * 10 + (*e - '0'); if (neg) e10 = -e10; }
    if (x >= 1e-4 && x < 1e16) {
        if (e10 >= 0) {
            for (int k = 0; k <= e10; ++k) *p++ = k < nd ? digits[k] : '0';
            *p++ = '.';
            if (nd > e10 + 1) { memcpy(p, digits + e10 + 1, (size_t)(nd - e10 - 1)); p += nd - e10 - 1; } else *p++ = '0';
        } else {
            *p++ = '0'; *p++ = '.';
            for (int k = 0; k < -e10 - 1; ++k) *p++ = '0';
            memcpy(p, digits, (size_t)nd); p += nd;
        }
    } else {
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, (size_t)(nd - 1)); p += nd - 1; }
        *p++ = 'e';
        *p++ = e10 < 0 ? '-' : '+';
        const int a = e10 < 0 ? -e10 : e10;
        if (a >= 100) *p++ = (char)('0' + a / 100);
        *p++ = (char)('0' + a / 10 % 10);
        *p++ = (char)('0' + a % 10);
    }
    return (int)(p - out);
}

extern "C" int lgio_format_doubles(uint64_t n, const double* x, char* out, uint32_t stride) {
    return guarded([&] {
        if ((n && (!x || !out)) || stride < 33) return fail(LGIO_E_ARG, "lgio_format_doubles: NULL argument or stride < 33");
        for (uint64_t k = 0; k < n; ++k) { char* o = out + k * stride; const int len = format_double_repr(x[k], o); o[len] = 0; }
        return (int)LGIO_OK;
    });
}

static int write_table_impl(const char* path, int append, int header, uint64_t n, uint32_t n_cols, const lgio_table_col* cols, int threads) {
    if (!path || !n_cols || !cols) return fail(LGIO_E_ARG, "NULL argument");
    auto plain = [](const char* s) { return s && *s && !strpbrk(s, "\t\"\r\n"); };
    std::vector<std::vector<std::string>> names(n_cols);
    for (uint32_t c = 0; c < n_cols; ++c) {
        const lgio_table_col& col = cols[c];
        if (!plain(col.name)) return fail(LGIO_E_ARG, "column %u: a name that needs quoting", c);
        if (col.kind > LGIO_COL_DICT || (n && !col.data)) return fail(LGIO_E_ARG, "column %u: kind %u / no data", c, col.kind);
        if (col.kind == LGIO_COL_DICT) {
            names[c].resize(col.n_names);
            for (uint32_t k = 0; k < col.n_names; ++k) {
                if (!col.names || !plain(col.names[k])) return fail(LGIO_E_ARG, "column %u: name %u needs quoting", c, k);
                names[c][k] = col.names[k];
            }
            const int32_t* code = static_cast<const int32_t*>(col.data);
            for (uint64_t k = 0; k < n; ++k)
                if (code[k] < 0 || (uint32_t)code[k] >= col.n_names) return fail(LGIO_E_ARG, "column %u, row %llu: code out of range", c, (unsigned long long)k);
        }
    }
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (n < 100000) threads = 1;
    std::vector<std::string> buf((size_t)threads);
    auto format = [&](int t) {
        const uint64_t k0 = n * (uint64_t)t / (uint64_t)threads, k1 = n * (uint64_t)(t + 1) / (uint64_t)threads;
        std::string& out = buf[(size_t)t];
        out.reserve((size_t)(k1 - k0) * 24 * n_cols);
        char num[40];
        for (uint64_t k = k0; k < k1; ++k)
            for (uint32_t c = 0; c < n_cols; ++c) {
                const lgio_table_col& col = cols[c];
                if (col.kind == LGIO_COL_DICT) out += names[c][(size_t)static_cast<const int32_t*>(col.data)[k]];
                else if (col.kind == LGIO_COL_F64) out.append(num, (size_t)format_double_repr(static_cast<const double*>(col.data)[k], num));
                else {
                    const int64_t v = static_cast<const int64_t*>(col.data)[k];
                    const bool neg = v < 0;
                    uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
                    int at = 24;
                    do { num[--at] = (char)('0' + u % 10); u /= 10; } while (u);
                    if (neg) num[--at] = '-';
                    out.append(num + at, (size_t)(24 - at));
                }
                out += c + 1 < n_cols ? '\t' : '\n';
            }
    };
    {
        std::vector<std::thread> th;
        int started = 1;
        for (; started < threads; ++started) { try { th.emplace_back(format, started); } catch (...) { break; } }
        format(0);
        for (int t = started; t < threads; ++t) format(t);           // (threads the system refused: their ranges here)
        for (auto& x : th) x.join();
    }
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return fail(LGIO_E_IO, "cannot open %s for writing", path);
    bool ok = true;
    if (header) {
        std::string h;
        for (uint32_t c = 0; c < n_cols; ++c) { h += cols[c].name; h += c + 1 < n_cols ? '\t' : '\n'; }
        ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    }
    for (const std::string& b : buf) if (ok && !b.empty()) ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    if (fclose(f) != 0) ok = false;
    return ok ? LGIO_OK : fail(LGIO_E_IO, "writing %s failed", path);
}
extern "C" int lgio_write_table(const char* path, int append, int header, uint64_t n_rows, uint32_t n_cols, const lgio_table_col* cols,
                                int threads) {
    return guarded([&] { return write_table_impl(path, append, header, n_rows, n_cols, cols, threads); });
}

extern "C" uint64_t lgio_bam_bytes_read(const lgio_bam* b) { return b ? b->z.bytes_read : 0; }

static int build_index_impl(const char* bam_path, const char* bai_path) {
    if (!bam_path) return fail(LGIO_E_ARG, "NULL argument");
    lgio_bam b;
    b.path = bam_path;
    b.z.f = fopen(bam_path, "rb");
    if (!b.z.f) return fail(LGIO_E_IO, "cannot open %s", bam_path);
    int rc = read_header(&b);
    if (rc) return rc;
    std::vector<RefIndex> idx;
    if ((rc = scan_index(&b, idx))) return rc;
    return write_bai(bai_path ? std::string(bai_path) : b.path + ".bai", idx);
}
extern "C" int lgio_bam_build_index(const char* bam_path, const char* bai_path) {
    return guarded([&] { return build_index_impl(bam_path, bai_path); });
}

extern "C" void lgio_reads_free(lgio_reads* r) {
    if (!r) return;
    delete static_cast<ReadsOwner*>(r->owner_);
    memset(r, 0, sizeof *r);
}

static int fetch_impl(lgio_bam* b, int tid, int64_t start, int64_t end, uint32_t what, lgio_reads* out) {
    if (!b || !out) return fail(LGIO_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    ReadsOwner* o = new ReadsOwner();
    struct Drop { ReadsOwner* p; ~Drop() { delete p; } } drop{o};
    const int rc = for_each_overlap(b, tid, start, end, [&](const Record& r) { o->add(r, what); });
    if (rc) return rc;
    o->view(out);
    drop.p = nullptr;
    return LGIO_OK;
}
extern "C" int lgio_bam_fetch(lgio_bam* b, int tid, int64_t start, int64_t end, uint32_t what, lgio_reads* out) {
    return guarded([&] { return fetch_impl(b, tid, start, end, what, out); });
}

namespace {
struct PileOwner { ReadsOwner reads; std::vector<int64_t> pos; std::vector<uint64_t> col_off; std::vector<uint32_t> read; std::vector<char> base; };
}

extern "C" void lgio_pileup_free(lgio_pileup* p) {
    if (!p) return;
    delete static_cast<PileOwner*>(p->owner_);
    memset(p, 0, sizeof *p);
}

static int pileup_impl(lgio_bam* b, int tid, int64_t start, int64_t end, int min_bq, int max_depth, lgio_pileup* out) {
    if (!b || !out) return fail(LGIO_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    if (max_depth <= 0) max_depth = 8000;
    PileOwner* o = new PileOwner();
    struct Drop { PileOwner* p; ~Drop() { delete p; } } drop{o};
    // pass 1: the reads (records kept whole: the second pass walks their CIGARs)
    std::vector<Record> recs;
    int rc = for_each_overlap(b, tid, start, end, [&](const Record& r) {
        if (r.flag & (4 | 256 | 512 | 1024)) return;            // unmapped, secondary, QC-fail, duplicate
        if ((r.flag & 1) && !(r.flag & 2)) return;              // orphan of a paired read
        recs.push_back(r);
    });
    if (rc) return rc;
    if (!recs.empty()) {
        int64_t lo = recs[0].pos, hi = recs[0].end;
        for (const Record& r : recs) { lo = std::min(lo, r.pos); hi = std::max(hi, r.end); }
        // three arrays of this length are made below: a region's reads spanning more than 2^31 positions is not a
        // footprint any caller of this path produces (a chromosome is < 2^28), and not something to try to allocate
        if (hi - lo > (int64_t)1 << 31) return fail(LGIO_E_ARG, "pile-up over %lld reference positions refused", (long long)(hi - lo));
        const size_t span = (size_t)(hi - lo);
        // columns are not truncated to [start, end): every position any of the reads aligns to
        std::vector<uint32_t> depth(span, 0);
        static const char code[] = "=ACMGRSVTWYHKDBN";
        // walk(r, emit): emit(ref position, base or 0) for every reference position of the read
        auto walk = [&](const Record& r, auto emit) {
            int64_t ref = r.pos; uint32_t q = 0;
            const uint8_t *c = r.cigar(), *s = r.seq(), *ql = r.qual();
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t v = le32(c + 4 * k), op = v & 0xF, n = v >> 4;
                if (op == 0 || op == 7 || op == 8) {
                    for (uint32_t j = 0; j < n; ++j, ++ref, ++q) {
                        if (q < r.l_seq && ql[q] != 0xFF && ql[q] < (uint32_t)min_bq) continue;
                        emit(ref, q < r.l_seq ? code[(s[q >> 1] >> ((~q & 1) << 2)) & 0xF] : 'N');
                    }
                } else if (op == 2 || op == 3) {
                    for (uint32_t j = 0; j < n; ++j, ++ref) emit(ref, (char)0);
                } else if (op == 1 || op == 4) q += n;
            }
        };
        for (const Record& r : recs)
            walk(r, [&](int64_t p, char) { uint32_t& d = depth[(size_t)(p - lo)]; if (d < (uint32_t)max_depth) ++d; });
        std::vector<uint64_t> at(span + 1, 0);
        for (size_t k = 0; k < span; ++k) at[k + 1] = at[k] + depth[k];
        o->read.resize((size_t)at[span]); o->base.resize((size_t)at[span]);
        std::vector<uint32_t> fill(span, 0);
        for (size_t ri = 0; ri < recs.size(); ++ri)
            walk(recs[ri], [&](int64_t p, char bs) {
                const size_t k = (size_t)(p - lo);
                if (fill[k] < depth[k]) { o->read[(size_t)at[k] + fill[k]] = (uint32_t)ri; o->base[(size_t)at[k] + fill[k]] = bs; ++fill[k]; }
            });
        o->col_off.push_back(0);
        for (size_t k = 0; k < span; ++k)
            if (depth[k]) { o->pos.push_back(lo + (int64_t)k); o->col_off.push_back(at[k + 1]); }
        for (const Record& r : recs) o->reads.add(r, LGIO_NAMES);
    } else {
        o->col_off.push_back(0);
    }
    o->reads.view(&out->reads);
    out->reads.owner_ = nullptr;                    // owned by the pile-up
    out->n_cols = o->pos.size();
    out->pos = o->pos.data(); out->col_off = o->col_off.data(); out->read = o->read.data(); out->base = o->base.data();
    out->owner_ = o;
    drop.p = nullptr;
    return LGIO_OK;
}
extern "C" int lgio_bam_pileup(lgio_bam* b, int tid, int64_t start, int64_t end, int min_bq, int max_depth, lgio_pileup* out) {
    return guarded([&] { return pileup_impl(b, tid, start, end, min_bq, max_depth, out); });
}

// ---------------------------------------------------------------- site extraction of one footprint (lgmi_io.h, round 3)
// The specification is lgmi/region.py's get_region_mismatches_with_filters (steps 1-6 there, mismatch.py:29-290 of
// the reference); the comments name its steps.  Anything that routine would treat in a way this one does not
// reproduce comes back as fallback = 1.
namespace {

struct SiteAllele { char nt; bool kept = true; std::vector<uint32_t> reads; };
struct Site {
    int64_t pos; char ref = 0; std::vector<SiteAllele> alleles; uint32_t neighbor[16] = {};
    bool gone = false;
    SiteAllele& allele(char nt) {
        for (SiteAllele& a : alleles) if (a.nt == nt) return a;
        alleles.push_back(SiteAllele{nt, true, {}});
        return alleles.back();
    }
};
struct StrandSites {
    std::vector<Site> sites;                        // first-seen order
    std::map<int64_t, size_t> at;                   // position -> index in sites
    Site& site(int64_t pos) {
        auto it = at.find(pos);
        if (it != at.end()) return sites[it->second];
        at.emplace(pos, sites.size());
        sites.push_back(Site{pos});
        return sites.back();
    }
};
struct SitesOwner {
    std::vector<uint8_t> strand; std::vector<int64_t> pos; std::vector<char> ref; std::vector<uint32_t> neighbor;
    std::vector<uint64_t> allele_off{0}; std::vector<char> allele_nt; std::vector<uint64_t> reads_off{0}; std::vector<uint32_t> reads;
    std::vector<int64_t> removed_pos[2]; std::vector<uint8_t> removed_code[2];
    std::vector<uint64_t> name_off{0}; std::vector<char> names;
    std::vector<uint32_t> read_uid;
};
struct Sub { int64_t pos; char ref, alt; };

inline int base_index(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
inline bool is_op(char c) { return c == ':' || c == '*' || c == '+' || c == '-' || c == '~'; }

// minimap2's short cs form only: ':'n  '*'xy  '+'seq  '-'seq  '~'xxNyy, lower-case bases; false = let the caller decide
bool parse_cs(const uint8_t* s, size_t n, int64_t origin, std::vector<Sub>& subs, std::vector<int64_t>& junctions) {
    int64_t at = 0;
    size_t i = 0;
    auto lower = [](uint8_t c) { return c >= 'a' && c <= 'z'; };
    auto digit = [](uint8_t c) { return c >= '0' && c <= '9'; };
    while (i < n) {
        const char op = (char)s[i++];
        size_t j = i;
        if (op == ':') {
            int64_t v = 0;
            while (j < n && digit(s[j])) { v = v * 10 + (s[j] - '0'); if (v > ((int64_t)1 << 40)) return false; ++j; }
            if (j == i) return false;
            at += v;
        } else if (op == '*') {
            if (i + 2 > n) return false;
            const char r = (char)s[i], a = (char)s[i + 1];
            auto acgt = [](char c) { return c == 'a' || c == 'c' || c == 'g' || c == 't'; };
            if (!acgt(r) || !acgt(a)) return false;
            subs.push_back(Sub{origin + at, (char)(r - 32), (char)(a - 32)});
            at += 1;
            j = i + 2;
        } else if (op == '+' || op == '-') {
            while (j < n && lower(s[j])) ++j;
            if (j == i) return false;
            if (op == '-') at += (int64_t)(j - i);
        } else if (op == '~') {
            if (i + 2 > n || !lower(s[i]) || !lower(s[i + 1])) return false;
            j = i + 2;
            int64_t v = 0;
            const size_t d0 = j;
            while (j < n && digit(s[j])) { v = v * 10 + (s[j] - '0'); if (v > ((int64_t)1 << 40)) return false; ++j; }
            if (j == d0 || j + 2 > n || !lower(s[j]) || !lower(s[j + 1])) return false;
            j += 2;
            junctions.push_back(origin + at);
            junctions.push_back(origin + at + v);
            at += v;
        } else {
            return false;
        }
        if (j < n && !is_op((char)s[j])) return false;      // the value ends where the next operation starts
        i = j;
    }
    return true;
}

}  // namespace

extern "C" void lgio_sites_free(lgio_sites* s) {
    if (!s) return;
    delete static_cast<SitesOwner*>(s->owner_);
    memset(s, 0, sizeof *s);
}

static int region_sites_impl(lgio_bam* b, int tid, int64_t start, int64_t end, const lgio_site_params* P, lgio_sites* out) {
    if (!b || !out || !P) return fail(LGIO_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    if (start < 0 || end < start) return fail(LGIO_E_ARG, "region_sites needs 0 <= start <= end");
    const int max_depth = P->max_depth > 0 ? P->max_depth : 8000;
    std::vector<Record> recs;
    int rc = for_each_overlap(b, tid, start, end, [&](const Record& r) { recs.push_back(r); });
    if (rc) return rc;
    SitesOwner* o = new SitesOwner();
    struct Drop { SitesOwner* p; ~Drop() { delete p; } } drop{o};
    auto finish = [&](int fallback) {
        out->fallback = fallback;
        out->n_sites = o->pos.size();
        out->strand = o->strand.data(); out->pos = o->pos.data(); out->ref = o->ref.data(); out->neighbor = o->neighbor.data();
        out->allele_off = o->allele_off.data(); out->allele_nt = o->allele_nt.data();
        out->reads_off = o->reads_off.data(); out->reads = o->reads.data();
        for (int s = 0; s < 2; ++s) {
            out->n_removed[s] = o->removed_pos[s].size();
            out->removed_pos[s] = o->removed_pos[s].data(); out->removed_code[s] = o->removed_code[s].data();
        }
        out->n_reads = o->name_off.size() - 1;
        out->name_off = o->name_off.data(); out->names = o->names.data();
        out->read_uid = o->read_uid.data();
        out->owner_ = o;
        drop.p = nullptr;
        return LGIO_OK;
    };

    // ---- 1. the strand of a read NAME is that of its first record (read_strand_dict, :77-81); substitutions of the
    //         spliced reads, minus those next to a junction (:69-149)
    std::vector<uint8_t> strand(recs.size());
    {
        std::map<std::string, std::pair<uint8_t, uint32_t>> by_name;      // name -> (strand, index) of its first record
        o->read_uid.resize(recs.size());
        for (size_t ri = 0; ri < recs.size(); ++ri) {
            const Record& r = recs[ri];
            std::string name((const char*)r.name(), r.l_name - 1);
            o->names.insert(o->names.end(), name.begin(), name.end());
            o->name_off.push_back(o->names.size());
            auto ins = by_name.emplace(std::move(name), std::make_pair((uint8_t)((r.flag & 16) ? 1 : 0), (uint32_t)ri));
            strand[ri] = ins.first->second.first;
            o->read_uid[ri] = ins.first->second.second;
        }
    }
    StrandSites S[2];
    std::vector<Sub> subs;
    std::vector<int64_t> junctions;
    for (size_t ri = 0; ri < recs.size(); ++ri) {
        const uint8_t* text; size_t len;
        if (!find_cs(recs[ri], &text, &len)) return finish(1);
        subs.clear(); junctions.clear();
        if (!parse_cs(text, len, recs[ri].pos, subs, junctions)) return finish(1);
        if (!P->keep_non_spliced_read && junctions.empty()) continue;
        const int64_t d = P->min_dist_from_splice;
        for (const Sub& sb : subs) {
            bool close = false;
            if (d > 0) for (int64_t j : junctions) if (j - d <= sb.pos && sb.pos < j + d) { close = true; break; }
            if (close) continue;
            Site& site = S[strand[ri]].site(sb.pos);
            site.ref = sb.ref;
            site.allele(sb.alt).reads.push_back((uint32_t)ri);
        }
    }

    // ---- 2. the pile-up (pysam defaults, as lgio_bam_pileup): which positions have a column, and at the candidate
    //         sites the reads of the site's strand that show the reference base (:160-190)
    int64_t lo = 0, hi = 0;
    for (size_t ri = 0; ri < recs.size(); ++ri) {
        lo = ri ? std::min(lo, recs[ri].pos) : recs[ri].pos;
        hi = ri ? std::max(hi, recs[ri].end) : recs[ri].end;
    }
    if (hi - lo > (int64_t)1 << 31) return fail(LGIO_E_ARG, "pile-up over %lld reference positions refused", (long long)(hi - lo));
    const size_t span = (size_t)(hi - lo);
    std::vector<uint32_t> depth(span, 0);
    std::vector<int32_t> slot[2];
    size_t n_known[2];
    std::vector<std::pair<int64_t, int32_t>> cand[2];           // the strand's candidate sites inside the span, by position
    for (int s = 0; s < 2; ++s) {
        n_known[s] = S[s].sites.size();
        if (!n_known[s]) continue;
        slot[s].assign(span, -1);
        for (size_t k = 0; k < n_known[s]; ++k) {
            const int64_t p = S[s].sites[k].pos;
            if (p >= lo && p < hi) { slot[s][(size_t)(p - lo)] = (int32_t)k; cand[s].emplace_back(p, (int32_t)k); }
        }
        std::sort(cand[s].begin(), cand[s].end());
    }
    static const char code[] = "=ACMGRSVTWYHKDBN";
    const uint32_t min_bq = (uint32_t)(P->min_base_quality > 0 ? P->min_base_quality : 0);
    auto in_pileup = [](const Record& r) { return !(r.flag & (4 | 256 | 512 | 1024)) && !((r.flag & 1) && !(r.flag & 2)); };
    // Fast form (round 5): the work of a read is its CIGAR segments and the candidate sites inside them, not its bases — a
    // segment adds one to a range of columns (difference array; a base below min_base_quality takes its one back), and only
    // at a candidate site is a base decoded.  Exact as long as no column exceeds max_depth (then "the first max_depth reads
    // of a column" is everybody); otherwise the base-by-base form below is run instead.  500 reads x 1,000 bases per
    // footprint were 1.7 of the 5.4 ms a footprint cost natively.
    bool fast_ok = true;
    {
        std::vector<int32_t> diff(span + 1, 0);
        struct Hit { int32_t site; uint8_t strand; uint32_t read; };
        std::vector<Hit> hits;
        for (size_t ri = 0; ri < recs.size(); ++ri) {
            const Record& r = recs[ri];
            if (!in_pileup(r)) continue;
            int64_t ref = r.pos; uint32_t q = 0;
            const uint8_t *c = r.cigar(), *sq = r.seq(), *ql = r.qual();
            const int s = strand[ri];
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t v = le32(c + 4 * k), op = v & 0xF, n = v >> 4;
                if (op == 0 || op == 7 || op == 8) {
                    ++diff[(size_t)(ref - lo)]; --diff[(size_t)(ref - lo) + n];
                    const uint32_t nq = q < r.l_seq ? std::min<uint32_t>(n, r.l_seq - q) : 0u;     // bases that have a quality
                    uint8_t worst = 0xFF;                                  // (a min-reduction the compiler vectorises: no base of a HiFi
                    for (uint32_t j = 0; j < nq; ++j) worst = std::min(worst, ql[q + j]);   //  or Q20+ read is below 13)
                    if (worst < min_bq)
                        for (uint32_t j = 0; j < nq; ++j)
                            if (ql[q + j] < min_bq) { --diff[(size_t)(ref - lo) + j]; ++diff[(size_t)(ref - lo) + j + 1]; }   // (0xFF: no qualities, never below)
                    if (!cand[s].empty()) {
                        auto it = std::lower_bound(cand[s].begin(), cand[s].end(), std::make_pair(ref, (int32_t)-1));
                        for (; it != cand[s].end() && it->first < ref + (int64_t)n; ++it) {
                            const uint32_t qq = q + (uint32_t)(it->first - ref);
                            if (qq < r.l_seq && ql[qq] != 0xFF && ql[qq] < min_bq) continue;
                            const char base = qq < r.l_seq ? code[(sq[qq >> 1] >> ((~qq & 1) << 2)) & 0xF] : 'N';
                            if (base == S[s].sites[(size_t)it->second].ref) hits.push_back(Hit{it->second, (uint8_t)s, (uint32_t)ri});
                        }
                    }
                    ref += n; q += n;
                } else if (op == 2 || op == 3) {
                    ++diff[(size_t)(ref - lo)]; --diff[(size_t)(ref - lo) + n];
                    ref += n;
                } else if (op == 1 || op == 4) q += n;
            }
        }
        int64_t run = 0;
        for (size_t k = 0; k < span; ++k) { run += diff[k]; depth[k] = (uint32_t)run; if (run > (int64_t)max_depth) fast_ok = false; }
        if (fast_ok)
            for (const Hit& h : hits) { Site& site = S[h.strand].sites[(size_t)h.site]; site.allele(site.ref).reads.push_back(h.read); }
    }
    if (!fast_ok) {
        std::fill(depth.begin(), depth.end(), 0u);
        for (size_t ri = 0; ri < recs.size(); ++ri) {
            const Record& r = recs[ri];
            if (!in_pileup(r)) continue;
            int64_t ref = r.pos; uint32_t q = 0;
            const uint8_t *c = r.cigar(), *sq = r.seq(), *ql = r.qual();
            auto emit = [&](int64_t p, char base) {
                const size_t k = (size_t)(p - lo);
                if (depth[k] >= (uint32_t)max_depth) return;
                ++depth[k];
                const int s = strand[ri];
                if (!slot[s].empty() && slot[s][k] >= 0) {
                    Site& site = S[s].sites[(size_t)slot[s][k]];
                    if (base == site.ref) site.allele(site.ref).reads.push_back((uint32_t)ri);
                }
            };
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t v = le32(c + 4 * k), op = v & 0xF, n = v >> 4;
                if (op == 0 || op == 7 || op == 8) {
                    for (uint32_t j = 0; j < n; ++j, ++ref, ++q) {
                        if (q < r.l_seq && ql[q] != 0xFF && ql[q] < min_bq) continue;
                        emit(ref, q < r.l_seq ? code[(sq[q >> 1] >> ((~q & 1) << 2)) & 0xF] : 'N');
                    }
                } else if (op == 2 || op == 3) {
                    for (uint32_t j = 0; j < n; ++j, ++ref) emit(ref, (char)0);
                } else if (op == 1 || op == 4) q += n;
            }
        }
    }
    // (a candidate site whose column exists gets the reference allele's key even with no read in it: the depth dict
    //  then holds a zero for it, which changes nothing below but is what the Python path holds)
    for (int s = 0; s < 2; ++s)
        for (size_t k = 0; k < n_known[s]; ++k) {
            Site& site = S[s].sites[k];
            const int64_t p = site.pos;
            if (p >= lo && p < hi && depth[(size_t)(p - lo)]) site.allele(site.ref);
        }

    for (int s = 0; s < 2; ++s) {
        if (!n_known[s]) continue;
        StrandSites& T = S[s];
        // every position with a column is a site of this strand now (the look-up creates it, :166): the snapshot of step 4
        std::vector<int64_t> snap;
        std::vector<int32_t> sidx;                  // index into T.sites, -1 for the empty ones
        {
            auto it = T.at.begin();
            for (size_t k = 0; k < span || it != T.at.end();) {
                const int64_t pc = k < span ? lo + (int64_t)k : INT64_MAX;
                if (it != T.at.end() && it->first <= pc) {
                    snap.push_back(it->first); sidx.push_back((int32_t)it->second);
                    if (it->first == pc) ++k;
                    ++it;
                } else {
                    if (depth[k]) { snap.push_back(pc); sidx.push_back(-1); }
                    ++k;
                }
            }
        }
        const size_t n = snap.size();
        // ---- 4. the window filter (:211-240).  cpos: the sites that add to their neighbours' counts
        std::vector<size_t> cpos;
        for (size_t k = 0; k < n; ++k) {
            if (sidx[k] < 0) continue;
            const Site& q = T.sites[(size_t)sidx[k]];
            for (const SiteAllele& a : q.alleles) if (a.nt != q.ref) { cpos.push_back(k); break; }
        }
        std::vector<uint8_t> dead(n, 0), back(n, 0);
        std::vector<size_t> removed_here;
        const int64_t half = P->half_window;
        size_t wa = 0, wb = 0, ca = 0, cb = 0, ra = 0;
        for (size_t k = 0; k < n; ++k) {
            const int64_t wl = snap[k] - half, wh = snap[k] + half;
            while (wa < n && snap[wa] < wl) ++wa;
            if (wb < wa) wb = wa;
            while (wb < n && snap[wb] < wh) ++wb;
            if (wb - wa < 2) continue;
            while (ra < removed_here.size() && snap[removed_here[ra]] < wl) ++ra;
            for (size_t x = ra; x < removed_here.size() && snap[removed_here[x]] < wh; ++x) back[removed_here[x]] = 1;
            while (ca < cpos.size() && snap[cpos[ca]] < wl) ++ca;
            if (cb < ca) cb = ca;
            while (cb < cpos.size() && snap[cpos[cb]] < wh) ++cb;
            uint32_t cnt[16] = {};
            for (size_t x = ca; x < cb; ++x) {
                const size_t qk = cpos[x];
                if (qk == k || dead[qk]) continue;
                const Site& q = T.sites[(size_t)sidx[qk]];
                const int rb = base_index(q.ref);
                for (const SiteAllele& a : q.alleles) if (a.nt != q.ref) ++cnt[4 * rb + base_index(a.nt)];
            }
            uint64_t total = 0; int kinds = 0;
            for (int t = 0; t < 16; ++t) { total += cnt[t]; kinds += cnt[t] != 0; }
            if (sidx[k] >= 0) memcpy(T.sites[(size_t)sidx[k]].neighbor, cnt, sizeof cnt);
            if ((double)total > P->max_window_mismatch && (double)kinds > P->max_window_mismatch_type) {
                dead[k] = 1;
                removed_here.push_back(k);
            }
        }
        std::vector<int64_t>& gpos = o->removed_pos[s];
        std::vector<uint8_t>& gcode = o->removed_code[s];
        std::vector<size_t> gslot(n, (size_t)-1);
        for (size_t k : removed_here) { gslot[k] = gpos.size(); gpos.push_back(snap[k]); gcode.push_back(LGIO_REMOVED_WINDOW); }
        // ---- 5. shallow alleles, rare alleles (:243-266); 6. shallow sites, sites left with one allele (:268-290)
        std::vector<uint8_t> out_now(n, 0);
        auto remove = [&](size_t k, uint8_t why) {
            out_now[k] = 1;
            if (gslot[k] != (size_t)-1) gcode[gslot[k]] = why;      // a site that came back empty: same dict key, new value
            else { gpos.push_back(snap[k]); gcode.push_back(why); }
            if (sidx[k] >= 0) T.sites[(size_t)sidx[k]].gone = true;
        };
        for (size_t k = 0; k < n; ++k) {
            if (dead[k]) {
                if (sidx[k] >= 0) T.sites[(size_t)sidx[k]].gone = true;
                if (!back[k]) { out_now[k] = 1; continue; }
                if (0.0 < P->min_total_depth) remove(k, LGIO_REMOVED_DEPTH);
                continue;
            }
            uint64_t total = 0;
            if (sidx[k] >= 0) {
                Site& site = T.sites[(size_t)sidx[k]];
                for (SiteAllele& a : site.alleles) total += a.reads.size();
                for (SiteAllele& a : site.alleles) if ((double)a.reads.size() < P->min_allele_depth) a.kept = false;
                for (SiteAllele& a : site.alleles)
                    if (a.kept && (double)a.reads.size() / (double)total < P->min_allele_ratio) a.kept = false;
            }
            if ((double)total < P->min_total_depth) remove(k, LGIO_REMOVED_DEPTH);
        }
        for (size_t k = 0; k < n; ++k) {
            if (out_now[k]) continue;
            size_t kept = 0;
            if (sidx[k] >= 0 && !dead[k]) for (const SiteAllele& a : T.sites[(size_t)sidx[k]].alleles) kept += a.kept;
            if (kept < 2) remove(k, LGIO_REMOVED_ALLELES);
        }
        // the survivors, in the order the reads first showed them
        for (const Site& site : T.sites) {
            if (site.gone) continue;
            o->strand.push_back((uint8_t)s); o->pos.push_back(site.pos); o->ref.push_back(site.ref);
            o->neighbor.insert(o->neighbor.end(), site.neighbor, site.neighbor + 16);
            for (const SiteAllele& a : site.alleles) {
                if (!a.kept) continue;
                o->allele_nt.push_back(a.nt);
                o->reads.insert(o->reads.end(), a.reads.begin(), a.reads.end());
                o->reads_off.push_back(o->reads.size());
            }
            o->allele_off.push_back(o->allele_nt.size());
        }
    }
    return finish(0);
}
extern "C" int lgio_bam_region_sites(lgio_bam* b, int tid, int64_t start, int64_t end, const lgio_site_params* P, lgio_sites* out) {
    return guarded([&] { return region_sites_impl(b, tid, start, end, P, out); });
}

// ---------------------------------------------------------------- whole-reference interval scan, blocks inflated in parallel
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <thread>

namespace {

struct IntervalsOwner { std::vector<int64_t> start, end; };
struct BlockRef { uint64_t addr; uint32_t head; uint32_t clen; };      // file offset, bytes before the deflate stream, its length

// header of the BGZF block at `addr` (pread: no shared file position) -> 0 ok, 1 end of file, negative error
int block_ref(int fd, uint64_t addr, BlockRef& br, uint64_t& next) {
    uint8_t h[18];
    const ssize_t got = pread(fd, h, 18, (off_t)addr);
    if (got == 0) return 1;
    if (got != 18) return fail(LGIO_E_FORMAT, "truncated BGZF block header at %llu", (unsigned long long)addr);
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(LGIO_E_FORMAT, "not a BGZF block at %llu", (unsigned long long)addr);
    const uint32_t xlen = le16(h + 10);
    uint32_t bsize = 0;
    if (xlen == 6 && h[12] == 'B' && h[13] == 'C') {
        bsize = le16(h + 16);
    } else {
        std::vector<uint8_t> extra(xlen);
        if (pread(fd, extra.data(), xlen, (off_t)(addr + 12)) != (ssize_t)xlen) return fail(LGIO_E_FORMAT, "truncated BGZF extra field");
        bool found = false;
        for (uint32_t p = 0; p + 4 <= xlen;) {
            const uint32_t slen = le16(&extra[p + 2]);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) { bsize = le16(&extra[p + 4]); found = true; break; }
            p += 4 + slen;
        }
        if (!found) return fail(LGIO_E_FORMAT, "gzip member without a BC subfield at %llu", (unsigned long long)addr);
    }
    const uint64_t total = (uint64_t)bsize + 1;
    if (total < 12ull + xlen + 8) return fail(LGIO_E_FORMAT, "bad BGZF block size at %llu", (unsigned long long)addr);
    br.addr = addr; br.head = 12 + xlen; br.clen = (uint32_t)(total - 12 - xlen - 8);
    next = addr + total;
    return 0;
}

// one block inflated into `out` (CRC checked); the message of a failure goes to `why` (fail()'s text is thread-local)
int inflate_block(int fd, const BlockRef& br, std::vector<uint8_t>& comp, std::vector<uint8_t>& out, std::string& why) {
    char msg[160];
    comp.resize((size_t)br.clen + 8);
    if (pread(fd, comp.data(), comp.size(), (off_t)(br.addr + br.head)) != (ssize_t)comp.size()) {
        snprintf(msg, sizeof msg, "truncated BGZF block at %llu", (unsigned long long)br.addr); why = msg; return LGIO_E_FORMAT;
    }
    const uint32_t isize = le32(comp.data() + br.clen + 4);
    if (isize > 65536) { snprintf(msg, sizeof msg, "BGZF block larger than 64 KiB at %llu", (unsigned long long)br.addr); why = msg; return LGIO_E_FORMAT; }
    out.resize(isize);
    if (!isize) return 0;
    const int zr = inflate_raw(comp.data(), br.clen, out.data(), isize, le32(comp.data() + br.clen));
    if (zr == 1) { snprintf(msg, sizeof msg, "inflate failed at block %llu", (unsigned long long)br.addr); why = msg; return LGIO_E_FORMAT; }
    if (zr == 2) { snprintf(msg, sizeof msg, "CRC mismatch in BGZF block %llu", (unsigned long long)br.addr); why = msg; return LGIO_E_FORMAT; }
    return 0;
}

}  // namespace

extern "C" void lgio_intervals_free(lgio_intervals* iv) {
    if (!iv) return;
    delete static_cast<IntervalsOwner*>(iv->owner_);
    memset(iv, 0, sizeof *iv);
}

// Round 5: the scan in SEGMENTS that start at record boundaries the index already names.  The serial part of the windowed
// scan below — walking record to record over blocks other threads inflated, one cold cache line per record — had become
// its floor once libdeflate took the inflation below it (4 M reads: 1.03 s whatever the thread count).  The BAI's linear
// index holds, per 16-kb window of the reference, the virtual offset of the first read overlapping it: each is a RECORD
// START, so the stretch of the file between two consecutive ones can be inflated AND walked by one thread, start to end,
// with nothing to wait for (the blocks are walked while they are in the cache of the core that inflated them).
// -> 0 ok, 1 not applicable (no anchors: the caller takes the windowed scan), negative error
static int ref_intervals_segments(lgio_bam* b, int tid, int threads, int fd, uint64_t first_voff, IntervalsOwner* o) {
    std::vector<uint64_t> anchors;
    for (uint64_t v : b->index[tid].linear) if (v > first_voff) anchors.push_back(v);
    std::sort(anchors.begin(), anchors.end());
    anchors.erase(std::unique(anchors.begin(), anchors.end()), anchors.end());
    anchors.insert(anchors.begin(), first_voff);
    if (anchors.size() < 2) return 1;
    const size_t n_seg = anchors.size();
    struct Seg { std::vector<int64_t> start, end; uint64_t bytes = 0; bool saw_other = false; };
    std::vector<Seg> segs(n_seg);
    std::atomic<size_t> take{0};
    std::atomic<size_t> stop_at{n_seg};          // the first segment in which another reference's record turned up: later ones hold none of ours
    std::atomic<int> err{0};
    std::mutex mu;
    std::string why;
    auto bad = [&](int code, const std::string& msg) { std::lock_guard<std::mutex> g(mu); if (!err.load()) { err = code; why = msg; } };
    auto work = [&] {
        std::vector<uint8_t> comp, data, rec;
        std::string msg;
        for (;;) {
            const size_t k = take.fetch_add(1);
            if (k >= n_seg || k > stop_at.load() || err.load()) return;
            Seg& sg = segs[k];
            const uint64_t v_end = k + 1 < n_seg ? anchors[k + 1] : ~0ull;
            uint64_t addr = anchors[k] >> 16, next = 0;
            size_t upos = (size_t)(anchors[k] & 0xFFFF);
            bool have = false, eof = false;
            // the next n bytes of the stream into dst (dst == NULL: skipped); false at the end of the file
            auto load = [&]() -> int {
                BlockRef br;
                const int rc = block_ref(fd, addr, br, next);
                if (rc == 1) { eof = true; return 1; }
                if (rc) { bad(rc, lgio_last_error()); return rc; }
                const int ri = inflate_block(fd, br, comp, data, msg);
                if (ri) { bad(ri, msg); return ri; }
                sg.bytes += br.head + br.clen + 8;
                have = true;
                return 0;
            };
            auto take_bytes = [&](uint8_t* dst, size_t n) -> int {       // 0 ok, 1 clean end before the first byte, 2 end inside, negative error
                size_t done = 0;
                while (done < n) {
                    if (!have || upos >= data.size()) {
                        if (have) { addr = next; upos = 0; }
                        const int rc = load();
                        if (rc == 1) return done ? 2 : 1;
                        if (rc) return rc;
                        if (upos > data.size()) { bad(LGIO_E_FORMAT, "virtual offset beyond its block"); return LGIO_E_FORMAT; }
                        continue;
                    }
                    const size_t t = std::min(n - done, data.size() - upos);
                    if (dst) memcpy(dst + done, data.data() + upos, t);
                    upos += t; done += t;
                }
                return 0;
            };
            for (;;) {
                // where the next record starts (a position at the end of a block belongs to the following block)
                const uint64_t here = !have ? anchors[k] : (upos >= data.size() ? next << 16 : (addr << 16) | upos);
                if (here >= v_end) break;
                uint8_t head[36];
                int rc = take_bytes(head, 4);
                if (rc == 1) break;                                  // the file ends here
                if (rc) { if (rc == 2) bad(LGIO_E_FORMAT, "file ends inside a record"); return; }
                const uint32_t bs = le32(head);
                if (bs < 32 || bs > (1u << 29)) { bad(LGIO_E_FORMAT, "implausible BAM record size"); return; }
                rc = take_bytes(head + 4, 32);
                if (rc) { if (rc > 0) bad(LGIO_E_FORMAT, "file ends inside a record"); return; }
                const uint8_t* r = head + 4;
                const int32_t rtid = (int32_t)le32(r);
                if (rtid != tid) {                                   // the reference is over (a sorted file): nothing of ours behind this
                    size_t cur = stop_at.load();
                    while (k < cur && !stop_at.compare_exchange_weak(cur, k)) {}
                    sg.saw_other = true;
                    break;
                }
                const int64_t pos = (int32_t)le32(r + 4);
                const uint32_t l_name = r[8], n_cigar = le16(r + 12), flag = le16(r + 14), l_seq = le32(r + 16);
                if (32ull + l_name + 4ull * n_cigar + (l_seq + 1ull) / 2 + l_seq > bs || l_name == 0) { bad(LGIO_E_FORMAT, "BAM record fields exceed the record"); return; }
                size_t left = bs - 32;
                if (!(flag & 4) && pos < MAX_POS) {
                    rec.resize((size_t)l_name + 4ull * n_cigar);
                    rc = take_bytes(rec.data(), rec.size());
                    if (rc) { if (rc > 0) bad(LGIO_E_FORMAT, "file ends inside a record"); return; }
                    left -= rec.size();
                    int64_t span = 0;
                    const uint8_t* c = rec.data() + l_name;
                    for (uint32_t q = 0; q < n_cigar; ++q) {
                        const uint32_t v = le32(c + 4 * q), op = v & 0xF;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
                    }
                    sg.start.push_back(pos);
                    sg.end.push_back(pos + (span > 0 ? span : 1));
                }
                rc = take_bytes(nullptr, left);
                if (rc) { if (rc > 0) bad(LGIO_E_FORMAT, "file ends inside a record"); return; }
            }
        }
    };
    {
        std::vector<std::thread> pool;
        const int nt = (int)std::min<size_t>((size_t)threads, n_seg);
        int started = 1;
        for (; started < nt; ++started) { try { pool.emplace_back(work); } catch (...) { break; } }
        work();
        for (std::thread& t : pool) t.join();
    }
    if (err.load()) return fail(err.load(), "%s", why.c_str());
    const size_t last = std::min(stop_at.load(), n_seg - 1);
    size_t total = 0;
    for (size_t k = 0; k <= last; ++k) total += segs[k].start.size();
    o->start.reserve(total); o->end.reserve(total);
    for (size_t k = 0; k <= last; ++k) {
        o->start.insert(o->start.end(), segs[k].start.begin(), segs[k].start.end());
        o->end.insert(o->end.end(), segs[k].end.begin(), segs[k].end.end());
        b->z.bytes_read += segs[k].bytes;
    }
    return 0;
}

static int ref_intervals_impl(lgio_bam* b, int tid, int threads, lgio_intervals* out) {
    if (!b || !out) return fail(LGIO_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    if (tid < 0 || (size_t)tid >= b->ref_names.size()) return fail(LGIO_E_ARG, "reference id %d out of range", tid);
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    IntervalsOwner* o = new IntervalsOwner();
    struct Drop { IntervalsOwner* p; ~Drop() { delete p; } } drop{o};
    auto finish = [&] {
        out->n = o->start.size(); out->start = o->start.data(); out->end = o->end.data(); out->owner_ = o;
        drop.p = nullptr;
        return LGIO_OK;
    };
    std::vector<Chunk> chunks;
    query_chunks(b->index[tid], 0, MAX_POS, chunks);
    if (chunks.empty()) return finish();
    const int fd = open(b->path.c_str(), O_RDONLY);
    if (fd < 0) return fail(LGIO_E_IO, "cannot open %s", b->path.c_str());
    struct Close { int fd; ~Close() { close(fd); } } closer{fd};
    if (!getenv("LGIO_SCAN_WINDOWS")) {                      // (LGIO_SCAN_WINDOWS=1: the windowed scan below, for comparisons and tests)
        const int rs = ref_intervals_segments(b, tid, threads, fd, chunks[0].beg, o);
        if (rs < 0) return rs;
        if (rs == 0) return finish();
    }
    // the reads of a reference are contiguous in a sorted file: from the first record the index points at until the
    // reference id changes.  Windows of WIN blocks: headers hopped over by the calling thread, blocks inflated by the worker
    // threads, records walked by the calling thread (a record may straddle blocks and windows: `stream` carries the
    // remainder).  Round 4: the walk of a window runs WHILE the workers inflate the next one (two buffer sets) — the walk
    // was ~40 % of the scan at 16 threads (4 M reads: 1.27 s, the CLI's first stage).
    const size_t WIN = 64 * (size_t)threads;
    uint64_t addr = chunks[0].beg >> 16;
    size_t skip = (size_t)(chunks[0].beg & 0xFFFF);
    struct Window {
        std::vector<BlockRef> refs;
        std::vector<std::vector<uint8_t>> bufs;
        std::vector<std::thread> pool;
        std::atomic<size_t> take{0};
        std::atomic<int> err{0};
        std::mutex mu;
        std::string why;
        bool launched = false;
    };
    Window win[2];
    win[0].bufs.resize(WIN); win[1].bufs.resize(WIN);
    std::vector<uint8_t> stream;
    bool done = false, eof = false;
    // one whole record (without its size word): 0 taken or skipped, 1 = the reference is over, negative error
    auto one_record = [&](const uint8_t* r, uint32_t bs) -> int {
        const int32_t rtid = (int32_t)le32(r);
        if (rtid != tid) { done = true; return 1; }
        const int64_t pos = (int32_t)le32(r + 4);
        const uint32_t l_name = r[8], n_cigar = le16(r + 12), flag = le16(r + 14), l_seq = le32(r + 16);
        if (32ull + l_name + 4ull * n_cigar + (l_seq + 1ull) / 2 + l_seq > bs || l_name == 0) return fail(LGIO_E_FORMAT, "BAM record fields exceed the record");
        if ((flag & 4) || pos >= MAX_POS) return 0;
        int64_t span = 0;
        const uint8_t* c = r + 32 + l_name;
        for (uint32_t k = 0; k < n_cigar; ++k) {
            const uint32_t v = le32(c + 4 * k), op = v & 0xF;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
        }
        o->start.push_back(pos);
        o->end.push_back(pos + (span > 0 ? span : 1));
        return 0;
    };
    // headers of the next window; 0 ok (possibly empty at the end of the file), negative error
    auto fill = [&](Window& w) -> int {
        w.refs.clear();
        while (!eof && w.refs.size() < WIN) {
            BlockRef br; uint64_t next = 0;
            const int rc = block_ref(fd, addr, br, next);
            if (rc == 1) { eof = true; break; }
            if (rc) return rc;
            w.refs.push_back(br);
            addr = next;
        }
        return LGIO_OK;
    };
    auto launch = [&](Window& w, bool with_caller) {
        w.take = 0; w.err = 0; w.why.clear();
        Window* const wp = &w;
        auto work = [wp, fd] {
            std::vector<uint8_t> comp;
            std::string msg;
            for (;;) {
                const size_t i = wp->take.fetch_add(1);
                if (i >= wp->refs.size() || wp->err.load()) return;
                const int rc = inflate_block(fd, wp->refs[i], comp, wp->bufs[i], msg);
                if (rc) { std::lock_guard<std::mutex> g(wp->mu); if (!wp->err.load()) { wp->err = rc; wp->why = msg; } return; }
            }
        };
        const size_t nt = std::min<size_t>((size_t)threads, w.refs.size());
        for (size_t t = with_caller ? 1 : 0; t < nt; ++t) w.pool.emplace_back(work);
        if (with_caller) work();                             // (the first window: nothing to walk yet)
        w.launched = true;
    };
    auto join = [&](Window& w) { for (std::thread& t : w.pool) t.join(); w.pool.clear(); w.launched = false; };
    struct JoinAll { Window* w; ~JoinAll() { for (int k = 0; k < 2; ++k) { for (std::thread& t : w[k].pool) t.join(); w[k].pool.clear(); } } } join_all{win};   // (error returns)
    int cur = 0;
    { const int rc = fill(win[0]); if (rc) return rc; }
    if (!win[0].refs.empty()) launch(win[0], true);
    while (!done && !win[cur].refs.empty()) {
        Window& w = win[cur];
        join(w);
        if (w.err.load()) return fail(w.err.load(), "%s", w.why.c_str());
        Window& nx = win[cur ^ 1];
        { const int rc = fill(nx); if (rc) return rc; }
        if (!nx.refs.empty()) launch(nx, threads == 1);       // inflated while this window's records are walked (one thread: before)
        const std::vector<BlockRef>& refs = w.refs;
        std::vector<std::vector<uint8_t>>& bufs = w.bufs;
        for (size_t i = 0; i < refs.size() && !done; ++i) {
            b->z.bytes_read += refs[i].head + refs[i].clen + 8;
            if (skip > bufs[i].size()) return fail(LGIO_E_FORMAT, "virtual offset beyond its block");
            const uint8_t* d = bufs[i].data() + skip;
            size_t n = bufs[i].size() - skip;
            skip = 0;
            // records are walked where the block lies; only one that straddles blocks is assembled in `stream`
            while (n && !done) {
                if (!stream.empty()) {
                    size_t need = stream.size() < 4 ? 4 - stream.size() : 0;
                    if (!need) {
                        const uint32_t bs = le32(stream.data());
                        if (bs < 32 || bs > (1u << 29)) return fail(LGIO_E_FORMAT, "implausible BAM record size %u", bs);
                        need = 4 + (size_t)bs - stream.size();
                    }
                    const size_t take = std::min(need, n);
                    stream.insert(stream.end(), d, d + take);
                    d += take; n -= take;
                    if (take == need && stream.size() >= 4) {
                        // the size word may have been completed by this very piece: it is checked before it is believed
                        // (a record size of 0 would otherwise satisfy the completion test with nothing behind the word)
                        const uint32_t bs = le32(stream.data());
                        if (bs < 32 || bs > (1u << 29)) return fail(LGIO_E_FORMAT, "implausible BAM record size %u", bs);
                    }
                    if (take == need && stream.size() >= 4 && stream.size() == 4 + (size_t)le32(stream.data())) {
                        const int rc = one_record(stream.data() + 4, le32(stream.data()));
                        if (rc < 0) return rc;
                        stream.clear();
                    }
                    continue;
                }
                if (n < 4) { stream.assign(d, d + n); break; }
                const uint32_t bs = le32(d);
                if (bs < 32 || bs > (1u << 29)) return fail(LGIO_E_FORMAT, "implausible BAM record size %u", bs);
                if (4 + (size_t)bs > n) { stream.assign(d, d + n); break; }
                const int rc = one_record(d + 4, bs);
                if (rc < 0) return rc;
                d += 4 + (size_t)bs; n -= 4 + (size_t)bs;
            }
        }
        cur ^= 1;
    }
    if (!done && !stream.empty()) return fail(LGIO_E_FORMAT, "file ends inside a record");
    return finish();
}
extern "C" int lgio_bam_ref_intervals(lgio_bam* b, int tid, int threads, lgio_intervals* out) {
    return guarded([&] { return ref_intervals_impl(b, tid, threads, out); });
}
