// pc_gz.h -- gzip at the speed of the rest of the host path.
//
// The reference reads .gz input through Python's gzip module (porechop/misc.py:60-81,151-168: one thread) and
// compresses its output by shelling out to `pigz -p <threads>`, or to gzip (porechop/porechop.py:640-651,685-729).
// Here:
//   * OUTPUT is deflated by all cores as a chain of independent gzip members of <= 65 280 input bytes each, every one
//     carrying its compressed size in a 'BC' extra subfield -- the BGZF layout of the SAM specification (section 4.1):
//     any gunzip / zlib / Python gzip reads it as an ordinary multi-member gzip file, and a reader that knows the
//     subfield (ours below, bgzip, htslib) can inflate the members in parallel.
//   * INPUT whose members carry that subfield is inflated by all cores; any other gzip file (one big member: gzip, pigz)
//     is inflated by one producer thread that runs AHEAD of the parser (pc_gzstream in pc_io.cpp), so that inflating
//     block k+1 overlaps parsing, scanning and writing block k.
// DEFLATE itself comes from libdeflate (dlopen'ed: 2-3 x zlib's speed both ways) when the machine has it, else zlib.
#pragma once
#include <zlib.h>

#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace pcz {

constexpr size_t kBlockIn = 0xff00;          // input bytes per member (bgzip's choice: a stored block still fits 64 KB)
constexpr size_t kBlockMax = 0x10000;        // a member is at most 64 KB (BSIZE is 16 bits)
constexpr size_t kHeader = 18, kTrailer = 8;

// the empty member every BGZF file ends with (SAM specification 4.1.2)
static const unsigned char kEofBlock[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00,
                                            0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};

struct LibDeflate {
    void *(*alloc_compressor)(int) = nullptr;
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*free_compressor)(void *) = nullptr;
    void *(*alloc_decompressor)(void) = nullptr;
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_decompressor)(void *) = nullptr;
    uint32_t (*crc32_)(uint32_t, const void *, size_t) = nullptr;
    // (optional: the one-shot route of the streamed reader, pc_io.cpp oneshot_member) -> 0 done, 1 bad data, 3 out of room
    int (*deflate_decompress_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        const char *off = getenv("PC_NO_LIBDEFLATE");
        if (off && *off && *off != '0') return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_compressor = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        deflate_compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_compressor = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        alloc_decompressor = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        deflate_decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_decompressor = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc32_ = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        deflate_decompress_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_deflate_decompress_ex");
        ok = alloc_compressor && deflate_compress && free_compressor && alloc_decompressor && deflate_decompress &&
             free_decompressor && crc32_;
    }
};
inline const LibDeflate &libdeflate() { static const LibDeflate l; return l; }

// level <= 0: the default.  libdeflate's level 3 already makes SMALLER files than zlib's level 6 -- what gzip and pigz, the
// reference's compressors, use by default -- at five times the speed (measured on FASTQ with random qualities, 8 cores:
// 0.516 of the input at 503 MB/s against 0.539 at 108 MB/s); without libdeflate the default is zlib's own 6.
inline int default_level(int level)
{
    if (level > 9) return 9;
    if (level >= 1) return level;
    return libdeflate().ok ? 3 : 6;
}

inline uint32_t crc_of(const void *p, size_t n)
{
    const LibDeflate &l = libdeflate();
    if (l.ok) return l.crc32_(0, p, n);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)p, (uInt)n);
}

// running CRC-32 (start from 0): libdeflate's runs at >10 GB/s, zlib 1.2.11's at 0.7 -- a fifth of what its own inflate takes
inline uint32_t crc_update(uint32_t crc, const void *p, size_t n)
{
    const LibDeflate &l = libdeflate();
    if (l.ok) return l.crc32_(crc, p, n);
    const unsigned char *q = (const unsigned char *)p;
    uLong c = crc;
    while (n) { const uInt k = (uInt)(n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n); c = crc32(c, q, k); q += k; n -= k; }
    return (uint32_t)c;
}

inline void put16(unsigned char *p, unsigned v) { p[0] = (unsigned char)(v & 0xff); p[1] = (unsigned char)(v >> 8); }
inline void put32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); p[2] = (unsigned char)(v >> 16); p[3] = (unsigned char)(v >> 24); }
inline uint32_t get32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// One thread's compressor: appends members to a byte vector.
class Deflater {
    void *ld_ = nullptr;
    z_stream zs_;
    bool z_ok_ = false;
    int level_;
public:
    explicit Deflater(int level) : level_(level < 1 ? 1 : (level > 9 ? 9 : level))
    {
        const LibDeflate &l = libdeflate();
        if (l.ok) ld_ = l.alloc_compressor(level_);
        if (!ld_) {
            memset(&zs_, 0, sizeof zs_);
            z_ok_ = deflateInit2(&zs_, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        }
    }
    Deflater(const Deflater &) = delete;
    ~Deflater()
    {
        if (ld_) libdeflate().free_compressor(ld_);
        if (z_ok_) deflateEnd(&zs_);
    }
    bool usable() const { return ld_ || z_ok_; }
    // one member from at most kBlockIn bytes
    bool member(const char *in, size_t n, std::vector<char> &out)
    {
        if (n > kBlockIn) return false;
        const size_t at = out.size();
        out.resize(at + kBlockMax);
        unsigned char *o = (unsigned char *)out.data() + at;
        const size_t room = kBlockMax - kHeader - kTrailer;
        size_t z = 0;
        if (ld_) {
            z = libdeflate().deflate_compress(ld_, in, n, o + kHeader, room);
        } else if (z_ok_) {
            deflateReset(&zs_);
            zs_.next_in = (Bytef *)in; zs_.avail_in = (uInt)n;
            zs_.next_out = o + kHeader; zs_.avail_out = (uInt)room;
            if (deflate(&zs_, Z_FINISH) == Z_STREAM_END) z = room - zs_.avail_out;
        }
        if (z == 0) {
            // did not fit (incompressible bytes): one stored block -- 5 bytes of framing, always fits for n <= kBlockIn
            unsigned char *d = o + kHeader;
            d[0] = 1; put16(d + 1, (unsigned)n); put16(d + 3, (unsigned)(~n & 0xffffu));
            memcpy(d + 5, in, n);
            z = n + 5;
        }
        const size_t total = kHeader + z + kTrailer;
        o[0] = 0x1f; o[1] = 0x8b; o[2] = 8; o[3] = 4; put32(o + 4, 0); o[8] = 0; o[9] = 0xff;
        put16(o + 10, 6); o[12] = 'B'; o[13] = 'C'; put16(o + 14, 2); put16(o + 16, (unsigned)(total - 1));
        put32(o + kHeader + z, crc_of(in, n));
        put32(o + kHeader + z + 4, (uint32_t)n);
        out.resize(at + total);
        return true;
    }
    bool append(const char *in, size_t n, std::vector<char> &out)
    {
        for (size_t i = 0; i < n; i += kBlockIn)
            if (!member(in + i, n - i < kBlockIn ? n - i : kBlockIn, out)) return false;
        return true;
    }
};

// A member that carries its own size: header with FEXTRA and a 'BC' subfield of two bytes.  -> total member bytes, or 0.
inline size_t sized_member(const unsigned char *p, size_t avail, size_t *payload_off)
{
    if (avail < kHeader + kTrailer || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    if (p[3] & ~4u) return 0;                               // name / comment / header crc: not what bgzip or we write
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (12 + xlen + kTrailer > avail) return 0;
    size_t at = 12, total = 0;
    while (at + 4 <= 12 + xlen) {
        const size_t slen = (size_t)p[at + 2] | ((size_t)p[at + 3] << 8);
        if (p[at] == 'B' && p[at + 1] == 'C' && slen == 2 && at + 6 <= 12 + xlen) total = ((size_t)p[at + 4] | ((size_t)p[at + 5] << 8)) + 1;
        at += 4 + slen;
    }
    if (total < 12 + xlen + kTrailer || total > avail) return 0;
    // ISIZE (the member's last four bytes) is what the readers size their buffers from BEFORE anything is inflated: a member
    // of this layout holds at most 64 KiB of data (BSIZE is 16 bits, and neither bgzip nor this package puts more than 65 280
    // input bytes into one); a trailer that claims more is damage or forgery -- such a file is not "sized" and takes the
    // ordinary gzip route, which inflates it member by member and reports what is wrong with it
    {
        const unsigned char *t = p + total - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        if (isize > 65536) return 0;
    }
    if (payload_off) *payload_off = 12 + xlen;
    return total;
}

// The header of the gzip member at p (RFC 1952): -> where its deflate data starts, 0 = no member header here (or cut off).
// The readers below inflate members RAW and check CRC-32 and ISIZE themselves (crc_update): zlib's own gzip wrapper spends a
// fifth of its time in its table-driven CRC.
inline size_t gzip_header_len(const unsigned char *p, size_t avail)
{
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    size_t at = 10;
    if (p[3] & 4) { if (at + 2 > avail) return 0; at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8)); }
    if (p[3] & 8) { while (at < avail && p[at]) ++at; ++at; }
    if (p[3] & 16) { while (at < avail && p[at]) ++at; ++at; }
    if (p[3] & 2) at += 2;
    return at + kTrailer <= avail ? at : 0;
}

class Inflater {
    void *ld_ = nullptr;
    z_stream zs_;
    bool z_ok_ = false;
public:
    Inflater()
    {
        const LibDeflate &l = libdeflate();
        if (l.ok) ld_ = l.alloc_decompressor();
        if (!ld_) { memset(&zs_, 0, sizeof zs_); z_ok_ = inflateInit2(&zs_, -15) == Z_OK; }
    }
    Inflater(const Inflater &) = delete;
    ~Inflater()
    {
        if (ld_) libdeflate().free_decompressor(ld_);
        if (z_ok_) inflateEnd(&zs_);
    }
    // raw deflate data -> exactly n_out bytes, checked against the member's crc
    bool raw(const unsigned char *in, size_t n_in, char *out, size_t n_out, uint32_t crc)
    {
        if (n_out == 0) return true;
        if (ld_) {
            if (libdeflate().deflate_decompress(ld_, in, n_in, out, n_out, nullptr) != 0) return false;
        } else if (z_ok_) {
            inflateReset(&zs_);
            zs_.next_in = (Bytef *)in; zs_.avail_in = (uInt)n_in;
            zs_.next_out = (Bytef *)out; zs_.avail_out = (uInt)n_out;
            if (inflate(&zs_, Z_FINISH) != Z_STREAM_END || zs_.avail_out != 0) return false;
        } else {
            return false;
        }
        return crc_of(out, n_out) == crc;
    }
};

}  // namespace pcz
