// pc_io.cpp -- host ingest of FASTA/FASTQ(.gz) into the packed read arena the scan kernels consume
// (the row after the hot path) and the output writer (8f-3).  Pure host
// code, no GPU involved.
//
// Mirrors what the reference does between the file and the first alignment:
//   porechop/misc.py:60-81    get_compression_type  (gzip by magic bytes; bz2/zip refused)
//   porechop/misc.py:84-105   get_sequence_file_type ('>' FASTA, '@' FASTQ)
//   porechop/misc.py:123-148  load_fasta  (multi-line records, blank lines skipped, strip())
//   porechop/misc.py:151-168  load_fastq  (4 stripped lines per record)
//   porechop/nanopore_read.py:23-35  NanoporeRead.__init__ (upper(); U->T when U's outnumber T's;
//                                    qualities padded with '+')
// but produces one contiguous arena (1 byte per base, reads back to back, 64 bytes of 'N' padding
// at the end) plus offset/length tables -- exactly the inputs of pc_align_batch_host /
// pc_scan_device -- instead of millions of Python tuples.
//
// The writer (pc_readset_write) is the byte-level half of porechop/nanopore_read.py:97-147
// (get_fasta / get_fastq) and porechop/porechop.py:607-734 (output_reads): the caller decides WHICH
// pieces of which reads go to which file (trim amounts, split points, barcode bins -- integer
// arrays), this code formats them straight from the arena.
#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/porechop_amd.h"

struct pc_readset {
    bool fastq = false;
    std::vector<char> arena;
    std::vector<int64_t> off;
    std::vector<int32_t> len;
    std::vector<uint8_t> rna;
    std::vector<std::string> names;      // full header without the leading marker
    std::vector<std::string> quals;      // FASTQ only (padded with '+' to the sequence length)
    std::vector<int32_t> file_index;     // which input file a read came from (pc_readset_load_many)
    std::string error;
};

namespace {

bool slurp(const char *path, std::vector<char> &data, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("could not find ") + path; return false; }
    unsigned char magic[4] = {0, 0, 0, 0};
    const size_t got = fread(magic, 1, 4, f);
    fclose(f);
    if (got >= 3 && magic[0] == 0x42 && magic[1] == 0x5a && magic[2] == 0x68) { err = "cannot use bzip2 format - use gzip instead"; return false; }
    if (got >= 4 && magic[0] == 0x50 && magic[1] == 0x4b && magic[2] == 0x03 && magic[3] == 0x04) { err = "cannot use zip format - use gzip instead"; return false; }
    // gzopen reads plain files transparently, but the reference decides by magic bytes: do the same
    const bool gz = got >= 3 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 0x08;
    if (gz) {
        gzFile g = gzopen(path, "rb");
        if (!g) { err = std::string("could not open ") + path; return false; }
        gzbuffer(g, 1 << 20);
        std::vector<char> buf(1 << 22);
        for (;;) {
            const int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) { err = "gzip stream error"; gzclose(g); return false; }
            if (n == 0) break;
            data.insert(data.end(), buf.begin(), buf.begin() + n);
        }
        gzclose(g);
    } else {
        f = fopen(path, "rb");
        if (!f) { err = std::string("could not open ") + path; return false; }
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        data.resize(sz > 0 ? (size_t)sz : 0);
        if (sz > 0 && fread(data.data(), 1, (size_t)sz, f) != (size_t)sz) { err = "short read"; fclose(f); return false; }
        fclose(f);
    }
    return true;
}

// Python str.strip(): ASCII whitespace on both sides
inline void strip(const char *&b, const char *&e)
{
    while (b < e && isspace((unsigned char)*b)) ++b;
    while (e > b && isspace((unsigned char)e[-1])) --e;
}

struct Lines {
    const char *p, *end;
    bool next(const char *&b, const char *&e)
    {
        if (p >= end) return false;
        b = p;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    }
};

struct UpperTable {
    unsigned char t[256];
    UpperTable() { for (int i = 0; i < 256; ++i) t[i] = (unsigned char)((i >= 'a' && i <= 'z') ? i - 32 : i); }
};
const UpperTable kUpper;

void add_read(pc_readset *rs, const char *name_b, const char *name_e, const char *seq_b, const char *seq_e,
              const char *q_b, const char *q_e)
{
    // NanoporeRead.__init__: upper(); if count('U') > count('T'): U -> T
    const size_t n = (size_t)(seq_e - seq_b);
    const size_t o = rs->arena.size();
    rs->arena.resize(o + n);
    unsigned char *dst = (unsigned char *)rs->arena.data() + o;
    size_t nu = 0, nt = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = kUpper.t[(unsigned char)seq_b[i]];
        dst[i] = c;
        nu += (c == 'U'); nt += (c == 'T');
    }
    const bool rna = nu > nt;
    if (rna) for (size_t i = 0; i < n; ++i) if (dst[i] == 'U') dst[i] = 'T';
    rs->off.push_back((int64_t)o);
    rs->len.push_back((int32_t)n);
    rs->rna.push_back(rna ? 1 : 0);
    rs->names.emplace_back(name_b, name_e);
    if (rs->fastq) {
        rs->quals.emplace_back(q_b, q_e);
        std::string &q = rs->quals.back();
        if (q.size() < n) q.append(n - q.size(), '+');
    }
}

}  // namespace

extern "C" {

// parse one file into rs (reads appended); rs->fastq is set by the first file
static int load_into(pc_readset *rs, const char *path, int32_t file_index, bool first)
{
    std::vector<char> data;
    if (!slurp(path, data, rs->error)) return PC_ERR_BAD_ARG;
    const char first_char = data.empty() ? '\0' : data[0];
    if (first_char != '>' && first_char != '@') { rs->error = "File is neither FASTA or FASTQ"; return PC_ERR_BAD_ARG; }
    const bool fastq = (first_char == '@');
    if (first) rs->fastq = fastq;
    else if (fastq != rs->fastq) { rs->error = std::string(path) + " is not of the same type as the files before it"; return PC_ERR_BAD_ARG; }
    rs->arena.reserve(rs->arena.size() + data.size() / (rs->fastq ? 2 : 1) + 128);
    const size_t before = rs->off.size();
    Lines ln{data.data(), data.data() + data.size()};
    const char *b, *e;
    if (rs->fastq) {
        while (ln.next(b, e)) {
            strip(b, e);
            const char *nb = b < e ? b + 1 : b;           // line.strip()[1:]
            const char *sb, *se, *pb, *pe, *qb, *qe;
            if (nb == e || !ln.next(sb, se) || !ln.next(pb, pe) || !ln.next(qb, qe)) {
                rs->error = std::string(path) + " could not be parsed - is it formatted correctly?";
                return PC_ERR_BAD_ARG;
            }
            strip(sb, se); strip(qb, qe);
            add_read(rs, nb, e, sb, se, qb, qe);
        }
    } else {
        std::string name, seq;
        bool have = false;
        while (ln.next(b, e)) {
            strip(b, e);
            if (b == e) continue;
            if (*b == '>') {
                if (have && !name.empty()) add_read(rs, name.data(), name.data() + name.size(), seq.data(), seq.data() + seq.size(), nullptr, nullptr);
                seq.clear();
                name.assign(b + 1, e);
                have = true;
            } else {
                seq.append(b, e);
            }
        }
        if (have && !name.empty()) add_read(rs, name.data(), name.data() + name.size(), seq.data(), seq.data() + seq.size(), nullptr, nullptr);
    }
    rs->file_index.resize(rs->off.size(), file_index);
    (void)before;
    return PC_OK;
}

int pc_readset_load_many(const char *const *paths, int npaths, pc_readset **out)
{
    if (!paths || npaths < 1 || !out) return PC_ERR_BAD_ARG;
    pc_readset *rs = new pc_readset();
    *out = rs;
    for (int i = 0; i < npaths; ++i) {
        const int rc = load_into(rs, paths[i], i, i == 0);
        if (rc) return rc;
    }
    rs->arena.insert(rs->arena.end(), 64, 'N');           // the kernels fetch a dword at a time
    return PC_OK;
}

int pc_readset_load(const char *path, pc_readset **out)
{
    if (!path || !out) return PC_ERR_BAD_ARG;
    return pc_readset_load_many(&path, 1, out);
}

void pc_readset_free(pc_readset *rs) { delete rs; }
const char *pc_readset_error(const pc_readset *rs) { return rs ? rs->error.c_str() : "null readset"; }
int64_t pc_readset_count(const pc_readset *rs) { return rs ? (int64_t)rs->off.size() : 0; }
int pc_readset_is_fastq(const pc_readset *rs) { return rs && rs->fastq ? 1 : 0; }
const char *pc_readset_arena(const pc_readset *rs, int64_t *bytes)
{
    if (!rs) return nullptr;
    if (bytes) *bytes = (int64_t)rs->arena.size();
    return rs->arena.data();
}
const int64_t *pc_readset_offsets(const pc_readset *rs) { return rs ? rs->off.data() : nullptr; }
const int32_t *pc_readset_lengths(const pc_readset *rs) { return rs ? rs->len.data() : nullptr; }
const char *pc_readset_name(const pc_readset *rs, int64_t i)
{
    return (rs && i >= 0 && i < (int64_t)rs->names.size()) ? rs->names[(size_t)i].c_str() : nullptr;
}
const char *pc_readset_quals(const pc_readset *rs, int64_t i)
{
    return (rs && rs->fastq && i >= 0 && i < (int64_t)rs->quals.size()) ? rs->quals[(size_t)i].c_str() : nullptr;
}
int pc_readset_is_rna(const pc_readset *rs, int64_t i)
{
    return (rs && i >= 0 && i < (int64_t)rs->rna.size()) ? rs->rna[(size_t)i] : 0;
}
const int32_t *pc_readset_file_index(const pc_readset *rs) { return rs ? rs->file_index.data() : nullptr; }

int pc_readset_write(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                     const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                     const char *const *file_paths, int fastq, int64_t *bytes_written)
{
    if (!rs || npieces < 0 || nfiles < 0 || (npieces > 0 && (!piece_read || !piece_start || !piece_len || !piece_file || !file_paths)))
        return PC_ERR_BAD_ARG;
    std::vector<FILE *> files((size_t)nfiles, nullptr);
    std::vector<std::vector<char>> bufs((size_t)nfiles);
    int64_t total = 0;
    int rc = PC_OK;
    auto flush = [&](int f) {
        std::vector<char> &b = bufs[(size_t)f];
        if (b.empty()) return;
        if (fwrite(b.data(), 1, b.size(), files[(size_t)f]) != b.size()) rc = PC_ERR_BAD_ARG;
        total += (int64_t)b.size();
        b.clear();
    };
    const int64_t nreads = (int64_t)rs->off.size();
    for (int64_t k = 0; k < npieces && rc == PC_OK; ++k) {
        const int64_t r = piece_read[k];
        const int f = piece_file[k];
        if (r < 0 || r >= nreads || f < 0 || f >= nfiles) { rc = PC_ERR_BAD_ARG; break; }
        const int64_t n = rs->len[(size_t)r];
        const int64_t st = piece_start[k], ln = piece_len[k];
        if (st < 0 || ln < 0 || st + ln > n) { rc = PC_ERR_BAD_ARG; break; }
        if (!files[(size_t)f]) {                                   // opened on first use, like the reference's bins
            const char *path = file_paths[f];
            files[(size_t)f] = (path[0] == '-' && path[1] == '\0') ? stdout : fopen(path, "wb");
            if (!files[(size_t)f]) { rc = PC_ERR_BAD_ARG; break; }
            bufs[(size_t)f].reserve(1 << 22);
        }
        std::vector<char> &b = bufs[(size_t)f];
        // header: add_number_to_read_name (nanopore_read.py:494-498): "_<k>" before the first space, or at the end
        b.push_back(fastq ? '@' : '>');
        const std::string &name = rs->names[(size_t)r];
        const int number = piece_number ? piece_number[k] : 0;
        if (number > 0) {
            char tag[24];
            const int tl = snprintf(tag, sizeof tag, "_%d", number);
            const size_t sp = name.find(' ');
            if (sp == std::string::npos) { b.insert(b.end(), name.begin(), name.end()); b.insert(b.end(), tag, tag + tl); }
            else { b.insert(b.end(), name.begin(), name.begin() + (long)sp); b.insert(b.end(), tag, tag + tl); b.insert(b.end(), name.begin() + (long)sp, name.end()); }
        } else {
            b.insert(b.end(), name.begin(), name.end());
        }
        b.push_back('\n');
        const char *seq = rs->arena.data() + rs->off[(size_t)r] + st;
        const bool rna = rs->rna[(size_t)r] != 0;
        const size_t seq_at = b.size();
        if (fastq) {
            b.insert(b.end(), seq, seq + ln);
            b.push_back('\n'); b.push_back('+'); b.push_back('\n');
            if (rs->fastq) { const std::string &q = rs->quals[(size_t)r]; b.insert(b.end(), q.begin() + st, q.begin() + st + ln); }
            else b.insert(b.end(), (size_t)ln, '+');             // FASTA input: NanoporeRead pads the empty qualities with '+'
            b.push_back('\n');
            if (rna) for (size_t i = seq_at; i < seq_at + (size_t)ln; ++i) if (b[i] == 'T') b[i] = 'U';
        } else {
            // add_line_breaks_to_sequence(seq, 70): every line, the last included, ends in '\n'
            if (ln == 0) b.push_back('\n');
            for (int64_t pos = 0; pos < ln; pos += 70) {
                const int64_t w = ln - pos < 70 ? ln - pos : 70;
                b.insert(b.end(), seq + pos, seq + pos + w);
                b.push_back('\n');
            }
            if (rna) for (size_t i = seq_at; i < b.size(); ++i) if (b[i] == 'T') b[i] = 'U';
        }
        if (b.size() >= ((size_t)1 << 22)) flush(f);
    }
    for (int f = 0; f < nfiles; ++f) {
        if (!files[(size_t)f]) continue;
        if (rc == PC_OK) flush(f);
        if (files[(size_t)f] == stdout) fflush(stdout); else if (fclose(files[(size_t)f]) != 0) rc = PC_ERR_BAD_ARG;
    }
    if (bytes_written) *bytes_written = total;
    return rc;
}

}  // extern "C"
