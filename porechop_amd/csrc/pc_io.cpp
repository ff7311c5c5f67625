// pc_io.cpp -- host ingest of FASTA/FASTQ(.gz) into the packed read arena the scan kernels consume
// (the row after the hot path).  Pure host code, no GPU involved.
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

int pc_readset_load(const char *path, pc_readset **out)
{
    if (!path || !out) return PC_ERR_BAD_ARG;
    pc_readset *rs = new pc_readset();
    *out = rs;
    std::vector<char> data;
    if (!slurp(path, data, rs->error)) return PC_ERR_BAD_ARG;
    const char first = data.empty() ? '\0' : data[0];
    if (first != '>' && first != '@') { rs->error = "File is neither FASTA or FASTQ"; return PC_ERR_BAD_ARG; }
    rs->fastq = (first == '@');
    rs->arena.reserve(data.size() / (rs->fastq ? 2 : 1) + 128);
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
    rs->arena.insert(rs->arena.end(), 64, 'N');           // the kernels fetch a dword at a time
    return PC_OK;
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

}  // extern "C"
