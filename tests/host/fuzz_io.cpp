// fuzz_io.cpp -- the host ingest / output / gzip code (porechop_amd/csrc/pc_io.cpp, pc_gz.h) under AddressSanitizer and
// UndefinedBehaviorSanitizer: test infrastructure, built by tests/test_io_sanitized.py with g++ from the two sources (the
// file has no HIP in it), never part of the library.
//
//   fuzz_io <workdir> <seed> <damaged files per layout>
//
// 1. A FASTQ file with irregular content (empty reads, lower case, U's, long names, CRLF-free) is written, compressed three
//    ways -- sized members (pc_gzip_file), ONE member (pigz style), several ordinary members back to back (`cat`) -- and read
//    back whole (pc_readset_load) and as a stream of small blocks (pc_gzstream_*): every route must give the plain file's reads.
// 2. The reads are written back plain and compressed (pc_readset_write, pc_readset_compress + pc_gzimage_write +
//    pc_gz_finish); the compressed output, read again, must be the same reads.
// 3. Each compressed layout and the plain file are DAMAGED (cut short, bytes flipped, ranges zeroed, garbage appended, a
//    chunk duplicated) and pushed through every reader: any return code is fine, a sanitizer report or a crash is not.
//    What the reference does with such files (porechop/misc.py:60-168: Python's gzip raises) is out of reach of a byte-level
//    comparison; the runner falls back to pc_readset_load, whose messages tests/test_ingest.py pins.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <random>
#include <string>
#include <vector>

#include "../../include/porechop_amd.h"

namespace {

std::mt19937_64 rng;
int failures = 0;

#define CHECK(cond, ...) do { if (!(cond)) { ++failures; fprintf(stderr, "CHECK failed %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

std::string slurp(const std::string &p)
{
    std::string s;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

void spit(const std::string &p, const std::string &s)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) { perror(p.c_str()); exit(2); }
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
}

struct Reads {
    std::vector<std::string> name, seq, qual;
    bool operator==(const Reads &o) const { return name == o.name && seq == o.seq && qual == o.qual; }
};

Reads reads_of(const pc_readset *rs)
{
    Reads r;
    const int64_t n = pc_readset_count(rs);
    int64_t bytes = 0;
    const char *arena = pc_readset_arena(rs, &bytes);
    const int64_t *off = pc_readset_offsets(rs);
    const int32_t *len = pc_readset_lengths(rs);
    for (int64_t i = 0; i < n; ++i) {
        r.name.push_back(pc_readset_name(rs, i));
        r.seq.emplace_back(arena + off[i], (size_t)len[i]);
        const char *q = pc_readset_quals(rs, i);
        r.qual.emplace_back(q ? std::string(q, (size_t)len[i]) : std::string());
    }
    return r;
}

void append(Reads &a, const Reads &b)
{
    a.name.insert(a.name.end(), b.name.begin(), b.name.end());
    a.seq.insert(a.seq.end(), b.seq.begin(), b.seq.end());
    a.qual.insert(a.qual.end(), b.qual.begin(), b.qual.end());
}

// -> rc of the load; *out filled when it succeeded
int load_whole(const std::string &path, Reads *out)
{
    pc_readset *rs = nullptr;
    const int rc = pc_readset_load(path.c_str(), &rs);
    if (rc == PC_OK && out) *out = reads_of(rs);
    if (rs) { (void)pc_readset_error(rs); pc_readset_free(rs); }
    return rc;
}

int load_stream(const std::string &path, int64_t target, int64_t min_reads, Reads *out)
{
    pc_gzstream *s = nullptr;
    int rc = pc_gzstream_open(path.c_str(), &s);
    if (rc != PC_OK) return rc;
    Reads all;
    for (int guard = 0; guard < 100000; ++guard) {
        pc_readset *rs = nullptr;
        int eof = 0;
        rc = pc_gzstream_next(s, target, min_reads, &rs, &eof);
        if (rc != PC_OK) break;
        if (rs) { append(all, reads_of(rs)); pc_readset_free(rs); }
        if (eof || !rs) break;
        min_reads = 0;
    }
    pc_gzstream_close(s);
    if (rc == PC_OK && out) *out = all;
    return rc;
}

int load_segments(const std::string &path, int64_t target, Reads *out)
{
    Reads all;
    int64_t at = 0;
    int rc = PC_OK;
    for (int guard = 0; guard < 100000; ++guard) {
        pc_readset *rs = nullptr;
        int64_t next = at;
        rc = pc_readset_load_segment(path.c_str(), at, target, &next, &rs);
        if (rc != PC_OK) { if (rs) pc_readset_free(rs); break; }
        const int64_t n = rs ? pc_readset_count(rs) : 0;
        if (rs) { append(all, reads_of(rs)); pc_readset_free(rs); }
        if (next <= at || n == 0) break;
        at = next;
    }
    if (rc == PC_OK && out) *out = all;
    return rc;
}

std::string make_fastq(size_t nreads, std::vector<size_t> *record_starts)
{
    static const char *alpha[] = {"ACGT", "ACGT", "ACGT", "acgt", "ACGU", "ACGTN", "ACGTRYKM-"};
    std::string s;
    for (size_t i = 0; i < nreads; ++i) {
        record_starts->push_back(s.size());
        size_t len = (size_t)(rng() % 3000);
        if (rng() % 50 == 0) len = 0;
        if (rng() % 40 == 0) len = 20000 + rng() % 60000;
        const char *al = alpha[rng() % 7];
        const size_t na = strlen(al);
        s += "@read" + std::to_string(i);
        if (rng() % 3 == 0) s += " runid=" + std::string(rng() % 64, 'x') + " ch=" + std::to_string(rng() % 512);
        s += '\n';
        for (size_t k = 0; k < len; ++k) s += al[rng() % na];
        s += "\n+\n";
        for (size_t k = 0; k < len; ++k) s += (char)(33 + rng() % 60);        // '@' and '+' among them, also as a line's first byte
        s += '\n';
    }
    record_starts->push_back(s.size());
    return s;
}

std::string damage(const std::string &good)
{
    std::string s = good;
    const int kinds = 1 + (int)(rng() % 3);
    for (int k = 0; k < kinds && !s.empty(); ++k) {
        switch (rng() % 6) {
            case 0: s.resize(rng() % s.size()); break;                                                  // cut short
            case 1: for (int j = 0, n = 1 + (int)(rng() % 8); j < n; ++j) s[rng() % s.size()] ^= (char)(1 << (rng() % 8)); break;
            case 2: { const size_t a = rng() % s.size(), n = std::min<size_t>(s.size() - a, rng() % 5000); memset(&s[a], 0, n); break; }
            case 3: for (int j = 0, n = (int)(rng() % 3000); j < n; ++j) s += (char)rng(); break;       // garbage after the end
            case 4: { const size_t a = rng() % s.size(), n = std::min<size_t>(s.size() - a, 1 + rng() % 70000); s.insert(a, s.substr(a, n)); break; }
            case 5: { const size_t a = rng() % s.size(), n = std::min<size_t>(s.size() - a, 1 + rng() % 70000); s.erase(a, n); break; }
        }
    }
    return s;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: fuzz_io <workdir> <seed> <damaged files per layout>\n"); return 2; }
    const std::string dir = argv[1];
    rng.seed((uint64_t)atoll(argv[2]));
    const int rounds = atoi(argv[3]);
    pc_io_set_thread_limit(4);

    // ---- 1. one file, four layouts, every reader ------------------------------------------------------------------
    std::vector<size_t> starts;
    const std::string fastq = make_fastq(1200, &starts);
    const std::string plain = dir + "/in.fastq";
    spit(plain, fastq);
    Reads want;
    CHECK(load_whole(plain, &want) == PC_OK, "plain load");
    CHECK(want.name.size() == 1200, "reads %zu", want.name.size());

    const std::string sized = dir + "/sized.fastq.gz", single = dir + "/single.fastq.gz", cat = dir + "/cat.fastq.gz";
    CHECK(pc_gzip_file(plain.c_str(), sized.c_str(), 0, 0) == PC_OK, "gzip sized");
    CHECK(pc_gzip_file(plain.c_str(), single.c_str(), 1, 1) == PC_OK, "gzip single");
    {
        std::string all;                                  // ordinary members back to back, with zero padding between two of them
        const size_t parts = 9;
        for (size_t p = 0; p < parts; ++p) {
            const size_t a = starts[(starts.size() - 1) * p / parts], b = starts[(starts.size() - 1) * (p + 1) / parts];
            spit(dir + "/part.fastq", fastq.substr(a, b - a));
            CHECK(pc_gzip_file((dir + "/part.fastq").c_str(), (dir + "/part.gz").c_str(), 1 + (int)(p % 6), 1) == PC_OK, "gzip part");
            all += slurp(dir + "/part.gz");
            if (p == 4) all += std::string(700, '\0');
        }
        spit(cat, all);
    }
    // a plain FASTA file by segments (streamed / sharded runs, round 6): multi-line records, blank and padded lines, headers
    // without a name (their bases belong to the NEXT record: a segment that would end in one is extended), a header the
    // reference only finds after stripping blanks; the segments must concatenate to the whole-file load at any target size
    {
        std::string fa;
        for (size_t i = 0; i < 4000; ++i) {
            const size_t len = rng() % 700;
            fa += (rng() % 40 == 0) ? "  >" : ">";
            if (rng() % 30 != 0) fa += "r" + std::to_string(i) + (rng() % 3 == 0 ? " some description" : "");
            fa += '\n';
            const size_t width = (rng() % 3 == 0) ? 60 : (rng() % 2 ? 80 : len + 1);
            for (size_t k = 0; k < len; k += width) {
                if (rng() % 50 == 0) fa += "  ";
                for (size_t t = k; t < std::min(len, k + width); ++t) fa += "ACGTacgtNU-"[rng() % 11];
                fa += (rng() % 60 == 0) ? " \n" : "\n";
            }
            if (rng() % 12 == 0) fa += "\n";
        }
        const std::string fap = dir + "/in.fasta";
        spit(fap, fa);
        Reads fwant;
        CHECK(load_whole(fap, &fwant) == PC_OK && fwant.name.size() > 3000, "fasta whole load (%zu reads)", fwant.name.size());
        for (int64_t target : {(int64_t)300, (int64_t)5000, (int64_t)1 << 16, (int64_t)3 << 20}) {
            Reads seg;
            CHECK(load_segments(fap, target, &seg) == PC_OK && seg == fwant, "fasta segments target %lld (%zu reads)", (long long)target, seg.name.size());
        }
        for (int k = 0; k < 50; ++k) {
            int64_t rec = -1;
            const int64_t pos = (int64_t)(rng() % (fa.size() + 10));
            CHECK(pc_fastq_find_record(fap.c_str(), pos, &rec) == PC_OK && rec >= std::min<int64_t>(pos, (int64_t)fa.size()) &&
                  (rec == (int64_t)fa.size() || fa[(size_t)rec] == '>'), "fasta find_record %lld -> %lld", (long long)pos, (long long)rec);
        }
    }
    // a file DENSE with the bytes a member starts with (1f 8b 08 00) inside its members -- stored members carry them verbatim,
    // deflated ones now and then: the reader guesses member starts from those bytes, and guesses it has already passed must
    // neither fill its window nor keep it waiting (the default reader hung on exactly this shape)
    {
        std::vector<size_t> dstarts;
        std::string dq;
        for (size_t i = 0; i < 9000; ++i) {
            dstarts.push_back(dq.size());
            const size_t len = 50 + rng() % 200;
            dq += "@d" + std::to_string(i) + "\n";
            for (size_t k = 0; k < len; ++k) dq += "ACGT"[rng() % 4];
            dq += "\n+\n";
            std::string q(len, '5');
            for (size_t k = 0; k + 4 <= len; k += 9) { q[k] = 0x1f; q[k + 1] = (char)0x8b; q[k + 2] = 8; q[k + 3] = 0; }
            dq += q + "\n";
        }
        dstarts.push_back(dq.size());
        auto stored_member = [](const std::string &d) {
            std::string m("\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff", 10);
            size_t at = 0;
            do {
                const size_t n = std::min<size_t>(65535, d.size() - at);
                m += (char)(at + n == d.size() ? 1 : 0);
                m += (char)(n & 0xFF); m += (char)(n >> 8); m += (char)(~n & 0xFF); m += (char)((~n >> 8) & 0xFF);
                m.append(d, at, n);
                at += n;
            } while (at < d.size());
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)d.data(), (uInt)d.size()), isz = (uint32_t)d.size();
            for (int k = 0; k < 4; ++k) m += (char)((crc >> (8 * k)) & 0xFF);
            for (int k = 0; k < 4; ++k) m += (char)((isz >> (8 * k)) & 0xFF);
            return m;
        };
        const std::string dplain = dir + "/dense.fastq", dense = dir + "/dense.fastq.gz";
        spit(dplain, dq);
        Reads dwant;
        CHECK(load_whole(dplain, &dwant) == PC_OK && dwant.name.size() == 9000, "dense plain load");
        std::string all;
        const size_t parts = 40;
        for (size_t p = 0; p < parts; ++p) {
            const size_t a = dstarts[(dstarts.size() - 1) * p / parts], b = dstarts[(dstarts.size() - 1) * (p + 1) / parts];
            if (p % 2 == 0) { all += stored_member(dq.substr(a, b - a)); continue; }
            spit(dir + "/part.fastq", dq.substr(a, b - a));
            CHECK(pc_gzip_file((dir + "/part.fastq").c_str(), (dir + "/part.gz").c_str(), 6, 1) == PC_OK, "gzip dense part");
            all += slurp(dir + "/part.gz");
        }
        spit(dense, all);
        pc_io_set_thread_limit(8);
        Reads got, st;
        CHECK(load_whole(dense, &got) == PC_OK && got == dwant, "whole dense (%zu reads)", got.name.size());
        CHECK(load_stream(dense, (int64_t)1 << 16, 100, &st) == PC_OK && st == dwant, "stream dense (%zu reads)", st.name.size());
        pc_io_set_thread_limit(4);
    }
    const std::string layouts[3] = {sized, single, cat};
    for (const std::string &p : layouts) {
        Reads got;
        CHECK(load_whole(p, &got) == PC_OK && got == want, "whole %s", p.c_str());
        for (int64_t target : {(int64_t)1 << 16, (int64_t)3 << 20, (int64_t)1 << 30}) {
            Reads st;
            CHECK(load_stream(p, target, 100, &st) == PC_OK && st == want, "stream %s target %lld (%zu reads)", p.c_str(), (long long)target, st.name.size());
        }
    }
    {
        // the members of the `cat` layout over W ranks: cuts at validated member starts of the compressed bytes
        // (pc_gz_member_start), every share through the stream over that byte range (pc_gzstream_open_range)
        const int64_t csize = (int64_t)slurp(cat).size();
        for (int world : {1, 2, 5}) {
            std::vector<int64_t> cuts{0};
            for (int r = 1; r < world; ++r) {
                int64_t at = -1;
                CHECK(pc_gz_member_start(cat.c_str(), csize * r / world, &at) == PC_OK && at >= csize * r / world && at <= csize, "member start %d/%d: %lld", r, world, (long long)at);
                cuts.push_back(at);
            }
            cuts.push_back(csize);
            Reads all;
            for (int r = 0; r < world; ++r) {
                if (cuts[(size_t)r + 1] <= cuts[(size_t)r]) continue;
                pc_gzstream *gs = nullptr;
                CHECK(pc_gzstream_open_range(cat.c_str(), cuts[(size_t)r], cuts[(size_t)r + 1], &gs) == PC_OK, "open range %d of %d", r, world);
                if (!gs) continue;
                pc_readset *rs = nullptr;
                int eof = 0;
                CHECK(pc_gzstream_next(gs, (int64_t)1 << 60, 0, &rs, &eof) == PC_OK, "range %d of %d", r, world);
                if (rs) { append(all, reads_of(rs)); pc_readset_free(rs); }
                pc_gzstream_close(gs);
            }
            CHECK(all == want, "member ranges of %d ranks (%zu reads)", world, all.name.size());
        }
    }
    {
        Reads seg;
        CHECK(load_segments(plain, 1 << 18, &seg) == PC_OK && seg == want, "segments (%zu reads)", seg.name.size());
        for (int k = 0; k < 50; ++k) {
            int64_t rec = -1;
            const int64_t pos = (int64_t)(rng() % (fastq.size() + 10));
            const int rc = pc_fastq_find_record(plain.c_str(), pos, &rec);
            if (rc == PC_OK) {
                bool is_start = false;
                for (size_t s : starts) if ((int64_t)s == rec) is_start = true;
                CHECK(is_start && rec >= std::min<int64_t>(pos, (int64_t)fastq.size()), "find_record %lld -> %lld", (long long)pos, (long long)rec);
            }
        }
    }

    {
        // the sized layout addressed by positions in its inflated bytes (sharded runs): same cuts, same reads as the plain file
        int64_t total = -1;
        CHECK(pc_gz_sized_size(sized.c_str(), &total) == PC_OK && total == (int64_t)fastq.size(), "sized size %lld", (long long)total);
        CHECK(pc_gz_sized_size(single.c_str(), &total) != PC_OK && pc_gz_sized_size(plain.c_str(), &total) != PC_OK, "sized size of other layouts");
        for (int world : {1, 2, 5, 33}) {
            std::vector<int64_t> cuts{0};
            for (int r = 1; r < world; ++r) {
                int64_t a = -1, b = -2;
                const int64_t pos = (int64_t)fastq.size() * r / world;
                CHECK(pc_gz_sized_find_record(sized.c_str(), pos, &a) == PC_OK && pc_fastq_find_record(plain.c_str(), pos, &b) == PC_OK && a == b,
                      "sized find_record %lld: %lld vs %lld", (long long)pos, (long long)a, (long long)b);
                cuts.push_back(a);
            }
            cuts.push_back((int64_t)fastq.size());
            Reads all;
            for (int r = 0; r < world; ++r) {
                pc_readset *rs = nullptr;
                CHECK(pc_readset_load_gz_range(sized.c_str(), cuts[(size_t)r], cuts[(size_t)r + 1], &rs) == PC_OK, "gz range %d of %d", r, world);
                if (rs) { append(all, reads_of(rs)); pc_readset_free(rs); }
            }
            CHECK(all == want, "gz ranges of %d ranks (%zu reads)", world, all.name.size());
        }
    }

    // ---- 2. the writer, plain and compressed, and back ---------------------------------------------------------------
    {
        pc_readset *rs = nullptr;
        CHECK(pc_readset_load(sized.c_str(), &rs) == PC_OK, "load for writing");
        const int64_t n = pc_readset_count(rs);
        const int32_t *len = pc_readset_lengths(rs);
        std::vector<int64_t> pr;
        std::vector<int32_t> ps, pl, pn, pf;
        Reads pieces;
        for (int64_t i = 0; i < n; ++i) {
            const int cuts = (int)(rng() % 3);                       // 0: dropped, 1: whole or trimmed, 2: split in two
            int32_t a = 0;
            for (int c = 0; c < cuts; ++c) {
                const int32_t left = len[i] - a;
                const int32_t l = c + 1 == cuts ? left - (int32_t)(left ? rng() % (left + 1) / 4 : 0) : (int32_t)(left ? rng() % (left + 1) : 0);
                pr.push_back(i); ps.push_back(a); pl.push_back(l); pn.push_back(cuts == 2 ? c + 1 : 0); pf.push_back((int32_t)(rng() % 2));
                a += l;
            }
        }
        const std::string o0 = dir + "/out0.fastq", o1 = dir + "/out1.fastq", z0 = dir + "/out0.fastq.gz", z1 = dir + "/out1.fastq.gz";
        const char *plain_paths[2] = {o0.c_str(), o1.c_str()}, *gz_paths[2] = {z0.c_str(), z1.c_str()};
        int64_t written[2] = {0, 0};
        CHECK(pc_readset_write(rs, (int64_t)pr.size(), pr.data(), ps.data(), pl.data(), pn.data(), pf.data(), 2, plain_paths, 1, written) == PC_OK, "write");
        int64_t sizes[2] = {0, 0};
        CHECK(pc_readset_write_sizes(rs, (int64_t)pr.size(), pr.data(), ps.data(), pl.data(), pn.data(), pf.data(), 2, 1, sizes) == PC_OK, "write_sizes");
        CHECK(sizes[0] == (int64_t)slurp(o0).size() && sizes[1] == (int64_t)slurp(o1).size(), "sizes %lld %lld", (long long)sizes[0], (long long)sizes[1]);
        pc_gzimage *img = nullptr;
        CHECK(pc_readset_compress(rs, (int64_t)pr.size(), pr.data(), ps.data(), pl.data(), pn.data(), pf.data(), 2, 1, 0, &img) == PC_OK && img, "compress");
        int64_t zbytes[2] = {0, 0}, pbytes[2] = {0, 0}, pos[2] = {0, 0};
        CHECK(pc_gzimage_sizes(img, 2, zbytes, pbytes) == PC_OK && pbytes[0] == sizes[0] && pbytes[1] == sizes[1], "image sizes");
        CHECK(pc_gzimage_write(img, 2, gz_paths, pos, 0) == PC_OK && pos[0] == zbytes[0] && pos[1] == zbytes[1], "image write");
        pc_gzimage_free(img);
        CHECK(pc_gz_finish(z0.c_str()) == PC_OK && pc_gz_finish(z1.c_str()) == PC_OK, "finish");
        for (int f = 0; f < 2; ++f) {
            Reads a, b, c;
            CHECK(load_whole(plain_paths[f], &a) == PC_OK && load_whole(gz_paths[f], &b) == PC_OK && a == b, "written file %d: plain = gz", f);
            CHECK(load_stream(gz_paths[f], 1 << 20, 0, &c) == PC_OK && a == c, "written file %d: streamed", f);
        }
        pc_readset_free(rs);
    }

    // ---- 3. damaged files through every reader: no crash, no sanitizer report -------------------------------------------
    const std::string originals[4] = {slurp(sized), slurp(single), slurp(cat), fastq};
    int64_t accepted = 0, refused = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int l = 0; l < 4; ++l) {
            const std::string bad = dir + (l < 3 ? "/bad.fastq.gz" : "/bad.fastq");
            spit(bad, damage(originals[l]));
            Reads a;
            const bool whole_ok = load_whole(bad, &a) == PC_OK;
            (whole_ok ? accepted : refused) += 1;
            // ONE member has one CRC-32 over everything: whatever is read from a damaged copy without complaint is the original
            // (a flipped bit in the header's time stamp, say) -- anything else got past the checks
            if (l == 1 && whole_ok) CHECK(a == want, "damaged single-member file accepted with other reads (round %d)", r);
            if (l == 0) {                          // (damaged sized members through the sharded run's readers)
                int64_t total = 0, rec = 0;
                if (pc_gz_sized_size(bad.c_str(), &total) == PC_OK && total > 0) {
                    (void)pc_gz_sized_find_record(bad.c_str(), (int64_t)(rng() % (uint64_t)total), &rec);
                    const int64_t a = (int64_t)(rng() % (uint64_t)total), b = a + (int64_t)(rng() % (uint64_t)(total - a + 1));
                    pc_readset *rs = nullptr;
                    (pc_readset_load_gz_range(bad.c_str(), a, b, &rs) == PC_OK ? accepted : refused) += 1;
                    if (rs) pc_readset_free(rs);
                }
            }
            if (l < 3) {
                const bool stream_ok = load_stream(bad, (int64_t)1 << (12 + rng() % 12), (int64_t)(rng() % 200), &a) == PC_OK;
                (stream_ok ? accepted : refused) += 1;
                if (l == 1 && stream_ok) CHECK(a == want, "damaged single-member file streamed with other reads (round %d)", r);
            } else {
                (load_segments(bad, (int64_t)1 << (12 + rng() % 10), &a) == PC_OK ? accepted : refused) += 1;
                int64_t rec = 0;
                (void)pc_fastq_find_record(bad.c_str(), (int64_t)(rng() % (originals[l].size() + 1)), &rec);
                const char *two[2] = {plain.c_str(), bad.c_str()};
                pc_readset *rs = nullptr;
                (pc_readset_load_many(two, 2, &rs) == PC_OK ? accepted : refused) += 1;
                if (rs) pc_readset_free(rs);
            }
        }
    }
    // ---- 4. the 2-bit packer against the definition in the header (lengths around the 64-base spans of its threads) -----
    for (int r = 0; r < 40; ++r) {
        static const char letters[] = "ACGTacgtUuNn-RYKMacgtACGTACGTACGT";
        const size_t n = r < 6 ? (size_t)r : (rng() % 3 ? rng() % 700 : 262144 * (1 + rng() % 3) + rng() % 130);
        std::string bases(n, 'A');
        for (char &c : bases) c = letters[rng() % (sizeof letters - 1)];
        std::vector<uint8_t> plane((n + 3) / 4 + 4, 0xEE);
        std::vector<int64_t> want_exc;
        for (size_t i = 0; i < n; ++i) if (!strchr("ACGTUacgtu", bases[i])) want_exc.push_back((int64_t)i);
        std::vector<int64_t> exc(want_exc.size() + 1, -7);
        int64_t nexc = -1;
        if (!want_exc.empty()) {
            CHECK(pc_pack_reads(bases.data(), (int64_t)n, plane.data(), exc.data(), (int64_t)want_exc.size() - 1, &nexc) == PC_ERR_BAD_ARG &&
                  nexc == (int64_t)want_exc.size() && exc[0] == -7, "pack: too little room must list nothing");
        }
        CHECK(pc_pack_reads(bases.data(), (int64_t)n, plane.data(), exc.data(), (int64_t)want_exc.size(), &nexc) == PC_OK, "pack n=%zu", n);
        exc.resize(want_exc.size());
        CHECK(nexc == (int64_t)want_exc.size() && exc == want_exc, "pack: exceptions n=%zu", n);
        bool same = true;
        for (size_t i = 0; i < n && same; ++i) {
            const char *at = strchr("AaCcGgTtUu", bases[i]);
            const unsigned want = at ? std::min<unsigned>(3u, (unsigned)(at - "AaCcGgTtUu") / 2) : 0u;
            same = ((plane[i / 4] >> (2 * (i % 4))) & 3u) == want;
        }
        CHECK(same && plane[(n + 3) / 4] == 0xEE, "pack: plane n=%zu", n);
    }

    printf("fuzz_io: %d check failure(s); damaged files: %lld read without complaint, %lld refused\n", failures, (long long)accepted, (long long)refused);
    return failures ? 1 : 0;
}
