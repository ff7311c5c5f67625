// pc_kernels.h -- argument structures shared by the HIP kernels and the host-side C ABI.
#pragma once
#include <stdint.h>

namespace pck {

// One tile = one wavefront's worth of work: up to 128 (window, adapter) pairs.  Lanes 0..63
// carry pair  pair_base + lane  in the LOW int16 half of every packed register and pair
// pair_base + 64 + lane  in the HIGH half.  All low-half pairs of a tile share adapter_lo,
// all high-half pairs share adapter_hi.
struct Tile {
    int64_t win_lo, win_hi;      // first WINDOW index of the low / high half (64 consecutive each);
                                 // win_lo == win_hi: both halves scan the same windows with two
                                 // different adapters (one read stream instead of two)
    int64_t out_lo, out_hi;      // first OUTPUT (pair) index of each half
    int32_t count_lo, count_hi;  // valid lanes of each half (0..64)
    int32_t adapter_lo, adapter_hi;
    int32_t rows;                // register rows R this tile must run with (>= both adapter lengths)
    int32_t pad_;
};

// A run of consecutive tiles of one job: n windows from win0 against one adapter (tiles of 128 windows,
// the high half 64 windows after the low one) or -- dual -- against two adapters on the same windows
// (tiles of 64 windows, the second adapter's records n after the first's).  The host describes a scan by
// a handful of runs; the tile table is expanded from them on the device (pc_reduce.hip).
struct TileRun {
    int64_t win0, out0, n;
    int64_t tile0;               // index of the run's first tile
    int32_t adapter_lo, adapter_hi, rows, dual;
};
int launch_expand_tiles(const TileRun *d_runs, int nruns, Tile *d_tiles, int64_t ntiles, void *stream);

struct ScanArgs {
    const uint8_t *arena;        // read bytes, 1 B/base, as delivered by the caller
    const int64_t *win_off;      // [nwindows] byte offset of the window's first column
    const int32_t *win_len;      // [nwindows] columns to run
    int32_t win_by_out;          // 1: the window arrays are indexed by OUTPUT index (pass-2 windows)
    const int32_t *col0;         // [npairs] global column of the window start   (null => 0)
    const int32_t *n_total;      // [npairs] whole read length                   (null => win_len)
    const int32_t *force_row;    // [npairs] end cell row (1..m) at the window's last column (null => scout)
    const int32_t *force_score;  // [npairs] score the forced cell must reproduce (null => unchecked)
    const int64_t *perm;         // [npairs] optional: slot s of a tile holds pair perm[s] (a pair of the same segment: the second
                                 // pass of the two-pass end scan takes a segment's pairs by end column -- bucket_pairs)
    const int32_t *trace_cols;   // [npairs] columns before the end cell the traced path can touch, + 2 (plan_kernel: from the
                                 // pair's own end row and score; null => the adapter's W + 2)
    const uint32_t *ad_codes;    // [nadapters][128] Dna5 codes 0..4
    const int32_t *ad_len;       // [nadapters]
    const Tile *tiles;
    int32_t ntiles;
    int32_t *out;                // trace kernels: 8 x int32 per pair; score kernels: 4 x int32 per pair
    int32_t *walk_req;           // optional: [ntiles][2][64][4] -- the traced kernel leaves every pair's end cell (score, I, J, tie) here
    int32_t *walk_req_tile;      // and the tile's trace-free prefix, and walk_kernel (one block per tile, a launch of its own) does the
                                 // traceback + digest: the scan's waves never sit through a walk's dependent loads
    uint32_t *slab;              // trace scratch: [grid][slab_cols][NW][64] dwords
    int64_t slab_stride;         // dwords per block
    int32_t slab_cols;
    int32_t match, mismatch, gap_open, gap_extend;   // gap_extend = pcb::kLinearExtend in linear mode
    int32_t init_extend;         // the scheme's real gap_extend (interior-window start state)
    int32_t linear;              // gap_open == gap_extend: no _correctTraceValue at the end cell
    int32_t kren;                // register variants: columns between renormalisations of the drifting coordinates
    uint32_t *err;               // err[0] += 1 on any internal inconsistency (reported loudly by the host)
    uint32_t one2, two2, sixteen2;   // packed constants kept opaque to the compiler (set by the launcher)
    int32_t gen_max_rows;            // generic (LDS-state) variant: largest tile.rows in the launch
    uint32_t *fin_scratch;           // [grid][rows*64*2] dwords: previous column kept for the last-column scan
    int32_t chunks, chunk_len;       // score-only pass: column chunks per tile (1 = whole window) and their length
    const int32_t *ad_span;          // [nadapters] warm-up columns (SPAN) for chunked passes
    const int32_t *ad_window;        // [nadapters] W + SPAN + 1 (pass-2 windows; W = window - SPAN - 1)
    int32_t debug;                   // PC_DEBUG_TRACE (timing experiments): 1 = no traceback, 2 = no slab stores; 4 = range-checking build (PC_CHECK_RANGE)
    int32_t f16_cen, f16_max_cols;   // packed-fp16 traced kernel: centring constant C and the columns it may run (pc_bounds.h f16_plan)
    uint32_t *work_counter;          // score pass: units beyond the grid are handed out by this counter (zeroed by the host) in
                                     // launch order -- longest tiles first -- instead of a fixed stride; null = fixed stride
    const int32_t *unit_prefix;      // score pass: [ntiles + 1] or null -- tile t owns units [unit_prefix[t], unit_prefix[t + 1]): only
                                     // the chunks that hold columns of its longest window (launch_unit_prefix)
};
// unit_prefix of a chunked score pass: one wave per tile takes the tile's longest window, a block scans the counts
int launch_unit_prefix(const Tile *tiles, int ntiles, const int32_t *win_len, int chunk_len, int32_t *real, int32_t *prefix, void *stream);

// pass-2 planner: from the score-only pass's (score, I, J) build the bounded windows
struct PlanArgs {
    const int64_t *win_off; const int32_t *win_len;     // the whole-read scan descriptors
    const int32_t *k1;                                  // [npairs][4] score, I, J, 0
    int64_t *win_off2; int32_t *win_len2; int32_t *col02; int32_t *ntot2; int32_t *force_row2; int32_t *force_score2;
    const int64_t *perm;                                // optional, as ScanArgs::perm (the tiles' slots -> pairs)
    int32_t *trace_cols2;                               // [npairs] out: I + floor((match I - score) / g) + 2 (see plan_kernel), or null
    int32_t match, gap_unit;                            // the scheme's match score and g = min(|open|, |extend|) for that bound
    const Tile *tiles; int32_t ntiles;                  // same tiles as the score pass
    int32_t chunks;                                     // chunk results per pair in k1 (>= 1)
    int32_t chunk_len;                                  // their length in columns (chunks > 1): only the chunks that start inside a window are merged
    const int32_t *ad_window;                           // [nadapters] W+SPAN+1 for that adapter length
    int32_t *score_out;                                 // PC_MODE_SCORE: [npairs][8] records (-2, J, I, 0, score, 0, 0, 0)
    int32_t end_align;                                  // 1: every window of a tile gets the same length (the larger of the two
                                                        // adapters' windows), a window that would start before the read's column 0
                                                        // by a lead-in of the bytes before the read (col0 < 0): all windows then END
                                                        // in the same local column and share their trace-free warm-up (trace16_kernel)
    const int32_t *end_records;                         // PC_MODE_TRACE_AT: the end cells come from the caller's PC_MODE_SCORE records
                                                        // ([npairs][8], indexed like k1 by output slot) instead of a score pass
    int32_t window_cap;                                 // > 0: no traced window longer than this (PC_MODE_TRACE_AT: the caller's max_len --
                                                        // a window that holds its read's column 0 needs no warm-up before it)
    uint32_t *err;
};

// per-read reduction of the end-window records (pc_reduce.hip)
struct ReduceArgs {
    const int32_t *records;          // TRACE_OUT_INTS per record
    int64_t n;                       // reads
    int32_t njobs;
    const int64_t *job_off;          // [njobs] first record of the job (device)
    const int32_t *job_side;         // [njobs] 0 = start window, 1 = end window (device)
    int32_t end_size, min_trim_size, extra_end_trim;
    double end_threshold;
    int32_t *start_trim, *end_trim;  // [n]
    int32_t nbins;                   // barcode bins (0 = no barcode calling)
    const int32_t *bin_start, *bin_end;   // [nbins] job of the bin's start / end entry, or -1 (device)
    double barcode_threshold, barcode_diff;
    int32_t require_two;
    int32_t *call;                   // [n] bin index or -1
    const unsigned long long *traced_mask;   // optional [njobs][mask_words]: bit r % 64 of word r / 64 of row j set = pair (j, r) holds
    int64_t mask_words;                      // a traced record; a clear bit = "no alignment" WITHOUT loading the record (pruned phase B:
                                             // 94 % of the records are score records the reduction would only skip)
};
int launch_reduce(const ReduceArgs &a, void *stream);
int launch_copy_windows(const uint8_t *arena, const int64_t *src_off, const int32_t *len, int64_t n, uint8_t *dst,
                        const int64_t *dst_off, int pad, void *stream);

// exact bit-parallel prefilter of the whole-read scan (pc_prefilter.hip)
struct PrefilterArgs {
    const uint8_t *arena;
    const int64_t *win_off;          // [nwindows]
    const int32_t *win_len;          // [nwindows]
    int64_t nwindows;
    int32_t chunks, chunk_len, warm; // every window is cut into `chunks` column chunks of chunk_len (+ warm columns before)
    const uint32_t *tables;          // [ngroups][256 byte values][P] Eq words (piece in the top bits, wildcard bits below)
    const int32_t *piece_meta;       // [ngroups][P][4]: piece length (0 = unused slot), max edits, mask word, mask bit
    uint32_t *mask;                  // [nwindows][words], zeroed by the host; bit set = the pair survives
    int32_t words;
    int32_t max_len;                 // the caller's bound on win_len: a longer window is a contract violation, flagged in
    uint32_t *err;                   // err[0] (a clear bit means PROVEN not a hit: an unscanned tail must never look like one)
};
int launch_prefilter(const PrefilterArgs &a, int pieces_per_lane, int ngroups, void *stream);

// seed stage of the prefilter (pc_prefilter.hip): exact q-grams of the adapter pieces, found by one pass over the reads,
// then verified by the same edit-distance test on the few columns around each find
struct SeedScanArgs {
    const uint8_t *arena;
    const int64_t *win_off;
    const int32_t *win_len;
    int64_t nwindows;
    int32_t chunks, chunk_len, warm;     // warm = longest seed - 1
    int32_t nq;                          // seed lengths present (1..3)
    int32_t q[3];                        // the lengths, longest first (8 >= q[0] >= q[1] >= q[2] >= 6; unused: 6)
    const uint32_t *bitmaps;             // kSeedBitmapWords words: 4^8 bits for q[0], then 4^7 for q[1], then 4^6 for q[2]
    uint32_t *cand;                      // candidates, 2 words each: window, column of the seed's last base | class << 28
    unsigned long long *count;           // appended so far (may exceed cap: then the host falls back to the exhaustive kernel)
    int64_t cap;
    int32_t max_len;                     // as in PrefilterArgs
    uint32_t *err;
};
struct SeedVerifyArgs {
    const uint8_t *arena;
    const int64_t *win_off;
    const int32_t *win_len;
    const uint32_t *cand;
    const unsigned long long *count;
    int64_t cap;
    int32_t q[3];
    int32_t first_off[3];                // per class: offset of its [4^q + 1] entry-range table in `first`
    const uint32_t *first;
    const int32_t *entries;              // 4 ints each: piece, offset of the seed in the piece, 0, 0; sorted by (class, q-gram)
    const int32_t *piece_meta;           // 4 ints each: len, k, mask word, mask bit
    const uint32_t *piece_eq;            // [npieces][8] Eq words of codes 0..4
    int32_t npieces;
    uint32_t *mask;
    int32_t words;
};
constexpr int kSeedBitmapWords = (1 << 16) / 32 + (1 << 14) / 32 + (1 << 12) / 32;
int launch_seed_scan(const SeedScanArgs &a, void *stream);
int launch_seed_verify(const SeedVerifyArgs &a, int64_t ncand, void *stream);
// The same two stages over reads held at 2 BITS PER BASE (pc_pack_reads' plane: base i in bits 2 (i % 16) of dword i / 16,
// SeqAn's Dna ordinals; bases that were not A/C/G/T/U sit there as 'A').  `arena` of the argument structs is the plane,
// win_off counts BASES; q-grams are little-endian here (first base in the lowest bits), and so are the
// host's tables for this route.  No exception list is consulted: a non-base read as 'A' can only ADD matches against
// adapters made of A/C/G/T (the only ones this route takes), so every pair it clears is still proven.
int launch_seed_scan_packed(const SeedScanArgs &a, void *stream);
int launch_seed_verify_packed(const SeedVerifyArgs &a, int64_t ncand, void *stream);
// Windows of the plane as bytes ('A' 'C' 'G' 'T', 'N' at the listed exceptions), window i to dst + dst_off[i], padded with
// `pad` up to dst_off[i + 1]; src_off counts bases and is ascending.
int launch_unpack_windows(const void *plane, const int64_t *exc_pos, int64_t nexc, const int64_t *src_off, const int32_t *len,
                          int64_t n, uint8_t *dst, const int64_t *dst_off, int pad, void *stream);

// exact pruning of phase B (pc_select.hip): which end-window pairs have to be traced
struct SelectArgs {
    const int32_t *records;          // score records (-2, Jc, I, 0, S, 0, 0, 0) -- or traced ones (round 2), 8 ints each
    int64_t n;                       // reads
    int32_t njobs;
    const int64_t *job_off;          // [njobs] first record of the job (device)
    const int32_t *job_side, *job_len, *job_call;   // [njobs] 0 = start window / 1 = end window, adapter length, feeds a barcode call (device)
    const int32_t *start_len, *end_len;             // [n] window lengths
    int32_t end_size, min_trim_size, extra_end_trim;
    int32_t match, gap_open, gap_extend, pen_max;   // pen_max: the dearest non-matching column
    double ident_c;                  // tau match - (1 - tau) pen_max, tau = (end_threshold - 1e-6) / 100
    int32_t round;                   // 1 or 2
    int32_t lds_counts;              // set by the launcher
    double call_level, call_level_diff;             // barcode threshold - diff, diff; call_level >= 1e8: no barcode call
    const unsigned long long *mask_prev;            // round 2: the pairs of round 1
    const int32_t *start_trim, *end_trim;           // round 2: trims so far [n]
    const double *best_full;         // round 2: [2][n] best traced full identity of a barcode pair per side
    unsigned long long *mask_out;    // [njobs][words]: bit r % 64 of word r / 64 = trace pair (job, r)
    int64_t words;
    unsigned long long *counts;      // [njobs], zeroed by the caller
    int32_t *ub_trim_out;            // optional [njobs][n] (round 1): the bounds themselves, for the tests
    double *ub_full_out;
};
int launch_select(const SelectArgs &a, void *stream);
struct GatherArgs {
    const unsigned long long *mask; int64_t words; int32_t njobs;
    const int64_t *first;            // [njobs] first window of the job in the lists (exclusive prefix of counts)
    unsigned long long *cursor;      // [njobs], zeroed by the caller
    const int64_t *job_off; const int32_t *job_side;
    const int64_t *start_off, *end_off; const int32_t *start_len, *end_len;   // [n]
    int64_t *win_off; int32_t *win_len;             // the traced scan's windows
    int64_t *dest; int32_t *pair_job; int64_t *pair_read;   // record index / job / read of each
};
int launch_gather(const GatherArgs &a, void *stream);
struct ScatterArgs {
    const int32_t *traced; int64_t count; const int64_t *dest; const int32_t *pair_job; const int64_t *pair_read;
    int32_t *records; const int32_t *job_side, *job_call; double *best_full; int64_t n;
};
int launch_scatter(const ScatterArgs &a, void *stream);
int launch_gather_records(const int32_t *records, const int64_t *index, int64_t count, int32_t *out, void *stream);

constexpr int TRACE_OUT_INTS = 8;
constexpr int SCORE_OUT_INTS = 4;

// rows-in-registers variants that are instantiated.
//  exact : adapter fills all R rows (no padding rows, adapter codes in SGPRs) -- the fast path
//  padded: any adapter length <= R (top padding rows; per-row constants via LDS broadcast)
static const int kExactRows[] = {22, 24, 28, 32};
static const int kPaddedRows[] = {16, 20, 24, 26, 28, 30, 32, 34, 36, 38, 40, 48, 56, 64, 68, 72, 112, 128};
constexpr int kMaxRows = 128;
// row classes of the packed-fp16 traced kernel (every class of the lists above up to 72 rows)
static const int kTrace16Rows[] = {16, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 48, 56, 64, 68, 72};

// m_lo/m_hi: adapter lengths of the two halves.  -> rows, *pad; 0 = no register variant fits: use
// the generic LDS-state kernel (rows = max length at run time)
inline int pick_rows(int m_lo, int m_hi, bool *pad)
{
    const int m = m_lo > m_hi ? m_lo : m_hi;
    if (m_lo == m_hi)
        for (int r : kExactRows) if (r == m) { *pad = false; return r; }
    for (int r : kPaddedRows) if (r >= m) { *pad = true; return r; }
    *pad = true;
    return 0;
}

// launchers (pc_kernels.hip); stream is a hipStream_t passed as void*
int launch_trace(const ScanArgs &a, int rows, bool pad, int grid, void *stream);
int launch_score(const ScanArgs &a, int rows, bool pad, int grid, void *stream);
bool trace16_has(int rows);
int launch_walk(const ScanArgs &a, int rows, int ntiles, void *stream);     // the tracebacks of a launch_trace16 launch that left requests
int launch_trace16(const ScanArgs &a, int rows, int grid, void *stream, bool score_only = false);   // packed-fp16 traced scan (needs a.f16_*);
                                                                                   // score_only: its first pass (score records, no trace)

// The pairs of every segment (a run of output slots that share one adapter) in the order of their end columns, coarsely:
// perm[segment's slots] = the segment's pairs, bucket (J / kBucketWidth, capped) by bucket; within a bucket in no particular
// order.  records: the score records (-2, J, I, 0, score, ...) indexed by pair.  blocks: the segments cut into pieces of at most
// kBucketBlock slots, {first slot, count, segment}; counts / cursors: [nsegments][kBuckets] scratch.
constexpr int kBuckets = 12, kBucketWidth = 14, kBucketBlock = 2048;
struct BucketBlock { int64_t first; int32_t count, segment; };
struct BucketArgs {
    const int32_t *records;
    const BucketBlock *blocks; int32_t nblocks;
    const int64_t *seg_first; int32_t nsegments;       // [nsegments] first slot of each segment
    uint32_t *counts, *cursors;                        // [nsegments][kBuckets]; zeroed by the launcher
    int64_t *perm;
};
int launch_bucket_pairs(const BucketArgs &a, void *stream);

// the glue of the middle scan (pc_middle.hip)
int launch_trim_windows(const int64_t *off, const int32_t *len, const int32_t *st, const int32_t *et, int64_t n, int64_t *toff, int32_t *tlen,
                        int64_t *stats, void *stream);
int launch_middle_hits(const int32_t *rec, int64_t n, double threshold, double *full, uint8_t *hit, void *stream);
int launch_group_survivors(const int32_t *mask, int64_t n, int words, const int32_t *gmask, int ngroups, uint8_t *cand, int64_t *counts, void *stream);
int launch_round_consume(const double *full_all, const int32_t *rec_all, const int64_t *cur, const int64_t *act, int64_t nact, int A, int64_t Dn,
                         double threshold, uint8_t *anyh, int32_t *a_hit, int64_t *cnt, int64_t *stats, void *stream);
int launch_plan(const PlanArgs &a, void *stream);
int trace_words_per_col(int rows);   // NW

// 2-bit plane (+ exception positions) -> bytes 'A','C','G','T' / 'N', followed by `pad` bytes of 'N' (pc_reduce.hip)
int launch_unpack(const void *packed, int64_t nbases, const int64_t *exc_pos, int64_t nexc, void *arena, int pad, void *stream);

// the plain-int32 kernel of pc_slow.hip: windows first_window .. first_window + count of the caller's table against ONE
// adapter (m Dna5 codes at `adapter`), records to out[(out_base + i) * 8]; state == nullptr: column state in LDS (m <= 128)
struct SlowArgs {
    const uint8_t *arena;
    const int64_t *win_off;
    const int32_t *win_len;
    int64_t first_window, count, out_base;
    const uint8_t *adapter;
    int32_t m, max_len;
    int32_t match, mismatch, gap_open, gap_extend;
    int32_t *out;
    int32_t *state;              // [2 * m][P] ints, or nullptr
    uint8_t *trace;              // [max_len][m][P] bytes
    int64_t P;                   // pairs of a launch (stride of the scratch)
    uint32_t *err;
};
int launch_slow(const SlowArgs &a, void *stream);

}  // namespace pck
