/*
 * porechop_amd.h -- C ABI of the MI355X-native adapter-alignment core (libporechop_amd.so,
 * also installable as Porechop's porechop/cpp_functions.so).
 *
 * Plain pointers and sizes only; no torch / HIP types.  Every entry point runs on the GPU:
 * there is NO CPU fallback anywhere behind this header (a missing device is an error).
 *
 * Part 1 is the reference's own FFI surface, kept bit-for-bit so that the unchanged
 * porechop/cpp_function_wrappers.py binds it.  Part 2 is the batch surface the reference does
 * not have (its API is one pair per call); it is what makes a GPU worthwhile.
 */
#ifndef PORECHOP_AMD_H
#define PORECHOP_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Part 1 -- drop-in replacements for the reference exports
 * ------------------------------------------------------------------------------------------ */

/* Replaces  porechop/include/adapter_align.h:12-16 / porechop/src/adapter_align.cpp:11-31
 * (bound by porechop/cpp_function_wrappers.py:27-33 with restype c_void_p).
 * Inputs are borrowed NUL-terminated ASCII strings.  Returns a malloc()ed C string
 *   "readStart,readEnd,adapterStart,adapterEnd,rawScore,alignedRegion%id,fullAdapter%id"
 * formatted exactly like porechop/src/alignment.cpp:113-121 ("%d" ints, "%f" doubles;
 * "-1,..." when either sequence is empty).  Served from the prefetch memo (Part 2) when the
 * pair is in it; otherwise by a GPU launch -- which, for a window of up to 1024 bases, covers that
 * window against EVERY adapter this process has asked about so far and memoises all of it (Porechop
 * walks a read's end window through the whole panel, porechop.py:296-322: one launch, then lookups;
 * PC_NO_SPECULATION=1 makes every miss a single-pair launch).  Any four integer scores (up to 2^20 in magnitude) and
 * adapters up to PC_MAX_ADAPTER_ANY bases are computed, as the reference computes them.  Never throws; on a device
 * error (or beyond those limits) it prints to stderr and returns NULL. */
char *adapterAlignment(char *readSeq, char *adapterSeq, int matchScore, int mismatchScore,
                       int gapOpenScore, int gapExtensionScore);

/* Replaces porechop/src/adapter_align.cpp:34-36 (cpp_function_wrappers.py:38-39): free(). */
void freeCString(char *p);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- batch API
 * ------------------------------------------------------------------------------------------ */

typedef struct pc_ctx pc_ctx;

enum {
    PC_OK = 0,
    PC_ERR_NO_DEVICE = -1,          /* no HIP device / HIP call failed */
    PC_ERR_UNSUPPORTED_SCORES = -2, /* a score above 2^20 in magnitude, or sums that leave the 32-bit range */
    PC_ERR_BAD_ARG = -3,
    PC_ERR_ADAPTER_TOO_LONG = -4,   /* adapter longer than PC_MAX_ADAPTER_ANY */
    PC_ERR_INTERNAL = -5            /* a kernel reported an inconsistency (never expected) */
};

#define PC_MAX_ADAPTER 128        /* the packed 16-bit kernels' limit; longer adapters take the plain-int32 kernel ... */
#define PC_MAX_ADAPTER_ANY 4096   /* ... up to this many bases */
#define PC_RESULT_INTS 8   /* readStart, readEnd, adapterStart, adapterEnd, rawScore,
                              matches, alignedRegionLength, fullAdapterLength.
                              identities are (100.0*matches)/length in double, as the reference
                              computes them (alignment.cpp:81-82,89-90); matches is the same
                              count for both.  Empty read or adapter: {-1,0,-1,0,INT_MIN,0,0,0}. */

/* `stream` arguments are hipStream_t handles passed as void*: NULL is HIP's default (null)
 * stream -- what torch.cuda.current_stream().cuda_stream is unless the caller changed it --
 * and PC_STREAM_CONTEXT selects the context's own non-blocking stream. */
#define PC_STREAM_CONTEXT ((void *)(intptr_t)-1)

/* scan modes */
#define PC_MODE_AUTO 0      /* by window length */
#define PC_MODE_TRACE 1     /* one pass, full trace (end windows) */
#define PC_MODE_TWO_PASS 2  /* score-only pass + bounded traced window (whole reads) */
#define PC_MODE_SCORE 3     /* score-only pass alone: records are (-2, J, I, 0, score, 0, 0, 0) -- the
                             * reference's end cell (row I of the adapter, window column J) and raw
                             * score, no traceback (pc_scan_device only) */

#define PC_MODE_TRACE_AT 4  /* the traced alignment of pairs whose END CELL is already known: on entry d_out holds, for
                             * every pair, the PC_MODE_SCORE record of the same (window, adapter) pair; on return the
                             * traced record.  Only the columns the path can occupy are traced (the last W + 2 before
                             * the end cell, W the exact bound of csrc/pc_bounds.h), the columns before them run the
                             * bare recurrence, the columns after the end cell are not run at all -- the machinery of
                             * PC_MODE_TWO_PASS's second pass, for windows of any length (phase B's exact pruning traces
                             * its few selected end-window pairs this way).  The score the window reproduces must equal
                             * the record's, or the pair is flagged (pc_sync: PC_ERR_INTERNAL).  pc_scan_device only. */

const char *pc_version(void);
const char *pc_strerror(int code);

/* 1 if (match, mismatch, gap_open, gap_extend) with adapters of up to max_adapter_len bases runs the PACKED 16-bit
 * kernels (match > 0, match > mismatch, negative gap scores, magnitudes that fit the int16 / fp16 lanes, adapters up to
 * PC_MAX_ADAPTER; gap_open == gap_extend selects the reference's linear-gap recurrence,
 * seqan/align/global_alignment_unbanded.h:217-220).  0: such pairs run the PLAIN-INT32 kernel (csrc/pc_slow.hip: one
 * lane per pair, every cell's trace kept) -- the same answers as the reference for ANY four integers up to 2^20 in
 * magnitude and adapters up to PC_MAX_ADAPTER_ANY, as porechop/porechop.py:145,196-202 accepts them, only about a
 * hundred times slower per cell and without the batch pipeline's exact prunings (PC_MODE_SCORE is refused there).
 * Nothing is ever approximated, and nothing runs on the CPU. */
int pc_scores_supported(int match, int mismatch, int gap_open, int gap_extend, int max_adapter_len);

/* device < 0: current device.  The context owns its stream-ordered scratch buffers. */
int pc_create(pc_ctx **ctx, int device);
void pc_destroy(pc_ctx *ctx);

int pc_set_scores(pc_ctx *ctx, int match, int mismatch, int gap_open, int gap_extend);

/* Upload the adapter panel (the strings of porechop/adapters.py ADAPTERS start/end sequences
 * or any other): adapter index i in the calls below refers to seqs[i]. */
int pc_set_adapters(pc_ctx *ctx, const char *const *seqs, int nadapters);

/* Host-buffer batch: pair p aligns window  read_arena[win_off[p] .. win_off[p]+win_len[p])
 * against adapter adapter_idx[p]; results out[p*PC_RESULT_INTS ...].  Pairs may come in any
 * order; the library groups them.  Blocking. */
int pc_align_batch_host(pc_ctx *ctx, const char *read_arena, int64_t arena_bytes,
                        const int64_t *win_off, const int32_t *win_len, const int32_t *adapter_idx,
                        int64_t npairs, int mode, int32_t *out);

/* Device-buffer batch (inputs already resident in HBM; nothing crosses PCIe but the small job
 * table).  d_* are device pointers.  d_win_off/d_win_len describe `nwindows` windows; job k scans
 * windows [job_start[k], job_start[k+1]) against adapter job_adapter[k] and -- when job_adapter_b
 * is non-NULL and job_adapter_b[k] >= 0 -- also against that second adapter IN THE SAME PASS (each
 * window is then read from HBM once for both).  Results, PC_RESULT_INTS each, are written in job
 * order: for job k first its n_k records for job_adapter[k], then (if any) its n_k records for
 * job_adapter_b[k].  max_len is an upper bound on every win_len (checked on the device).  The
 * arena must be readable 16 bytes past its last window (the kernels fetch 16 columns per load).  Asynchronous on `stream`; call pc_sync()
 * (or otherwise order your reads after it on the same stream) before reading d_out.
 * A context is used from ONE host thread and ONE stream at a time: its scratch (trace slab, pass-1
 * buffer, work counters) is shared by successive calls and ordered only by that stream.  Other streams
 * of the process (an upload of the next batch, say) are never synchronised with: the tile table of a call
 * is built on the context's own stream in the table slot the call before last used. */
int pc_scan_device(pc_ctx *ctx, const void *d_arena, const int64_t *d_win_off,
                   const int32_t *d_win_len, int64_t nwindows, const int32_t *job_adapter,
                   const int32_t *job_adapter_b, const int64_t *job_start, int njobs, int max_len,
                   int mode, int32_t *d_out, void *stream);

/* Load-balancing hint for the whole-read scans of the following pc_scan_device calls: the typical
 * (mean) window length, when max_len is far above it -- real read sets are log-normal, the longest read
 * tens of times the mean.  The score pass then cuts every window into column chunks about that long,
 * hands them to the chip longest window first and lets workgroups draw further chunks as they finish,
 * so a launch no longer lasts as long as its longest read.  0 (the default) = lengths are about uniform.
 * Affects scheduling only, never results (the chunked pass is exact, pc_bounds.h). */
int pc_set_length_hint(pc_ctx *ctx, int typical_len);

/* Kernel-variant switch for cross-checks: enabled != 0 makes every later launch of this context use the packed-int16
 * kernels (21 / 6 ops per cell pair) even where the packed-fp16 ones (13.25 / 5) are proven exact by the host gates
 * of csrc/pc_bounds.h.  Results are bit-identical either way -- that is what a cross-check of the two asserts
 * (bench.py's device_crosscheck leg, tests/test_gpu_parity.py).  Default 0; PC_DISABLE_F16=1 / PC_JIT_INT16=1 in the
 * environment force the same process-wide. */
int pc_set_int16_only(pc_ctx *ctx, int enabled);

/* Waits for `stream` and returns PC_ERR_INTERNAL if any kernel since the last pc_sync reported
 * an inconsistency. */
int pc_sync(pc_ctx *ctx, void *stream);

/* Kernel timing hooks (bench.py roofline leg): when enabled, every kernel launch made by
 * pc_scan_device is bracketed by HIP events on the launch stream.  pc_get_timing waits for the
 * stream, then returns per kernel kind (0 = generic score-only scan, 1 = window planner, 2 = traced
 * scan, 3 = run-time specialised score-only scan, 4 = exact prefilter: all its launches and its one host round trip,
 * 5 = the prefilter's seed scan alone, a sub-interval of 4 whose "pairs" are windows, 6 = the selection kernels of
 * phase B's exact pruning: pc_phase_b_select / _gather / _scatter) the summed duration in milliseconds, the number
 * of launches and the number of pairs they covered, and resets the accumulators.  Each array
 * holds PC_KERNEL_KINDS entries. */
/* (PC_KERNEL_KINDS grew from 4 to 7 in round 3: a caller built against the older header must enlarge its three arrays.) */
#define PC_KERNEL_KINDS 7
int pc_set_timing(pc_ctx *ctx, int enabled);
int pc_get_timing(pc_ctx *ctx, void *stream, double *ms, int64_t *launches, int64_t *pairs);

/* Per-read reduction of end-window records on the device -- what porechop/nanopore_read.py:166-208
 * (find_start_trim / find_end_trim) and :399-466 (determine_barcode, without the Albacore rule) do per
 * read, for a whole batch.  d_records: the records pc_scan_device wrote for njobs jobs over the SAME n
 * reads; job j = one adapter sequence against every read's start (job_side[j] = 0) or end (1) window,
 * its record for read r at d_records[(job_record_offset[j] + r) * PC_RESULT_INTS].  Writes
 * d_start_trim[n] / d_end_trim[n]: the largest trim any alignment justifies (aligned identity >
 * end_threshold, not touching the window's inner edge, at least min_trim_size bases; + extra_end_trim).
 * nbins > 0 additionally calls barcodes: bin k's start / end entry is job bin_start_job[k] /
 * bin_end_job[k] (-1 = none, scores 0.0) and d_call[r] becomes the bin index or -1 ('none') by the
 * reference's rules, ties resolved like Python's stable sort (earlier entry wins; start entries
 * before end entries).  Identities are the %f-rounded doubles Python parses.  Host arrays are copied
 * before the call returns; asynchronous on `stream`. */
int pc_phase_b_reduce(pc_ctx *ctx, const int32_t *d_records, int64_t n, int njobs,
                      const int64_t *job_record_offset, const int32_t *job_side, int end_size,
                      int min_trim_size, int extra_end_trim, double end_threshold,
                      int32_t *d_start_trim, int32_t *d_end_trim, int nbins,
                      const int32_t *bin_start_job, const int32_t *bin_end_job,
                      double barcode_threshold, double barcode_diff, int require_two_barcodes,
                      int32_t *d_call, void *stream);

/* pc_phase_b_reduce for records most of which are PC_MODE_SCORE records left untraced by the exact pruning below:
 * d_traced_mask[njobs][(n + 63) / 64] (bit r % 64 of word r / 64 of row j) says which pairs hold a traced record -- the union
 * of the selection rounds' masks; a pair whose bit is clear is read as "no alignment" without its record being loaded (the
 * reduction would skip it anyway: 94 % of the records of a barcoded batch).  NULL = pc_phase_b_reduce. */
int pc_phase_b_reduce_masked(pc_ctx *ctx, const int32_t *d_records, int64_t n, int njobs,
                             const int64_t *job_record_offset, const int32_t *job_side, int end_size,
                             int min_trim_size, int extra_end_trim, double end_threshold,
                             int32_t *d_start_trim, int32_t *d_end_trim, int nbins,
                             const int32_t *bin_start_job, const int32_t *bin_end_job,
                             double barcode_threshold, double barcode_diff, int require_two_barcodes,
                             int32_t *d_call, const uint64_t *d_traced_mask, void *stream);

/* Exact pruning of phase B: of the ~200 end-window alignments a barcoded read gets, two or three decide its trims and
 * its barcode call; a score-only pass (PC_MODE_SCORE, 5 instead of 13.25 packed operations per two cells, no trace)
 * gives every alignment's end cell and score, and those bound what the alignment can contribute (the bounds and
 * their derivation: porechop_amd/csrc/pc_select.hip, porechop_amd/pipeline.py).  The caller runs
 *   score scan -> select(round 1) -> gather -> traced scan of the gathered windows -> scatter -> pc_phase_b_reduce
 *              -> select(round 2) -> gather -> traced scan -> scatter -> pc_phase_b_reduce (final),
 * which yields exactly the trims and calls of tracing everything (pc_phase_b_reduce reads a score record left in
 * place as "no alignment").  All pointers are DEVICE pointers; everything is asynchronous on `stream`.
 *
 * pc_phase_b_select: d_records as for pc_phase_b_reduce, but holding PC_MODE_SCORE records (round 1) or those with
 * round 1's traced records scattered over them (round 2); d_job_adapter_len[j] the job's adapter length,
 * d_job_calls[j] != 0 if its full identity feeds a barcode call; d_start_len / d_end_len the reads' end-window
 * lengths.  call_level = barcode_threshold - barcode_diff and call_level_diff = barcode_diff (call_level >= 1e8: no
 * barcode call).  Round 2 takes round 1's mask, the trims so far and d_best_full[2][n] (best traced barcode identity
 * per side, maintained by pc_phase_b_scatter; zero it before round 1).  Writes d_mask_out[njobs][(n + 63) / 64] -- bit
 * r % 64 of word r / 64 of row j: trace pair (j, r) -- and d_counts[njobs], the pairs selected per job.
 * d_ub_trim_out / d_ub_full_out (optional, round 1): the bounds themselves, [njobs][n], for tests. */
int pc_phase_b_select(pc_ctx *ctx, const int32_t *d_records, int64_t n, int njobs,
                      const int64_t *d_job_record_offset, const int32_t *d_job_side,
                      const int32_t *d_job_adapter_len, const int32_t *d_job_calls,
                      const int32_t *d_start_len, const int32_t *d_end_len, int end_size,
                      int min_trim_size, int extra_end_trim, double end_threshold, int round,
                      double call_level, double call_level_diff, const uint64_t *d_mask_prev,
                      const int32_t *d_start_trim, const int32_t *d_end_trim, const double *d_best_full,
                      uint64_t *d_mask_out, uint64_t *d_counts, int32_t *d_ub_trim_out,
                      double *d_ub_full_out, void *stream);
/* The selected pairs of a mask as the window lists of a traced scan: job j's pairs become windows
 * [d_job_first[j], d_job_first[j] + count[j]) (d_job_first = exclusive prefix sum of the counts), in no particular
 * order within the job; d_dest / d_pair_job / d_pair_read give each window's record index, job and read.  d_cursor:
 * njobs words of scratch. */
int pc_phase_b_gather(pc_ctx *ctx, const uint64_t *d_mask, int64_t n, int njobs, const int64_t *d_job_first,
                      uint64_t *d_cursor, const int64_t *d_job_record_offset, const int32_t *d_job_side,
                      const int64_t *d_start_off, const int32_t *d_start_len, const int64_t *d_end_off,
                      const int32_t *d_end_len, int64_t *d_win_off, int32_t *d_win_len, int64_t *d_dest,
                      int32_t *d_pair_job, int64_t *d_pair_read, void *stream);
/* d_out[k] = d_records[d_index[k]] (PC_RESULT_INTS each): the score records of the gathered pairs (d_index = d_dest) in the
 * order of the traced scan's windows -- what PC_MODE_TRACE_AT expects in its output buffer on entry. */
int pc_gather_records(pc_ctx *ctx, const int32_t *d_records, const int64_t *d_index, int64_t count, int32_t *d_out,
                      void *stream);
/* The traced records of the gathered windows over the score records they replace; d_best_full (may be NULL) is
 * raised to the full identity of every traced barcode pair. */
int pc_phase_b_scatter(pc_ctx *ctx, const int32_t *d_traced, int64_t count, const int64_t *d_dest,
                       const int32_t *d_pair_job, const int64_t *d_pair_read, int32_t *d_records,
                       const int32_t *d_job_side, const int32_t *d_job_calls, double *d_best_full, int64_t n,
                       void *stream);

/* Packed private copies of n windows on the device: window i of d_arena (d_src_off[i], d_len[i]) is copied
 * to d_dst + d_dst_off[i], and the bytes from its end up to d_dst_off[i+1] are set to `pad` (d_dst_off has
 * n + 1 entries).  Porechop masks every middle hit in a copy of the read and aligns again
 * (nanopore_read.py:212-243); the batch pipeline keeps such copies only for the reads that had a hit,
 * back to back whatever their lengths.  Asynchronous on `stream`. */
int pc_copy_windows(pc_ctx *ctx, const void *d_arena, const int64_t *d_src_off, const int32_t *d_len,
                    int64_t n, void *d_dst, const int64_t *d_dst_off, int pad, void *stream);

/* Reads cross PCIe at 2 bits per base (north_star: "2-bit-packed read windows"; a read set is 8 GB of bases per million
 * 8-kb reads and the link moves ~57 GB/s, so at one byte per base the upload, not the scan, bounds a run).
 * pc_pack_reads (host, all cores the caller may use: pc_io_set_thread_limit): base i of arena[0 .. nbases) -- one byte per
 * base as pc_readset_arena delivers them -- becomes bits 2*(i%4).. of packed[i/4] ((nbases + 3) / 4 bytes; keep the
 * buffer a multiple of 4 bytes long): SeqAn's Dna ordinal values, A 0, C 1, G 2, T/U 3, either case
 * (porechop/include/seqan/basic/alphabet_residue_tabs.h:113-140).  Every other byte ('N', '-', IUPAC codes ...: all of
 * them Dna5 'N' to the alignment, which is what the reference's conversion makes of them) is an EXCEPTION: code 0 in
 * the plane and its position appended to exc_pos (ascending).  *nexc = the number of exceptions; if that exceeds
 * exc_cap nothing is listed and PC_ERR_BAD_ARG is returned (call again with room for *nexc entries).
 * pc_unpack_device is the inverse on the device, into the byte arena every scan entry point takes: d_arena[i] = 'A' /
 * 'C' / 'G' / 'T', 'N' at the exceptions, then pad_bytes bytes of 'N' (the 16 readable bytes the scans need past the last
 * window, say).  Alignment-equivalent to the original bytes: Dna5(original) == Dna5(unpacked) for every base.
 * d_packed 4-byte aligned, d_arena 16-byte aligned, room for nbases + pad_bytes bytes.  HBM-bound (0.25 B in, 1 B out per
 * base), asynchronous on `stream`. */
int pc_pack_reads(const char *arena, int64_t nbases, uint8_t *packed, int64_t *exc_pos, int64_t exc_cap, int64_t *nexc);
int pc_unpack_device(pc_ctx *ctx, const void *d_packed, int64_t nbases, const int64_t *d_exc_pos, int64_t nexc,
                     void *d_arena, int pad_bytes, void *stream);

/* Exact bit-parallel prefilter of the whole-read ("middle") scan.  Porechop keeps a whole-read alignment only when
 * its full-adapter identity reaches --middle_threshold (porechop/nanopore_read.py:224-241); such an alignment has at
 * most pc_prefilter_max_edits(adapter length, threshold) non-matching columns inside the adapter's span, so the
 * adapter lies within that many unit-cost edits (substitutions, insertions, deletions; adapter global, overhanging
 * adapter bases are deletions; read local; equality of Dna5 codes, N == N, as in the reference) of a substring of
 * the read.  pc_prefilter_device decides that for every (window, adapter) pair with Myers' bit-vector algorithm,
 * one lane per (window chunk, adapter piece) (csrc/pc_prefilter.hip; the alternative the reference's README.md:355-357
 * points to).  Row w of d_mask ([nwindows][ceil(nadapters / 32)] words, written by the call) gets bit j set iff window
 * w MAY hold adapters[j] within max_edits[j] edits: always set when it does (so a pair whose bit is clear is PROVEN not
 * to be a hit, and only the survivors need the DP), never set when it does not for adapters of at most 32 bases;
 * a longer adapter is represented by its first 32 bases with the same bound when max_edits <= 8, and otherwise cut into
 * ceil(m / 32) pieces, surviving when one piece lies within floor(max_edits / pieces) edits (pigeonhole) -- a superset
 * either way.  max_edits[j] < 0 = do not filter adapter j (bit set for every non-empty window).
 * adapters[] indexes the table of pc_set_adapters; max_len bounds every win_len (a longer window is flagged on the
 * device and reported by pc_sync as PC_ERR_INTERNAL: its tail would otherwise go unscanned and look "proven"); the arena
 * must be readable 16 bytes past its last window, and -- because the kernels fetch whole aligned 16-byte blocks
 * (exhaustive kernel) and 128-byte lines (seed scan) -- from the 128-byte boundary at or below d_arena: d_arena itself
 * should be 128-byte aligned (hipMalloc and torch allocations are 256-byte aligned), or sit inside an allocation that
 * starts at or below that boundary.  Asynchronous on `stream`; honours pc_set_length_hint. */
int pc_prefilter_max_edits(int adapter_len, double threshold_percent);
int pc_prefilter_device(pc_ctx *ctx, const void *d_arena, const int64_t *d_win_off, const int32_t *d_win_len,
                        int64_t nwindows, int max_len, const int32_t *adapters, const int32_t *max_edits,
                        int nadapters, uint32_t *d_mask, void *stream);

/* The prefilter over reads held at 2 BITS PER BASE (north_star: "2-bit-packed read windows"): d_plane is pc_pack_reads' plane
 * in HBM -- a quarter of the bytes of the one HBM-bound kernel of the path, and sixteen bases ARE a dword: no byte -> code
 * step, one funnel shift per q-gram (csrc/pc_prefilter.hip seed_scan_packed_kernel) -- d_win_off counts BASES of the plane.
 * Same mask, same meaning.  Bases that were not A/C/G/T/U sit in the plane as 'A' and the exception list is NOT consulted:
 * against adapters made of A/C/G/T(U) that can only ADD survivors, so a cleared bit is still a proof.  The route takes only
 * such adapters, and only where the seed stage covers every piece (parts of >= 6 bases): otherwise
 * PC_ERR_UNSUPPORTED_SCORES -- unpack the reads (pc_unpack_device) and call pc_prefilter_device.  d_plane 16-byte aligned
 * and readable 64 bytes past its last base.
 * pc_unpack_windows: n windows of the plane (d_src_off in bases, ascending, not overlapping; d_len) as bytes -- 'A' 'C' 'G' 'T',
 * 'N' at the exceptions -- window i at d_dst + d_dst_off[i], padded with `pad` up to d_dst_off[i + 1] (d_dst_off has n + 1
 * entries).  What a run that keeps its reads packed unpacks: the 150-base end windows of every read and the whole of the
 * few reads that survive the prefilter. */
int pc_prefilter_packed(pc_ctx *ctx, const void *d_plane, const int64_t *d_win_off, const int32_t *d_win_len,
                        int64_t nwindows, int max_len, const int32_t *adapters, const int32_t *max_edits,
                        int nadapters, uint32_t *d_mask, void *stream);

/* The seed stage sizes its verification launch from a count it reads back from the device: ONE host round trip per call.
 * pc_prefilter_defer_count(ctx, 1) removes it -- the verification is launched for the whole candidate list (threads beyond
 * the count leave at once) and the count follows to pinned host memory; after the caller's next synchronisation of the
 * stream, pc_prefilter_overflowed(ctx) says whether the last call's list overflowed (1: its mask is NOT complete -- repeat
 * the call with the deferral off; rare: low-complexity reads against a low-complexity seed). */
/* The glue of the middle scan (porechop/nanopore_read.py:56-62,210-243 for a whole batch) as single launches -- as torch
 * expressions it is ~150 launches of a few microseconds per step, and the GPU idles between them.  All pointers: device memory.
 * pc_trim_windows: seq[start_trim : len - end_trim] with Python's slice semantics -> toff / tlen; stats int64[4] = reads with a
 *   non-empty interval, longest, 2^40 - shortest non-empty (0: none), sum.
 * pc_middle_hits: full-adapter identity (the %f-printed double) of whole-read records and whether it reaches the threshold.
 * pc_group_survivors: cand[g][w] = (mask row w AND gmask row g) != 0 over the prefilter's mask, counts[g] = how many.
 * pc_round_consume: per active masked read the first adapter >= its cursor whose record is a hit; stats int64[4] = alignments
 *   consumed, reads that hit, 2^40 - smallest hit adapter (0: none), bases to mask. */
int pc_trim_windows(pc_ctx *ctx, const int64_t *d_off, const int32_t *d_len, const int32_t *d_start_trim, const int32_t *d_end_trim,
                    int64_t n, int64_t *d_toff, int32_t *d_tlen, int64_t *d_stats, void *stream);
int pc_middle_hits(pc_ctx *ctx, const int32_t *d_records, int64_t n, double threshold, double *d_full, uint8_t *d_hit, void *stream);
int pc_group_survivors(pc_ctx *ctx, const int32_t *d_mask, int64_t n, int words, const int32_t *d_gmask, int ngroups, uint8_t *d_cand,
                       int64_t *d_counts, void *stream);
int pc_round_consume(pc_ctx *ctx, const double *d_full_all, const int32_t *d_rec_all, const int64_t *d_cur, const int64_t *d_act,
                     int64_t nact, int nadapters, int64_t ndirty, double threshold, uint8_t *d_anyh, int32_t *d_a_hit, int64_t *d_cnt,
                     int64_t *d_stats, void *stream);

int pc_prefilter_defer_count(pc_ctx *ctx, int enabled);
int pc_prefilter_overflowed(pc_ctx *ctx);
int pc_unpack_windows(pc_ctx *ctx, const void *d_plane, const int64_t *d_exc_pos, int64_t nexc, const int64_t *d_src_off,
                      const int32_t *d_len, int64_t n, void *d_dst, const int64_t *d_dst_off, int pad, void *stream);

/* Debug builds of the 16-bit kernels (PC_CHECK_RANGE=1: packed-fp16 traced kernel, row classes 24/28/30/40;
 * PC_JIT_CHECK_RANGE=1: the run-time specialised score kernel) record the extremes of every DP value they
 * hold, in the kernel's own offset coordinates; this returns them since the last call and resets.  The
 * exactness argument needs |value| <= 2040 in the fp16 kernels and <= 32000 in the int16 ones. */
int pc_debug_value_range(pc_ctx *ctx, int32_t *lo, int32_t *hi);

/* Packed VALU operations per TWO DP cells, times 100, of the traced end-window kernel this context's
 * scoring scheme selects (2100: packed-int16 kernel; 1325: packed-fp16 kernel) -- the denominator
 * of the VALU roofline bench.py reports. */
int pc_trace_ops_x100(pc_ctx *ctx);

/* Run-time specialised score kernels are compiled (hiprtc) once an adapter pair's accumulated work
 * pays for it.  By default the compile happens in place, inside the pc_scan_device call that
 * crosses the threshold -- provided that call alone is large enough to be worth the stall; a small
 * launch that merely tips the running total over never waits.  With pc_jit_async(1) every compile
 * runs on a worker thread and launches keep using the generic kernels until the kernel is ready --
 * no stall for one-shot runs.  Process-wide. */
void pc_jit_async(int enabled);
/* With asynchronous specialisation on, call this before the process exits: it drops compiles that
 * have not started and waits for the one in flight (a worker thread must not be inside hiprtc while
 * the process image is torn down).  The Python binding registers it with atexit. */
void pc_jit_shutdown(void);

/* Specialised kernels are pure functions of (kernel source, adapter pair's row pattern, scoring scheme): their code
 * objects are kept on disk and loaded instead of compiled -- from <directory of this library>/kernel_cache (filled
 * when the library is built: every set of the static panel, porechop/adapters.py:77-463, under the default scheme)
 * and from PC_JIT_CACHE_DIR / $XDG_CACHE_HOME/porechop_amd / ~/.cache/porechop_amd (what this machine compiled at
 * run time; PC_JIT_CACHE_DIR=off disables it).  A kernel found on disk is used from the first launch on.
 * pc_jit_precompile builds the kernel that scans adapter_a and adapter_b in one pass (adapter_b NULL or "": a alone)
 * into cache_dir (NULL or "": the in-tree directory).  It needs hiprtc but NO device.  Returns 0 = compiled and
 * written, 1 = already there, < 0 = this pair / scheme has no specialised kernel, or the compile or write failed. */
int pc_jit_precompile(const char *adapter_a, const char *adapter_b, int match, int mismatch, int gap_open,
                      int gap_extend, const char *cache_dir);
/* Kernels this process compiled with hiprtc / took from a kernel cache on disk so far. */
void pc_jit_stats(int64_t *compiled, int64_t *from_disk);

/* Format one result record exactly as the reference prints it; buf must hold >= 160 bytes. */
int pc_format_result(const int32_t *rec, char *buf, size_t buflen);
/* The same for n records at once: the strings back to back in buf, each followed by '\n'; *used = bytes written.
 * buflen >= 161 * n always suffices.  (The drop-in's prefetch turns millions of records into the strings the
 * reference's Python parses, porechop/nanopore_read.py:476-491: one call instead of one per record.) */
int pc_format_results(const int32_t *recs, int64_t n, char *buf, int64_t buflen, int64_t *used);

/* Prefetch memo for the per-call symbol: compute these pairs on the GPU now and remember the
 * answers, keyed by (window bytes, adapter bytes, scores), so that later adapterAlignment()
 * calls with the same arguments are lookups.  Uses the process-wide default context. */
int pc_prefetch(const char *read_arena, int64_t arena_bytes, const int64_t *win_off,
                const int32_t *win_len, const char *const *adapters, const int32_t *adapter_idx,
                int64_t npairs, int match, int mismatch, int gap_open, int gap_extend);
void pc_memo_clear(void);
/* hits, misses (single-pair launches), entries */
void pc_memo_stats(int64_t *hits, int64_t *misses, int64_t *entries);

/* ------------------------------------------------------------------------------------------
 * Part 3 -- host ingest (the row after the hot path): FASTA / FASTQ, plain or
 * gzip, parsed as porechop/misc.py:60-168 does and normalised as NanoporeRead.__init__
 * (porechop/nanopore_read.py:23-35: upper-case; U->T when U's outnumber T's; qualities padded with
 * '+'), but into ONE packed arena + offset/length tables -- the inputs of the batch API above --
 * instead of per-read Python tuples.  Host code only.
 * ------------------------------------------------------------------------------------------ */
typedef struct pc_readset pc_readset;
/* Always sets *out (so pc_readset_error can be read); free it with pc_readset_free. */
int pc_readset_load(const char *path, pc_readset **out);
void pc_readset_free(pc_readset *rs);
const char *pc_readset_error(const pc_readset *rs);
int64_t pc_readset_count(const pc_readset *rs);
int pc_readset_is_fastq(const pc_readset *rs);
/* reads back to back, 1 byte per base, followed by 64 bytes of 'N' padding; *bytes includes it */
const char *pc_readset_arena(const pc_readset *rs, int64_t *bytes);
const int64_t *pc_readset_offsets(const pc_readset *rs);
const int32_t *pc_readset_lengths(const pc_readset *rs);
const char *pc_readset_name(const pc_readset *rs, int64_t i);     /* full header, no leading marker */
const char *pc_readset_quals(const pc_readset *rs, int64_t i);    /* FASTQ only */
int pc_readset_is_rna(const pc_readset *rs, int64_t i);
/* Several FASTQ (or several FASTA) files into ONE read set, in the order given -- the Albacore
 * directory input of porechop/porechop.py:232-259; pc_readset_file_index()[i] = which path read i
 * came from. */
int pc_readset_load_many(const char *const *paths, int npaths, pc_readset **out);
/* Streaming ingest of a plain, regular 4-line FASTQ file (or a plain FASTA file, cut where a line begins with '>'): the
 * records that start in [byte_begin, cut), where
 * cut is the first record start at or after byte_begin + target_bytes (or the end of the file); *next_begin = cut.
 * Successive calls (byte_begin = the previous *next_begin, starting at 0) yield the reads of pc_readset_load in
 * the same order, a block at a time -- so that a run's memory is bounded by its blocks and ingest, scan and
 * writing of successive blocks overlap.  Returns PC_ERR_UNSUPPORTED_SCORES ("not streamable") for gzip (see
 * pc_gzstream_* below) or an irregular record: load the whole file with pc_readset_load then. */
int pc_readset_load_segment(const char *path, int64_t byte_begin, int64_t target_bytes, int64_t *next_begin,
                            pc_readset **out);
const int32_t *pc_readset_file_index(const pc_readset *rs);

/* Worker threads the ingest / writer calls made BY THE CALLING THREAD may use from now on (0 = the default: every core
 * the process may run on -- its affinity mask, capped by the container's CPU quota and at 64).  A pipelined caller that
 * loads one block on one thread while it writes another on a second gives each its share, so that the two together do
 * not oversubscribe the cores. */
void pc_io_set_thread_limit(int nthreads);

/* Output writer: the byte-level half of nanopore_read.py:97-147 (get_fasta /
 * get_fastq) and porechop.py:607-734 (output_reads).  The caller has decided which pieces of which
 * reads go where; piece k is bases [piece_start[k], piece_start[k] + piece_len[k]) of read
 * piece_read[k] (coordinates in the whole read), written to file_paths[piece_file[k]] in the
 * order given.  piece_number[k] > 0 appends "_<number>" to the read name the way
 * add_number_to_read_name does (nanopore_read.py:494-498); piece_number may be NULL.  FASTQ:
 * "@name\nseq\n+\nquals\n" (reads that came from FASTA get '+' qualities, as NanoporeRead pads
 * them); FASTA: ">name\n" + sequence wrapped at 70.  RNA reads are written with U for T.  A path
 * of "-" is stdout.  Files are created when their first piece arrives (a bin that receives
 * nothing leaves no file, like the reference). */
int pc_readset_write(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                     const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                     const char *const *file_paths, int fastq, int64_t *bytes_written);

/* pc_readset_write for a streamed run: file_pos[f] is where file f continues (0 = create / truncate it now) and is
 * updated to where it ends.  A "-" (stdout) path is simply appended to. */
int pc_readset_write_at(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                        const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                        const char *const *file_paths, int fastq, int64_t *file_pos);

/* A sharded run (one process per GPU) over ONE plain FASTQ (or FASTA) file: every rank parses only its own byte range and
 * writes its own span of the shared output files.
 * pc_fastq_find_record: the first record start at or after byte_pos (the cut pc_readset_load_segment would choose), or the
 * file size when none is left; rank r of W takes [find(size * r / W), find(size * (r + 1) / W)).  "Not streamable" (gzip,
 * an irregular record near byte_pos) is PC_ERR_UNSUPPORTED_SCORES, as for pc_readset_load_segment.
 * pc_readset_write_sizes: the bytes pc_readset_write would put into each file, nothing written -- the ranks exchange them
 * and take the prefix sums as their positions.
 * pc_readset_write_shared: pc_readset_write_at for files other processes write disjoint spans of: opened without
 * truncation whatever the position (create / truncate them once, before any rank writes). */
int pc_fastq_find_record(const char *path, int64_t byte_pos, int64_t *record_start);
int pc_readset_write_sizes(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                           const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                           int fastq, int64_t *bytes_per_file);
int pc_readset_write_shared(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                            const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                            const char *const *file_paths, int fastq, int64_t *file_pos);

/* ---- gzip at the speed of the rest (replaces: Python's gzip module on the way in, porechop/misc.py:60-81,151-168; `pigz -p
 * <threads>` / gzip over a temporary file on the way out, porechop/porechop.py:640-651,685-729) -------------------------------
 * Output is a chain of independent gzip members of <= 65 280 input bytes, each carrying its compressed size in a 'BC' extra
 * subfield (the BGZF layout of the SAM specification, 4.1): an ordinary multi-member .gz file to gunzip / zlib / Python, and
 * one whose members any reader that knows the subfield inflates in parallel.  DEFLATE comes from libdeflate when the
 * machine has it (dlopen), else zlib.
 * pc_readset_compress: the pieces pc_readset_write would write, formatted and deflated by all cores into memory (nothing is
 *   opened); level 1..9 (<= 0: the default -- libdeflate's 3, which already makes smaller files than the zlib level 6 of
 *   gzip and pigz; zlib's 6 without libdeflate).
 * pc_gzimage_sizes / pc_gzimage_write: the bytes per file, and the image of every file written at file_pos[f] (updated) --
 *   a streamed run appends block after block, the ranks of a sharded run exchange the sizes first and write disjoint spans
 *   (shared = 1: never truncate).  Files whose image is empty are not touched.
 * pc_gz_finish: the empty last member (28 bytes) that ends such a file; creates the file when there is none.
 * pc_gzip_file: a whole file, src -> dst, all cores (single_member = 1: ONE member the way pigz builds it, which nobody can
 *   inflate in parallel -- for tests and benchmarks of the reader's route for ordinary .gz input). */
typedef struct pc_gzimage pc_gzimage;
int pc_readset_compress(const pc_readset *rs, int64_t npieces, const int64_t *piece_read, const int32_t *piece_start,
                        const int32_t *piece_len, const int32_t *piece_number, const int32_t *piece_file, int nfiles,
                        int fastq, int level, pc_gzimage **out);
int pc_gzimage_sizes(const pc_gzimage *img, int nfiles, int64_t *compressed_bytes, int64_t *plain_bytes);
int pc_gzimage_write(const pc_gzimage *img, int nfiles, const char *const *file_paths, int64_t *file_pos, int shared);
void pc_gzimage_free(pc_gzimage *img);
int pc_gz_finish(const char *path);
int pc_gzip_file(const char *src, const char *dst, int level, int single_member);

/* A gzip FASTQ file made of SIZED members (what the writer above makes; bgzip) for the ranks of a sharded run: the inflated
 * stream is addressed like a plain file -- position x lies in the member whose inflated span holds it -- so that rank r of W
 * takes the records that start in [find(total * r / W), find(total * (r + 1) / W)) and inflates only the members that hold
 * them (the counterparts of pc_fastq_find_record / pc_readset_load_segment; porechop/misc.py:151-168 is what they replace).
 * pc_gz_sized_size: the inflated size; PC_ERR_UNSUPPORTED_SCORES for a file that is not made of sized members throughout.
 * pc_gz_sized_find_record: the first record start at or after `pos` of the inflated bytes (the inflated size when none is left).
 * pc_readset_load_gz_range: the records of [begin, end), both as pc_gz_sized_find_record gives them. */
int pc_gz_sized_size(const char *path, int64_t *inflated_bytes);
int pc_gz_sized_find_record(const char *path, int64_t pos, int64_t *record_start);
int pc_readset_load_gz_range(const char *path, int64_t begin, int64_t end, pc_readset **out);

/* A gzip FASTQ file as a stream of blocks (the streamed route for .gz input): a producer thread inflates ahead of the
 * caller -- members that carry their size on several cores, members without one guessed and inflated ahead, ONE big member by
 * a single libdeflate call whose output is handed over while it appears (zlib without libdeflate) -- so that inflating block
 * k+1 overlaps parsing, scanning and writing block k.
 * pc_gzstream_next: the records that start before the first record start at or after target_bytes of the bytes not yet
 *   handed out (where pc_readset_load_segment would cut the plain file), at least min_reads of them unless the file ends
 *   first; *out = NULL and *eof = 1 after the last block.  PC_ERR_UNSUPPORTED_SCORES = "not streamable" (not gzip, not a
 *   regular 4-line FASTQ, a damaged stream): load the whole file with pc_readset_load, which reproduces the reference's
 *   behaviour and messages. */
typedef struct pc_gzstream pc_gzstream;
int pc_gzstream_open(const char *path, pc_gzstream **out);
/* One rank's share of a multi-member gzip file WITHOUT sizes (`cat *.fastq.gz`): pc_gz_member_start = the first member that
 * starts at or after compressed byte pos (validated: the stream behind the magic inflates to its end and CRC-32 / ISIZE agree;
 * the file's size when there is none); pc_gzstream_open_range = the stream of pc_gzstream_open over the members in
 * [begin, end) of the compressed bytes (both member starts; end <= 0: to the end of the file). */
int pc_gz_member_start(const char *path, int64_t pos, int64_t *member_start);
int pc_gzstream_open_range(const char *path, int64_t begin, int64_t end, pc_gzstream **out);
int pc_gzstream_next(pc_gzstream *s, int64_t target_bytes, int64_t min_reads, pc_readset **out, int *eof);
void pc_gzstream_close(pc_gzstream *s);

#ifdef __cplusplus
}
#endif
#endif /* PORECHOP_AMD_H */
