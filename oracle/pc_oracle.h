/* pc_oracle.h -- C interface of the CPU restatement (TEST INFRASTRUCTURE ONLY; see pc_oracle.c). */
#ifndef PC_ORACLE_H
#define PC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int read_start, read_end, adapter_start, adapter_end, score;
    int aligned_matches, aligned_len, full_matches, full_len;
    int path_len;
    int failed; /* 1 => reference prints read_start == -1 (empty input) */
    int end_i, end_j; /* the scout's end cell: adapter bases / read columns consumed where the traceback starts
                         (dp_scout.h:165-179) -- what the GPU's score-only records carry, and what the bounds of
                         the pruned phase B are derived from (tests/test_phase_b_bounds.py) */
} pc_oracle_result;

/* returns 0 on success, -1 on allocation failure.  gap_open == gap_extend follows the reference's
   linear-gap (NeedlemanWunsch) dispatch. */
int pc_oracle_align_raw(const char *read, int n, const char *adapter, int m,
                        int match, int mismatch, int gap_open, int gap_extend,
                        pc_oracle_result *res);
int pc_oracle_format(const pc_oracle_result *r, char *buf, size_t buflen);
char *pc_oracle_adapterAlignment(const char *readSeq, const char *adapterSeq, int matchScore,
                                 int mismatchScore, int gapOpenScore, int gapExtensionScore);
void pc_oracle_free(char *p);
int pc_oracle_align_many(const char *read_arena, const int64_t *read_off, const int32_t *read_len,
                         const char *adapter_arena, const int64_t *ad_off, const int32_t *ad_len,
                         int64_t npairs, int match, int mismatch, int gap_open, int gap_extend,
                         int32_t *out9);
/* smallest unit-cost edit distance between the whole adapter and any substring of the read (checker of the
   exact prefilter, see pc_oracle.c) */
int pc_oracle_min_edits(const char *read, int n, const char *adapter, int m);
int pc_oracle_min_edits_many(const char *read_arena, const int64_t *read_off, const int32_t *read_len,
                             const char *adapter, int m, int64_t n, int32_t *out);
#ifdef __cplusplus
}
#endif
#endif
