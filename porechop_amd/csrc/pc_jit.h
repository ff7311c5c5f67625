// pc_jit.h -- run-time specialised score-only scan kernels (see pc_jit_source.h / pc_jit.cpp).
#pragma once
#include <stdint.h>

#include <string>

#include "pc_kernels.h"

namespace pcj {

struct Spec {
    int R = 0, K = 0, m_lo = 0, m_hi = 0;
    int waves = 2;               // resident waves per SIMD the register allocation is asked to allow
    int blocks_per_cu = 8;       // resident 64-thread workgroups per CU (runtime occupancy query: registers and LDS)
    bool f16 = false;            // packed-fp16 variant (5 ops per cell pair) vs packed-int16 (6)
    void *module = nullptr, *function = nullptr;
    void *d_table = nullptr;     // device [25 code pairs][K] substitution-term table
};

// must match `struct SpecArgs` inside the JIT source
struct SpecArgs {
    const uint8_t *arena; const int64_t *win_off; const int32_t *win_len;
    const pck::Tile *tiles; int32_t ntiles;
    int32_t *out;
    void *fin_scratch;
    const uint32_t *s_table;
    int32_t m_lo, m_hi, gap_open, gap_extend;
    int32_t chunks, chunk_len, span;
    uint32_t *err;
    uint32_t *work_counter;      // units beyond the grid are handed out by this counter (zeroed by the host); null = fixed stride
    uint32_t one2;               // 0x00010001 (set by launch())
    int32_t *rec_out;            // PC_MODE_SCORE with whole windows (chunks == 1): the pairs' 8-int score records, written by the
                                 // kernel itself -- (-2, J, I, 0, score, 0, 0, 0) at rec_out[p * 8] -- instead of by the planner
    const int32_t *unit_prefix;  // [ntiles + 1] or null: the real (tile, chunk) units only (pck::launch_unit_prefix)
};

bool disabled();
// nullptr when specialisation is unavailable for this pair, or not yet worth its compile: `cells`
// (the launch's work) is added to the pair's running total, and the kernel is compiled once that
// total reaches PC_JIT_MIN_CELLS (caller uses the generic kernels until then).
// int16_only: the packed-int16 variant (6 ops per cell pair) even where packed fp16 is exact (pc_set_int16_only).
Spec *get(int device, const std::string &ad_lo, const std::string &ad_hi, int match, int mismatch, int gap_open,
          int gap_extend, double cells, bool int16_only = false);
int launch(const Spec *sp, const SpecArgs &a, int grid, void *stream);
// Ahead-of-time build of one pair's kernel into a kernel cache directory ("" = the in-tree one next to the library).
// Needs hiprtc, not a device.  0 = compiled and written, 1 = was already there, < 0 = no such kernel / failure.
int precompile(const std::string &ad_lo, const std::string &ad_hi, int match, int mismatch, int gap_open, int gap_extend,
               const std::string &dir);
// kernels compiled by hiprtc / taken from a kernel cache on disk by this process so far
void stats(long *compiled, long *from_disk);
// Asynchronous specialisation: compiles run on a worker thread and launches keep using the generic
// kernels until a kernel is ready (no stall for one-shot runs); default off (compile in place).
void set_async(int on);
// Join the worker threads (dropping compiles that have not started when cancel_queued).
void wait_idle(bool cancel_queued);

}  // namespace pcj
