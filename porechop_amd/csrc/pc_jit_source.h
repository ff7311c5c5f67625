// pc_jit_source.h -- source of the adapter-SPECIALISED score-only scan kernel, compiled at run
// time with hiprtc for one (adapter_lo, adapter_hi, scoring scheme) (pc_jit.cpp).
//
// Two things make it cheaper than the generic kernels' 9 packed ops per cell pair:
//
//  1. Adapter known at compile time.  The per-column substitution terms of the (few) distinct
//     letter pairs k that occur in the adapter pair are fetched once per column from an LDS table
//     indexed by the read byte, and every row's diagonal is ONE op, d = T_diag + S[COMBO[r]] with
//     a static register index (generic: 4 ops, because the adapter base of a row is data).
//
//  2. Drifting coordinates.  Every DP value X(rho, j) (rho = register row 1..R, j = column) is
//     kept as  X~ = X + (rho + jj) * eps - C,  eps = -gap_extend, jj = columns since the last
//     renormalisation, C a centring constant; T = M + gap_open is kept one step ahead,
//     T~ = T + (rho + jj + 1) * eps - C.  A gap extension moves one row or one column and costs
//     -eps, so in these coordinates it is free:
//         H~(rho,j) = max(H~(rho,j-1), T~(rho,j-1))          (no  H + e)
//         V~(rho,j) = max(V~(rho-1,j), T~(rho-1,j))          (no  V + e)
//         d~        = T~(rho-1,j-1) + S~,   S~ = sub - open + eps   (folded into the table)
//         M~        = max3(d~, H~, V~);     T~ = M~ + (open + eps)
//     5 packed ops per cell pair with fp16's 3-input max (v_pk_maximum3_f16), 6 with int16.  The
//     values are the same integers as the reference's, only offset by a known amount, so results
//     are bit-identical: row 0 (M = 0) becomes the per-column scalar `top`, the tracked last-row
//     score is T~(R,j) - top(j) = M(R,j) + R*eps, compared with strict '>' like dp_scout.h.
//     Values drift up by eps per column; every PC_KREN columns the state is shifted back down
//     (2R+1 ops per PC_KREN columns).  PC_KREN and C are chosen by the host so that every value
//     ever formed is an integer fp16 (|v| <= 2048) or int16 represents exactly (pc_jit.cpp).
//
//  3. Two columns per wave, skewed by three rows (column2 below): two independent dependency chains
//     interleaved instruction by instruction instead of one chain padded with wait states.
//
// Padding rows (shorter adapter of a pair) use a letter whose substitution score is 0, which
// keeps M = 0 like row 0.  The kernel is score-only (pass 1 of the whole-read scan) and
// bit-identical in outputs to scan_kernel<R, *, false>; tests run both (PC_DISABLE_JIT=1 selects
// the generic one).
#pragma once

namespace pcj {

// defines prepended by pc_jit.cpp: PC_DUAL (1: the two adapters differ), PC_R, PC_K (multiple of 4), PC_COMBO_INIT (R comma-separated
// ints), PC_F16, PC_EPS (= -gap_extend), PC_OE (= gap_open + PC_EPS), PC_CEN (C), PC_KREN,
// PC_WAVES (resident waves per SIMD the register allocation must allow), PC_CHECK_RANGE (0/1)
static const char *kSpecSource = R"PCJIT(
typedef unsigned int u32;
typedef long long i64;
#if PC_F16
// Packed FP16: every value formed is an integer with |v| <= 2048 (host-side gate), which fp16
// holds and adds exactly; -infinity is the real one.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 SV(u32 x) { return __builtin_bit_cast(h16x2, x); }
__device__ __forceinline__ u32 WV(h16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pack2x(int l, int h) { const h16x2 v = {(_Float16)l, (_Float16)h}; return WV(v); }
__device__ __forceinline__ int lo16(u32 x) { return (int)(float)SV(x).x; }
__device__ __forceinline__ int hi16(u32 x) { return (int)(float)SV(x).y; }
#define PC_NEGBITS 0xFC00FC00u
#define PC_POSBITS 0x7C007C00u
#define PC_ADD "v_pk_add_f16"
#define PC_MAX "v_pk_max_f16"
#else
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 SV(u32 x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u32 WV(s16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pack2x(int l, int h) { return ((u32)l & 0xFFFFu) | ((u32)h << 16); }
__device__ __forceinline__ int lo16(u32 x) { return (int)(short)(x & 0xFFFFu); }
__device__ __forceinline__ int hi16(u32 x) { return (int)(short)(x >> 16); }
#define PC_NEGBITS 0x80008000u
#define PC_POSBITS 0x7FFF7FFFu
#define PC_ADD "v_pk_add_u16"
#define PC_MAX "v_pk_max_i16"
#endif
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return WV(SV(a) + SV(b)); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return WV(SV(a) - SV(b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return WV(__builtin_elementwise_max(SV(a), SV(b))); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { return WV(__builtin_elementwise_min(SV(a), SV(b))); }
__device__ __forceinline__ u32 pack2(int v) { return pack2x(v, v); }
#if PC_F16
// IEEE-754-2019 maximum = v_pk_maximum3_f16: no canonicalising op on operands the compiler cannot prove quiet (never NaN here)
__device__ __forceinline__ u32 pk_maxq(u32 a, u32 b) { return WV(__builtin_elementwise_maximum(SV(a), SV(b))); }
#else
__device__ __forceinline__ u32 pk_maxq(u32 a, u32 b) { return pk_max(a, b); }
#endif
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 UV16(u32 x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ u32 WV16(u16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_minu(u32 a, u32 b) { return WV16(__builtin_elementwise_min(UV16(a), UV16(b))); }
__device__ __forceinline__ u32 pk_subu(u32 a, u32 b) { return WV16(UV16(a) - UV16(b)); }
__device__ __forceinline__ u32 pk_madu(u32 a, u32 k, u32 c) { return WV16(UV16(a) * UV16(k) + UV16(c)); }
// packed "best so far" in the tracked-score domain (score + R*eps); the chunk>0 sentinel maps to -inf
__device__ __forceinline__ u32 packbest(int l, int h)
{
    const u32 v = pack2x(l > -20000 ? l + PC_R * PC_EPS : 0, h > -20000 ? h + PC_R * PC_EPS : 0);
    return (l > -20000 ? (v & 0xFFFFu) : (PC_NEGBITS & 0xFFFFu)) | (h > -20000 ? (v & 0xFFFF0000u) : (PC_NEGBITS & 0xFFFF0000u));
}

struct Tile { i64 win_lo, win_hi, out_lo, out_hi; int count_lo, count_hi, adapter_lo, adapter_hi, rows, pad_; };
struct SpecArgs {
    const unsigned char *arena; const i64 *win_off; const int *win_len;
    const Tile *tiles; int ntiles;
    int *out;                 // [npairs * chunks][4]: score, I, J, 0
    uint2 *fin_scratch;       // [grid][R][64]: row-major, lane fastest -- a store / load instruction of a wave is 512 contiguous bytes
    const u32 *s_table;       // [25 = code_lo * 5 + code_hi][PC_K]
    int m_lo, m_hi, gap_open, gap_extend;
    int chunks, chunk_len, span;
    u32 *err;
    u32 *work_counter;
    u32 one2;                 // 0x00010001, kept opaque to the compiler (with a literal it turns min_u16(x, 1) into compare / select chains)
    int *rec_out;             // score-only request over whole windows: [npairs][8] records (-2, J, I, 0, score, 0, 0, 0), else null
    const int *unit_prefix;   // [ntiles + 1] or null: tile t owns units [unit_prefix[t], unit_prefix[t + 1]) -- the chunks that hold columns
                              // of its longest window (a 4 Mb read among 20 kb reads: 2 048 chunks per tile, ten of them real for most)
};
struct FastT { static constexpr bool fast = true; };
struct SlowT { static constexpr bool fast = false; };

__device__ static const unsigned char COMBO[PC_R] = { PC_COMBO_INIT };

typedef u32 u32_unaligned __attribute__((aligned(1)));

extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PC_WAVES))) void pc_spec_score(SpecArgs a)
{
    constexpr int R = PC_R, K = PC_K;
    // substitution terms by (code of the low stream's base, code of the high stream's base): 25 rows of K
    // packed terms, 1-4 KB instead of a row per byte value -- LDS no longer limits the waves per CU --
    // and the two bytes of a column reach their row through two small byte -> row-offset tables
    __shared__ uint4 s_tab[25 * K / 4];
    __shared__ unsigned short lut_lo[256], lut_hi[256];      // row offsets in uint4 units
#if PC_DUAL
    // two DIFFERENT adapters only ever share a tile whose halves read the same windows (pc_api.cpp
    // build_tiles): one byte stream per lane, and the byte -> table row lookup is a single read
    __shared__ unsigned short lut_one[256];
#endif
    const int lane = threadIdx.x;
    for (int i = lane; i < 25 * K / 4; i += 64) s_tab[i] = ((const uint4 *)a.s_table)[i];
    for (int c = lane; c < 256; c += 64) {
        const int code = (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2
                       : (c == 'T' || c == 't' || c == 'U' || c == 'u') ? 3 : 4;
        lut_lo[c] = (unsigned short)(code * 5 * (K / 4));
        lut_hi[c] = (unsigned short)(code * (K / 4));
#if PC_DUAL
        lut_one[c] = (unsigned short)(code * 6 * (K / 4));
#endif
    }
    __syncthreads();
    const u32 OE2 = pack2(PC_OE), EPS2 = pack2(PC_EPS), NEG2 = PC_NEGBITS;
    const int pad_lo = R - a.m_lo, pad_hi = R - a.m_hi;
    // previous-column state of the lanes whose read ends (one region per half: in a tile of two read
    // streams the halves of a lane end in different columns)
    uint2 *fin = a.fin_scratch + (i64)blockIdx.x * 2 * R * 64;
    uint2 *fin_hi_buf = fin + (PC_DUAL ? 0 : R * 64);
    const int nchunks = a.chunks > 1 ? a.chunks : 1;

    // Units (tile x column chunk) are taken in launch order -- the host hands tiles over longest first -- the
    // first gridDim.x by the hardware dispatcher, the rest from a counter: a workgroup that drew short units
    // simply draws more of them, so reads of very different lengths still fill the chip to the end.
    auto next_unit = [&](int vt) -> int {
        if (!a.work_counter) return vt + (int)gridDim.x;
        u32 v = 0;
        if (lane == 0) v = atomicAdd(a.work_counter, 1u);
        return (int)gridDim.x + (int)__builtin_amdgcn_readfirstlane(v);
    };
    const int total_units = a.unit_prefix ? a.unit_prefix[a.ntiles] : a.ntiles * nchunks;
    for (int vt = blockIdx.x; vt < total_units; vt = next_unit(vt)) {
        int t, chunk;
        if (a.unit_prefix) {                                     // the last tile whose first unit is <= vt (wave-uniform)
            int lo = 0, hi = a.ntiles - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.unit_prefix[mid] <= vt) lo = mid; else hi = mid - 1; }
            t = lo; chunk = vt - a.unit_prefix[lo];
        } else {
            t = vt / nchunks; chunk = vt - t * nchunks;
        }
        const Tile tile = a.tiles[t];
#if PC_DUAL
        constexpr bool one_stream = true;
        if (tile.win_lo != tile.win_hi) {                        // host bug
            if (lane == 0) atomicAdd(a.err, 1u);
            continue;
        }
#else
        const bool one_stream = tile.win_lo == tile.win_hi;
#endif
        const i64 p_lo = tile.out_lo + lane, p_hi = tile.out_hi + lane;
        const bool have_lo = lane < tile.count_lo, have_hi = lane < tile.count_hi;
        const unsigned char *w_lo = a.arena + (have_lo ? a.win_off[tile.win_lo + lane] : 0);
        const unsigned char *w_hi = a.arena + (have_hi ? a.win_off[tile.win_hi + lane] : 0);
        int n_lo = have_lo ? a.win_len[tile.win_lo + lane] : 0;
        int n_hi = have_hi ? a.win_len[tile.win_hi + lane] : 0;
        int c0_lo = 0, c0_hi = 0, tf_lo = 0, tf_hi = 0;
        bool tail_lo = true, tail_hi = true;
        if (nchunks > 1) {
            const int L = a.chunk_len, start = chunk * L;
            auto cut = [&](int nfull, const unsigned char *&w, int &n, int &c0, int &tf, bool &tail) {
                if (start >= nfull) { n = 0; c0 = 0; tf = 0; tail = false; return; }
                c0 = start - a.span > 0 ? start - a.span : 0;
                const int end = start + L < nfull ? start + L : nfull;
                w += c0; n = end - c0; tf = start - c0; tail = (end == nfull);
            };
            cut(n_lo, w_lo, n_lo, c0_lo, tf_lo, tail_lo);
            cut(n_hi, w_hi, n_hi, c0_hi, tf_hi, tail_hi);
        }
        // column 0: M = 0 (T = open) for a window at the read's column 0, otherwise the lower-bound
        // state "row-0 start + vertical gap" (see pc_bounds.h); drifting coordinates with jj = 0
        u32 T[R], U[R];
#pragma clang loop unroll(full)
        for (int r = 0; r < R; ++r) {
            const int vl = (c0_lo > 0 && r >= pad_lo) ? 2 * a.gap_open + (r - pad_lo) * a.gap_extend : a.gap_open;
            const int vh = (c0_hi > 0 && r >= pad_hi) ? 2 * a.gap_open + (r - pad_hi) * a.gap_extend : a.gap_open;
            T[r] = pack2x(vl + (r + 2) * PC_EPS - PC_CEN, vh + (r + 2) * PC_EPS - PC_CEN);
            U[r] = NEG2;
        }
        u32 top = pack2(a.gap_open + PC_EPS - PC_CEN);      // T~(0, j-1) entering column j
        int bs_lo = 0, bi_lo = a.m_lo, bj_lo = 0, bs_hi = 0, bi_hi = a.m_hi, bj_hi = 0;
        if (chunk > 0) { bs_lo = -32768; bj_lo = -1; bs_hi = -32768; bj_hi = -1; }   // (m,0) belongs to chunk 0
        u32 best2 = packbest(bs_lo, bs_hi);
        u32 ftop = 0;                                       // T~(0, n-1) of each half when its read ended
        // In the block-resolved path the running maximum and its column live PACKED (best2: the tracked
        // score term, pos2: the column as two u16; 0xFFFF = none yet): a new maximum costs five packed ops
        // per column for both halves instead of two conversions and two compare / select chains.  The int
        // copies (bs, bj) are brought up to date where the per-stream path or the output needs them.
        u32 pos2 = ((u32)bj_lo & 0xFFFFu) | ((u32)bj_hi << 16);
        bool packed_ahead = false;                          // wave-uniform: (best2, pos2) are newer than (bs, bj)
        auto unpack_best = [&]() {
            bs_lo = ((best2 & 0xFFFFu) == (PC_NEGBITS & 0xFFFFu)) ? -32768 : lo16(best2) - R * PC_EPS;
            bs_hi = ((best2 >> 16) == (PC_NEGBITS >> 16)) ? -32768 : hi16(best2) - R * PC_EPS;
            bj_lo = ((pos2 & 0xFFFFu) == 0xFFFFu) ? -1 : (int)(pos2 & 0xFFFFu);
            bj_hi = ((pos2 >> 16) == 0xFFFFu) ? -1 : (int)(pos2 >> 16);
            bi_lo = a.m_lo; bi_hi = a.m_hi;                 // every cell tracked inside the column loop is a last-row cell
            packed_ahead = false;
        };
        const u32 ONE2 = a.one2;
        // wave-uniform extents: columns in (tfmax, nmin) are tracked by every stream of the tile
        int nmax = n_lo > n_hi ? n_lo : n_hi;
        int nmin = have_lo ? n_lo : 0x7FFFFFFF;
        if (have_hi && n_hi < nmin) nmin = n_hi;
        int tfmax = tf_lo > tf_hi ? tf_lo : tf_hi;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            int o = __shfl_xor(nmax, s); nmax = o > nmax ? o : nmax;
            o = __shfl_xor(nmin, s); nmin = o < nmin ? o : nmin;
            o = __shfl_xor(tfmax, s); tfmax = o > tfmax ? o : tfmax;
        }
        // Every stream of the tile ends in the SAME column (equal-length windows: the 150-column end windows of phase B's
        // score pass, the uniform reads of the benchmark): the column loop then stops exactly there and the last column's
        // cells are scanned straight from the registers.  Otherwise (ragged tiles) a lane that ends parks the state its last
        // column starts from in fin_scratch and the column is re-run from there after the loop.  (The parked state costs
        // R*64*8 B per half and tile, out and back in: 37 GB per step of a 196-job phase B, for nothing.)
        const bool uniform_fin = nmax > 0 && nmin == nmax && __all((!have_lo || tail_lo) && (!have_hi || tail_hi));

        // Read bytes are fetched 16 columns at a time into q_lo / q_hi and handed to the column loop a dword
        // (4 columns) at a time, one dword ahead; the substitution terms S[] of column j+1 are fetched from
        // the LDS table while column j computes (two register sets, the column loop statically unrolled by
        // 4), so neither the global load nor the LDS read latency sits on the column's critical path.
        // Why 16 bytes: every lane reads its own window, so a load instruction touches 64 (two streams: 128)
        // different cache lines, and the lines a CU's waves have open (12 waves x 128 x 128 B) are far more
        // than its L1 holds -- each load is an L2 request per lane.  With a dword per load a tile of two read
        // streams asked L2 for 32 x the bytes it used and ran at a third of the one-stream tiles' rate
        // (measured, 150-column windows: 3.6 vs 9.8 TCUPS); 16 bytes per load is a quarter of the requests.
        auto load_q = [&](const unsigned char *w, int n, int col) -> uint4 {   // the 16 columns from 0-based column col
            const int k = (col < n) ? col : (n > 0 ? ((n - 1) & ~15) : 0);     // finished streams re-read their last block
            uint4 v;
            __builtin_memcpy(&v, w + k, 16);                                   // any alignment: one global_load_dwordx4
            return v;
        };
        auto pick_dw = [&](const uint4 &q, int k) -> u32 {                     // k is wave-uniform
            return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w;
        };
#if PC_CHECK_RANGE
        // debug build (PC_JIT_CHECK_RANGE=1): every column takes the one-column path and the extremes of EVERY value
        // formed are recorded -- the column state T / U after each column, every row's diagonal term d and vertical
        // state V (the cell maximum M is one of d, H, V; the new T is recorded with the state), the top-row term, the
        // tracked last-row term the scout compares, the last-column re-run's values -- the host-side range gate
        // (pc_bounds.h spec_plan) asserted on the device.  (The substitution-table terms are bounded on the host, where
        // the table is built: pc_jit.cpp.)  The "-infinity" a column's H / V start from is absorbed by the first max.
        u32 vmax = PC_NEGBITS, vmin = PC_POSBITS;
#define PC_NOTE_V(x) { vmax = pk_max(vmax, (x)); vmin = pk_min(vmin, (x)); }
#else
#define PC_NOTE_V(x)
#endif
        auto fetch_S = [&](u32 (&S)[K], u32 bl, u32 bh) {
#if PC_DUAL
            const uint4 *row = s_tab + (u32)lut_one[bl];
#else
            const uint4 *row = s_tab + ((u32)lut_lo[bl] + (u32)lut_hi[one_stream ? bl : bh]);
#endif
#pragma clang loop unroll(full)
            for (int q = 0; q < K / 4; ++q) { const uint4 v = row[q]; S[4*q] = v.x; S[4*q+1] = v.y; S[4*q+2] = v.z; S[4*q+3] = v.w; }
        };
        // One column.  Returns the tracked last-row term T~(R,j) - top(j) = M(R,j) + R*eps.
        // FastT: every stream of the tile tracks this column and none ends in it (the caller
        // resolves maxima per 4-column block); SlowT: per-stream masks, last-column handling.
        auto column = [&](auto tag, const int j, const u32 (&S)[K]) -> u32 {
            constexpr bool FAST = decltype(tag)::fast;
            bool fin_lo = false, fin_hi = false, any_fin = false;
            if constexpr (!FAST) {
                // A read's LAST column is scanned row by row (every cell of it is an end candidate,
                // dp_scout.h:165-179) -- after the column loop, for all lanes of the tile at once: here the
                // lanes that end in this column only park the state the column starts from.
                fin_lo = (j == n_lo) && tail_lo && !uniform_fin; fin_hi = (j == n_hi) && tail_hi && !uniform_fin;
                any_fin = __any(fin_lo || fin_hi);
                if (any_fin) {
                    if (fin_lo) {
#pragma clang loop unroll(full)
                        for (int r = 0; r < R; ++r) fin[r * 64 + lane] = make_uint2(T[r], U[r]);
                    }
#if !PC_DUAL
                    if (fin_hi) {
#pragma clang loop unroll(full)
                        for (int r = 0; r < R; ++r) fin_hi_buf[r * 64 + lane] = make_uint2(T[r], U[r]);
                    }
#endif
                    ftop = (fin_lo ? (top & 0xFFFFu) : (ftop & 0xFFFFu)) | (fin_hi ? (top & 0xFFFF0000u) : (ftop & 0xFFFF0000u));
                }
            }
            const u32 topn = pk_add(top, EPS2);               // T~(0, j)
            // ---- the column, hand-scheduled.  gfx950 needs a wait state between dependent packed
            // ops; each row's serial chain (V, M, T') is interleaved with row r+2's independent half
            // (H in place into U[q], d into dh[q]); the one gap nothing can fill is an s_nop,
            // which the other waves of the SIMD use.
            {
                constexpr int KP = 2;
                u32 dh[R];     // d of the rows in flight
                auto ind_only = [&](int q, u32 diag) {
                    asm volatile(PC_MAX " %[uq], %[uq], %[tq]\n\t"
                                 PC_ADD " %[dq], %[diag], %[s]"
                                 : [dq] "=&v"(dh[q]), [uq] "+v"(U[q])
                                 : [diag] "v"(diag), [s] "v"(S[COMBO[q]]), [tq] "v"(T[q]));
                };
#pragma clang loop unroll(full)
                for (int q = 0; q < KP && q < R; ++q) ind_only(q, q == 0 ? top : T[q - 1]);
                u32 Tup = topn, Vprev = NEG2;
#if PC_F16
                // 5 ops: H' (in place), V, d', M = max3(d, H, V), T      (primed = row r+2)
#define PC_ROW_FULL(VP, TU, DIAG, SS, UQ, TQ, DHR, UR, TN, DHQ, VS)                  \
    PC_MAX " " UQ ", " UQ ", " TQ "\n\t"                                              \
    PC_MAX " " VS ", " VP ", " TU "\n\t"                                              \
    PC_ADD " " DHQ ", " DIAG ", " SS "\n\t"                                           \
    "v_pk_maximum3_f16 %[mn], " DHR ", " UR ", " VS "\n\t"                            \
    "s_nop 0\n\t"                                                                     \
    PC_ADD " " TN ", %[mn], %[oe]\n\t"
#define PC_ROW_TAIL                                                                    \
    "s_nop 0\n\t"                                                                      \
    PC_MAX " %[vs0], %[vprev], %[tup]\n\t" "s_nop 0\n\t"                               \
    "v_pk_maximum3_f16 %[mn], %[dhr0], %[ur0], %[vs0]\n\t" "s_nop 0\n\t"               \
    PC_ADD " %[tn0], %[mn], %[oe]"
#else
                // 6 ops: max(d,H), H' (in place), V, d', M = max(max(d,H), V), T
#define PC_ROW_FULL(VP, TU, DIAG, SS, UQ, TQ, DHR, UR, TN, DHQ, VS)                  \
    PC_MAX " %[mn], " DHR ", " UR "\n\t"                                              \
    PC_MAX " " UQ ", " UQ ", " TQ "\n\t"                                              \
    PC_MAX " " VS ", " VP ", " TU "\n\t"                                              \
    PC_ADD " " DHQ ", " DIAG ", " SS "\n\t"                                           \
    PC_MAX " %[mn], %[mn], " VS "\n\t"                                                \
    "s_nop 0\n\t"                                                                     \
    PC_ADD " " TN ", %[mn], %[oe]\n\t"
#define PC_ROW_TAIL                                                                    \
    PC_MAX " %[mn], %[dhr0], %[ur0]\n\t"                                               \
    PC_MAX " %[vs0], %[vprev], %[tup]\n\t" "s_nop 0\n\t"                               \
    PC_MAX " %[mn], %[mn], %[vs0]\n\t" "s_nop 0\n\t"                                   \
    PC_ADD " %[tn0], %[mn], %[oe]"
#endif
#pragma clang loop unroll(full)
                for (int r = 0; r < R; r += 2) {
                    u32 mn, vs0, vs1;
                    if (r + 1 + KP < R) {
                        const int q = r + KP;
                        asm volatile(
                            PC_ROW_FULL("%[vprev]", "%[tup]", "%[d0]", "%[s0]", "%[u0]", "%[t0]", "%[dhr0]", "%[ur0]", "%[tn0]", "%[dhq0]", "%[vs0]")
                            PC_ROW_FULL("%[vs0]", "%[tn0]", "%[t0]", "%[s1]", "%[u1]", "%[t1]", "%[dhr1]", "%[ur1]", "%[tn1]", "%[dhq1]", "%[vs1]")
                            : [mn] "=&v"(mn), [vs0] "=&v"(vs0), [vs1] "=&v"(vs1),
                              [u0] "+v"(U[q]), [u1] "+v"(U[q + 1]), [tn0] "=&v"(T[r]), [tn1] "=&v"(T[r + 1]),
                              [dhq0] "=&v"(dh[q]), [dhq1] "=&v"(dh[q + 1])
                            : [vprev] "v"(Vprev), [tup] "v"(Tup), [oe] "s"(OE2), [d0] "v"(T[q - 1]),
                              [t0] "v"(T[q]), [t1] "v"(T[q + 1]), [s0] "v"(S[COMBO[q]]), [s1] "v"(S[COMBO[q + 1]]),
                              [dhr0] "v"(dh[r]), [dhr1] "v"(dh[r + 1]), [ur0] "v"(U[r]), [ur1] "v"(U[r + 1]));
                        PC_NOTE_V(dh[r]) PC_NOTE_V(dh[r + 1]) PC_NOTE_V(vs0) PC_NOTE_V(vs1)
                        Tup = T[r + 1]; Vprev = vs1;
                    } else {
                        // tail rows (and an odd last row): one row at a time
#pragma clang loop unroll(full)
                        for (int rr = r; rr < r + 2 && rr < R; ++rr) {
                            if (rr + KP < R) {
                                const int q = rr + KP;
                                asm volatile(
                                    PC_ROW_FULL("%[vprev]", "%[tup]", "%[d0]", "%[s0]", "%[u0]", "%[t0]", "%[dhr0]", "%[ur0]", "%[tn0]", "%[dhq0]", "%[vs0]")
                                    : [mn] "=&v"(mn), [vs0] "=&v"(vs0),
                                      [u0] "+v"(U[q]), [tn0] "=&v"(T[rr]), [dhq0] "=&v"(dh[q])
                                    : [vprev] "v"(Vprev), [tup] "v"(Tup), [oe] "s"(OE2), [d0] "v"(T[q - 1]),
                                      [t0] "v"(T[q]), [s0] "v"(S[COMBO[q]]), [dhr0] "v"(dh[rr]), [ur0] "v"(U[rr]));
                            } else {
                                asm volatile(PC_ROW_TAIL
                                             : [vs0] "=&v"(vs0), [mn] "=&v"(mn), [tn0] "=&v"(T[rr])
                                             : [vprev] "v"(Vprev), [tup] "v"(Tup), [dhr0] "v"(dh[rr]),
                                               [ur0] "v"(U[rr]), [oe] "s"(OE2));
                            }
                            PC_NOTE_V(dh[rr]) PC_NOTE_V(vs0)
                            Tup = T[rr]; Vprev = vs0;
                        }
                    }
                }
#undef PC_ROW_FULL
#undef PC_ROW_TAIL
            }
            const u32 cand = pk_sub(T[R - 1], topn);
            PC_NOTE_V(cand) PC_NOTE_V(topn)
            if constexpr (!FAST) {
                const int cl = lo16(cand) - R * PC_EPS, ch = hi16(cand) - R * PC_EPS;
                const bool tr_lo = j > tf_lo && (tail_lo ? j < n_lo : j <= n_lo);
                const bool tr_hi = j > tf_hi && (tail_hi ? j < n_hi : j <= n_hi);
                if (tr_lo && cl > bs_lo) { bs_lo = cl; bi_lo = a.m_lo; bj_lo = j; }
                if (tr_hi && ch > bs_hi) { bs_hi = ch; bi_hi = a.m_hi; bj_hi = j; }
            }
            top = topn;
            return cand;
        };

        // Two columns at once, SKEWED: chain A walks column j, chain B column j+1 three rows behind,
        // on the same T/U registers (B reads what A has just produced and overwrites it in turn).
        // Each asm statement carries one row of A and one row of B interleaved instruction by
        // instruction, so every dependent pair of packed ops is separated by the other chain's op:
        // no s_nop, and the two chains give the VALU independent work (profiles/r01_ubench_row.txt:
        // 22.4 instead of 23.75 cycles per row at two waves per SIMD).  Only for blocks in which every
        // stream of the tile tracks the columns and none ends (the caller's fast path).
        auto column2 = [&](const u32 (&S1)[K], const u32 (&S2)[K], u32 &c1, u32 &c2) {
            constexpr int L = 3, KP = 2;
            const u32 topA = pk_add(top, EPS2), topB = pk_add(topA, EPS2);     // T~(0,j), T~(0,j+1)
            u32 dhA[R], dhB[R];
            auto ind = [&](u32 (&dh)[R], const u32 (&S)[K], int q, u32 diag) {
                asm volatile(PC_MAX " %[uq], %[uq], %[tq]\n\t"
                             PC_ADD " %[dq], %[diag], %[s]"
                             : [dq] "=&v"(dh[q]), [uq] "+v"(U[q])
                             : [diag] "v"(diag), [s] "v"(S[COMBO[q]]), [tq] "v"(T[q]));
            };
#if PC_F16
#define PC_ONE_FULL                                                                    \
    PC_MAX " %[uq], %[uq], %[tq]\n\t"                                                  \
    PC_MAX " %[vs], %[vp], %[tu]\n\t"                                                  \
    PC_ADD " %[dq], %[dg], %[s]\n\t"                                                   \
    "v_pk_maximum3_f16 %[mn], %[dr], %[ur], %[vs]\n\t" "s_nop 0\n\t"                   \
    PC_ADD " %[tn], %[mn], %[oe]"
#define PC_ONE_TAIL                                                                    \
    "s_nop 0\n\t"                                                                      \
    PC_MAX " %[vs], %[vp], %[tu]\n\t" "s_nop 0\n\t"                                    \
    "v_pk_maximum3_f16 %[mn], %[dr], %[ur], %[vs]\n\t" "s_nop 0\n\t"                   \
    PC_ADD " %[tn], %[mn], %[oe]"
#define PC_PAIR_FULL                                                                   \
    PC_MAX " %[uqa], %[uqa], %[tqa]\n\t"  PC_MAX " %[uqb], %[uqb], %[tqb]\n\t"         \
    PC_MAX " %[vsa], %[vpa], %[tua]\n\t"  PC_MAX " %[vsb], %[vpb], %[tub]\n\t"         \
    PC_ADD " %[dqa], %[dga], %[sa]\n\t"   PC_ADD " %[dqb], %[dgb], %[sb]\n\t"          \
    "v_pk_maximum3_f16 %[mna], %[dra], %[ura], %[vsa]\n\t"                             \
    "v_pk_maximum3_f16 %[mnb], %[drb], %[urb], %[vsb]\n\t"                             \
    PC_ADD " %[tna], %[mna], %[oe]\n\t"   PC_ADD " %[tnb], %[mnb], %[oe]"
#define PC_PAIR_ATAIL                                                                  \
    PC_MAX " %[uqb], %[uqb], %[tqb]\n\t"                                               \
    PC_MAX " %[vsa], %[vpa], %[tua]\n\t"  PC_MAX " %[vsb], %[vpb], %[tub]\n\t"         \
    PC_ADD " %[dqb], %[dgb], %[sb]\n\t"                                                \
    "v_pk_maximum3_f16 %[mna], %[dra], %[ura], %[vsa]\n\t"                             \
    "v_pk_maximum3_f16 %[mnb], %[drb], %[urb], %[vsb]\n\t"                             \
    PC_ADD " %[tna], %[mna], %[oe]\n\t"   PC_ADD " %[tnb], %[mnb], %[oe]"
#else
#define PC_ONE_FULL                                                                    \
    PC_MAX " %[mn], %[dr], %[ur]\n\t"                                                  \
    PC_MAX " %[uq], %[uq], %[tq]\n\t"                                                  \
    PC_MAX " %[vs], %[vp], %[tu]\n\t"                                                  \
    PC_ADD " %[dq], %[dg], %[s]\n\t"                                                   \
    PC_MAX " %[mn], %[mn], %[vs]\n\t" "s_nop 0\n\t"                                    \
    PC_ADD " %[tn], %[mn], %[oe]"
#define PC_ONE_TAIL                                                                    \
    PC_MAX " %[mn], %[dr], %[ur]\n\t"                                                  \
    PC_MAX " %[vs], %[vp], %[tu]\n\t" "s_nop 0\n\t"                                    \
    PC_MAX " %[mn], %[mn], %[vs]\n\t" "s_nop 0\n\t"                                    \
    PC_ADD " %[tn], %[mn], %[oe]"
#define PC_PAIR_FULL                                                                   \
    PC_MAX " %[mna], %[dra], %[ura]\n\t"  PC_MAX " %[mnb], %[drb], %[urb]\n\t"         \
    PC_MAX " %[uqa], %[uqa], %[tqa]\n\t"  PC_MAX " %[uqb], %[uqb], %[tqb]\n\t"         \
    PC_MAX " %[vsa], %[vpa], %[tua]\n\t"  PC_MAX " %[vsb], %[vpb], %[tub]\n\t"         \
    PC_ADD " %[dqa], %[dga], %[sa]\n\t"   PC_ADD " %[dqb], %[dgb], %[sb]\n\t"          \
    PC_MAX " %[mna], %[mna], %[vsa]\n\t"  PC_MAX " %[mnb], %[mnb], %[vsb]\n\t"         \
    PC_ADD " %[tna], %[mna], %[oe]\n\t"   PC_ADD " %[tnb], %[mnb], %[oe]"
#define PC_PAIR_ATAIL                                                                  \
    PC_MAX " %[mna], %[dra], %[ura]\n\t"  PC_MAX " %[mnb], %[drb], %[urb]\n\t"         \
    PC_MAX " %[uqb], %[uqb], %[tqb]\n\t"                                               \
    PC_MAX " %[vsa], %[vpa], %[tua]\n\t"  PC_MAX " %[vsb], %[vpb], %[tub]\n\t"         \
    PC_ADD " %[dqb], %[dgb], %[sb]\n\t"                                                \
    PC_MAX " %[mna], %[mna], %[vsa]\n\t"  PC_MAX " %[mnb], %[mnb], %[vsb]\n\t"         \
    PC_ADD " %[tna], %[mna], %[oe]\n\t"   PC_ADD " %[tnb], %[mnb], %[oe]"
#endif
            // one row of one chain on its own (prologue of A, epilogue of B)
            auto row_alone = [&](u32 (&dh)[R], const u32 (&S)[K], int rr, u32 &Vprev, u32 &Tup) {
                u32 mn, vs;
                if (rr + KP < R) {
                    const int q = rr + KP;
                    asm volatile(PC_ONE_FULL
                                 : [mn] "=&v"(mn), [vs] "=&v"(vs), [uq] "+v"(U[q]), [tn] "=&v"(T[rr]), [dq] "=&v"(dh[q])
                                 : [vp] "v"(Vprev), [tu] "v"(Tup), [oe] "s"(OE2), [dg] "v"(T[q - 1]), [tq] "v"(T[q]),
                                   [s] "v"(S[COMBO[q]]), [dr] "v"(dh[rr]), [ur] "v"(U[rr]));
                } else {
                    asm volatile(PC_ONE_TAIL
                                 : [mn] "=&v"(mn), [vs] "=&v"(vs), [tn] "=&v"(T[rr])
                                 : [vp] "v"(Vprev), [tu] "v"(Tup), [oe] "s"(OE2), [dr] "v"(dh[rr]), [ur] "v"(U[rr]));
                }
                Tup = T[rr]; Vprev = vs;
            };
            // chain A alone for its first L rows
#pragma clang loop unroll(full)
            for (int q = 0; q < KP && q < R; ++q) ind(dhA, S1, q, q == 0 ? top : T[q - 1]);
            u32 TupA = topA, VpA = NEG2;
#pragma clang loop unroll(full)
            for (int r = 0; r < L && r < R; ++r) row_alone(dhA, S1, r, VpA, TupA);
            // chain B starts: its rows 0 and 1 read column j's T(0), T(1), which A has produced
#pragma clang loop unroll(full)
            for (int q = 0; q < KP && q < R; ++q) ind(dhB, S2, q, q == 0 ? topA : T[q - 1]);
            u32 TupB = topB, VpB = NEG2;
#pragma clang loop unroll(full)
            for (int r = L; r < R; ++r) {
                const int pb = r - L, qb = pb + KP;
                u32 mna, mnb, vsa, vsb;
                if (r + KP < R) {
                    const int qa = r + KP;
                    asm volatile(PC_PAIR_FULL
                                 : [mna] "=&v"(mna), [mnb] "=&v"(mnb), [vsa] "=&v"(vsa), [vsb] "=&v"(vsb),
                                   [uqa] "+v"(U[qa]), [uqb] "+v"(U[qb]), [tna] "=&v"(T[r]), [tnb] "=&v"(T[pb]),
                                   [dqa] "=&v"(dhA[qa]), [dqb] "=&v"(dhB[qb])
                                 : [vpa] "v"(VpA), [tua] "v"(TupA), [vpb] "v"(VpB), [tub] "v"(TupB), [oe] "s"(OE2),
                                   [dga] "v"(T[qa - 1]), [tqa] "v"(T[qa]), [sa] "v"(S1[COMBO[qa]]), [dra] "v"(dhA[r]), [ura] "v"(U[r]),
                                   [dgb] "v"(T[qb - 1]), [tqb] "v"(T[qb]), [sb] "v"(S2[COMBO[qb]]), [drb] "v"(dhB[pb]), [urb] "v"(U[pb]));
                } else {
                    asm volatile(PC_PAIR_ATAIL
                                 : [mna] "=&v"(mna), [mnb] "=&v"(mnb), [vsa] "=&v"(vsa), [vsb] "=&v"(vsb),
                                   [uqb] "+v"(U[qb]), [tna] "=&v"(T[r]), [tnb] "=&v"(T[pb]), [dqb] "=&v"(dhB[qb])
                                 : [vpa] "v"(VpA), [tua] "v"(TupA), [vpb] "v"(VpB), [tub] "v"(TupB), [oe] "s"(OE2),
                                   [dra] "v"(dhA[r]), [ura] "v"(U[r]),
                                   [dgb] "v"(T[qb - 1]), [tqb] "v"(T[qb]), [sb] "v"(S2[COMBO[qb]]), [drb] "v"(dhB[pb]), [urb] "v"(U[pb]));
                }
                TupA = T[r]; VpA = vsa; TupB = T[pb]; VpB = vsb;
            }
            c1 = pk_sub(T[R - 1], topA);               // column j's last row, before chain B overwrites it
#pragma clang loop unroll(full)
            for (int pb = (R > L ? R - L : 0); pb < R; ++pb) row_alone(dhB, S2, pb, VpB, TupB);
            c2 = pk_sub(T[R - 1], topB);
            top = topB;
#undef PC_ONE_FULL
#undef PC_ONE_TAIL
#undef PC_PAIR_FULL
#undef PC_PAIR_ATAIL
        };
#if PC_CHECK_RANGE
        auto note_range = [&]() {
#pragma clang loop unroll(full)
            for (int r = 0; r < R; ++r) { vmax = pk_max(vmax, pk_max(T[r], U[r])); vmin = pk_min(vmin, T[r]); }
        };
#endif
        uint4 q_lo = load_q(w_lo, n_lo, 0), q_hi = one_stream ? make_uint4(0u, 0u, 0u, 0u) : load_q(w_hi, n_hi, 0);
        u32 cur_lo = q_lo.x, cur_hi = q_hi.x;
        u32 nxt_lo = q_lo.y, nxt_hi = q_hi.y;
        u32 SA[K], SB[K], SC[K], SD[K];
        fetch_S(SA, cur_lo & 0xFF, cur_hi & 0xFF);
        int jj = 0;                                           // columns since the last renormalisation
        for (int j0 = 1; j0 <= nmax; j0 += 4) {
            // the dword this block hands on at its end is the one two blocks ahead (0-based column j0 + 7); when
            // that opens a new 16-column block, the old one is used up -- its last dword is already in nxt -- and
            // the new one is requested here, a whole 4-column block of arithmetic before it is needed
            const int kq = ((j0 + 7) >> 2) & 3;
            if (kq == 0) {
                q_lo = load_q(w_lo, n_lo, j0 + 7);
                if (!one_stream) q_hi = load_q(w_hi, n_hi, j0 + 7);
            }
            if (jj >= PC_KREN) {
                const u32 DK = pack2(PC_KREN * PC_EPS);
#pragma clang loop unroll(full)
                for (int r = 0; r < R; ++r) { T[r] = pk_sub(T[r], DK); U[r] = pk_sub(U[r], DK); }
                top = pk_sub(top, DK);
                jj = 0;
            }
            if (!PC_CHECK_RANGE && j0 > tfmax && j0 + 3 < nmin) {
                u32 c0, c1, c2, c3;
                fetch_S(SB, (cur_lo >> 8) & 0xFF, (cur_hi >> 8) & 0xFF);
                fetch_S(SC, (cur_lo >> 16) & 0xFF, (cur_hi >> 16) & 0xFF);
                fetch_S(SD, cur_lo >> 24, cur_hi >> 24);
                column2(SA, SB, c0, c1);
                fetch_S(SA, nxt_lo & 0xFF, nxt_hi & 0xFF);
                column2(SC, SD, c2, c3);
                const u32 nb = pk_max(pk_max(best2, pk_max(c0, c1)), pk_max(c2, c3));
                if (__any(nb != best2)) {
                    // a new maximum somewhere in the block: resolve the column, in visiting order (strict '>',
                    // the earlier column wins)
                    const u32 cs[4] = {c0, c1, c2, c3};
                    if (nmax <= 65000) {
                        u32 b = best2, ps = pos2;
#pragma clang loop unroll(full)
                        for (int k = 0; k < 4; ++k) {
                            const u32 m = pk_maxq(b, cs[k]);
                            const u32 f = pk_minu(m ^ b, ONE2);              // 1 in the halves where column k is a new maximum
                            const u32 jk = (u32)(j0 + k) * 0x00010001u;
                            ps = pk_madu(f, pk_subu(jk, ps), ps);            // their column becomes j0 + k
                            b = m;
                        }
                        best2 = b; pos2 = ps; packed_ahead = true;
                    } else {                                                  // columns beyond u16: the plain way
#pragma clang loop unroll(full)
                        for (int k = 0; k < 4; ++k) {
                            const int cl = lo16(cs[k]) - R * PC_EPS, ch = hi16(cs[k]) - R * PC_EPS;
                            if (cl > bs_lo) { bs_lo = cl; bi_lo = a.m_lo; bj_lo = j0 + k; }
                            if (ch > bs_hi) { bs_hi = ch; bi_hi = a.m_hi; bj_hi = j0 + k; }
                        }
                        best2 = nb;
                    }
                }
            } else {
#if PC_CHECK_RANGE
#define PC_NOTE note_range();
#else
#define PC_NOTE
#endif
                if (packed_ahead) unpack_best();
                // (a tile whose streams all end in column nmax computes no column beyond it: T[] then IS the last column)
                const int last = uniform_fin ? nmax : 0x7FFFFFFF;
                fetch_S(SB, (cur_lo >> 8) & 0xFF, (cur_hi >> 8) & 0xFF);
                column(SlowT{}, j0, SA); PC_NOTE
                if (j0 + 1 <= last) {
                    fetch_S(SA, (cur_lo >> 16) & 0xFF, (cur_hi >> 16) & 0xFF);
                    column(SlowT{}, j0 + 1, SB); PC_NOTE
                    if (j0 + 2 <= last) {
                        fetch_S(SB, cur_lo >> 24, cur_hi >> 24);
                        column(SlowT{}, j0 + 2, SA); PC_NOTE
                        if (j0 + 3 <= last) {
                            fetch_S(SA, nxt_lo & 0xFF, nxt_hi & 0xFF);
                            column(SlowT{}, j0 + 3, SB); PC_NOTE
                        }
                    }
                }
                best2 = packbest(bs_lo, bs_hi);
                pos2 = ((u32)bj_lo & 0xFFFFu) | ((u32)bj_hi << 16);
            }
            jj += 4;
            cur_lo = nxt_lo; cur_hi = nxt_hi;
            nxt_lo = pick_dw(q_lo, kq);
            if (!one_stream) nxt_hi = pick_dw(q_hi, kq);
        }
        if (packed_ahead) unpack_best();
        // ---- the reads' last columns, all lanes at once: rolled re-run of the column from the parked
        // state, cells visited top to bottom with strict '>' (dp_scout.h:165-179).  The last column is the
        // last one the reference visits, so doing it after the loop keeps the visiting order.
        if (uniform_fin) {
            // T[r] = T~(r+1, n) of the tile's common last column n, top = T~(0, n): the candidates (rho, n), rho = 1..R, top to
            // bottom, strict '>' -- what the rolled re-run below derives from the parked state, read off the registers
#pragma clang loop unroll(full)
            for (int r = 0; r < R; ++r) {
                const u32 c = pk_sub(T[r], top);                     // M(rho, n) + rho*eps
#if PC_CHECK_RANGE
                PC_NOTE_V(c)
#endif
                const int il = r - pad_lo + 1, ih = r - pad_hi + 1;
                const int cl = lo16(c) - (r + 1) * PC_EPS, ch = hi16(c) - (r + 1) * PC_EPS;
                if (have_lo && il >= 1 && cl > bs_lo) { bs_lo = cl; bi_lo = il; bj_lo = n_lo; }
                if (have_hi && ih >= 1 && ch > bs_hi) { bs_hi = ch; bi_hi = ih; bj_hi = n_hi; }
            }
        } else {
            const bool ev_lo = have_lo && tail_lo && n_lo > 0, ev_hi = have_hi && tail_hi && n_hi > 0;
            if (__any(ev_lo || ev_hi)) {
                const u32 topn = pk_add(ftop, EPS2);
                u32 diag = ftop, Tup = topn, Vprev = NEG2;
                const u32 *srow = (const u32 *)s_tab;
                const int bl = ev_lo ? w_lo[n_lo - 1] : 0;
                const int bh = one_stream ? bl : (ev_hi ? w_hi[n_hi - 1] : 0);
                const u32 rowbase = 4 * ((u32)lut_lo[bl] + (u32)lut_hi[bh]);
#pragma unroll 1
                for (int r = 0; r < R; ++r) {
                    const uint2 ol = ev_lo ? fin[r * 64 + lane] : make_uint2(0u, 0u);
#if PC_DUAL
                    const uint2 old = ol;
#else
                    const uint2 oh = ev_hi ? fin_hi_buf[r * 64 + lane] : make_uint2(0u, 0u);
                    const uint2 old = make_uint2((ol.x & 0xFFFFu) | (oh.x & 0xFFFF0000u), (ol.y & 0xFFFFu) | (oh.y & 0xFFFF0000u));
#endif
                    const u32 s = srow[rowbase + COMBO[r]];
                    const u32 d = pk_add(diag, s);
                    const u32 Hs = pk_max(old.y, old.x);
                    const u32 Vs = pk_max(Vprev, Tup);
                    const u32 Tn = pk_add(pk_max(pk_max(d, Hs), Vs), OE2);
                    diag = old.x; Tup = Tn; Vprev = Vs;
                    const u32 c = pk_sub(Tn, topn);                  // M(rho, n) + rho*eps, rho = r+1
#if PC_CHECK_RANGE
                    if (ev_lo || ev_hi) { PC_NOTE_V(d) PC_NOTE_V(Vs) PC_NOTE_V(Tn) PC_NOTE_V(c) PC_NOTE_V(pk_max(pk_max(d, Hs), Vs)) }
#endif
                    const int il = r - pad_lo + 1, ih = r - pad_hi + 1;
                    const int cl = lo16(c) - (r + 1) * PC_EPS, ch = hi16(c) - (r + 1) * PC_EPS;
                    if (ev_lo && il >= 1 && cl > bs_lo) { bs_lo = cl; bi_lo = il; bj_lo = n_lo; }
                    if (ev_hi && ih >= 1 && ch > bs_hi) { bs_hi = ch; bi_hi = ih; bj_hi = n_hi; }
                }
            }
        }
#if PC_CHECK_RANGE
        {
            const int hi = lo16(vmax) > hi16(vmax) ? lo16(vmax) : hi16(vmax);
            const int lo = lo16(vmin) < hi16(vmin) ? lo16(vmin) : hi16(vmin);
            if (have_lo || have_hi) {
                atomicMax((int *)a.err + 4, hi); atomicMax((int *)a.err + 5, -lo);
                // the on-device assertion: outside what the lane type holds exactly -> the launch is reported as failed
                if (hi > (PC_F16 ? 2040 : 32000) || -lo > (PC_F16 ? 2040 : 32000)) atomicAdd(a.err, 1u);
            }
        }
#endif
        if (a.rec_out) {
            // (nchunks == 1: the end cell and its score are the whole answer -- what plan_kernel would make of the line below)
            if (have_lo) { int4 *o = (int4 *)(a.rec_out + p_lo * 8); o[0] = int4{-2, bj_lo + c0_lo, bi_lo, 0}; o[1] = int4{bs_lo, 0, 0, 0}; }
            if (have_hi) { int4 *o = (int4 *)(a.rec_out + p_hi * 8); o[0] = int4{-2, bj_hi + c0_hi, bi_hi, 0}; o[1] = int4{bs_hi, 0, 0, 0}; }
        } else {
            // (a chunk that starts beyond its window's end is never read by the planner -- it merges chunks 0 .. ceil(len / L) - 1 --
            // and is not written: a 4 Mb read among 20 kb reads makes two thousand chunks per window, ten of them real)
            const bool live_lo = have_lo && (chunk == 0 || n_lo > 0), live_hi = have_hi && (chunk == 0 || n_hi > 0);
            if (live_lo) { int4 o = {bs_lo, bi_lo, bj_lo + c0_lo, 0}; *(int4 *)(a.out + (p_lo * nchunks + chunk) * 4) = o; }
            if (live_hi) { int4 o = {bs_hi, bi_hi, bj_hi + c0_hi, 0}; *(int4 *)(a.out + (p_hi * nchunks + chunk) * 4) = o; }
        }
    }
}
)PCJIT";

}  // namespace pcj
