// pc_jit_source.h -- source of the adapter-SPECIALISED score-only scan kernel, compiled at run
// time with hiprtc for one (adapter_lo, adapter_hi) pair (pc_jit.cpp).
//
// Why: in the generic kernels 4 of the 11 packed ops per cell pair build the diagonal term
// (y = h - v_i, z = min_u16(y, D), dq = T_diag + (a - o), d = dq - z) because the adapter base of a
// row is run-time data.  With the adapter known at compile time, the per-column substitution
// terms S[k] = sub(h, v_lo) - o | (sub(h, v_hi) - o) << 16 of the (few) distinct letter pairs k
// that occur in the adapter pair are fetched once per column from an LDS table indexed by the
// read byte, and every row's diagonal is ONE op: d = T_diag + S[COMBO[r]] with a static register
// index.  8 packed ops per cell pair instead of 11.  Padding rows (shorter adapter of a pair)
// use a letter whose S is -o, i.e. substitution score 0, which keeps M = 0 like row 0.
//
// The kernel is score-only (pass 1 of the whole-read scan) and bit-identical in outputs to
// scan_kernel<R, *, false>; tests run both (PC_DISABLE_JIT=1 selects the generic one).
#pragma once

namespace pcj {

// defines prepended by pc_jit.cpp: PC_R, PC_K (multiple of 4), PC_COMBO_INIT (R comma-separated ints)
static const char *kSpecSource = R"PCJIT(
typedef unsigned int u32;
typedef long long i64;
#if PC_F16
// Packed FP16 variant: every DP value is a small integer (|v| <= 1002 by the host-side gate), which
// fp16 represents and adds exactly, and gfx950 has a 3-input packed max (v_pk_maximum3_f16) where
// the integer ISA only has 2-input ones: 7 instead of 8 packed ops per cell pair.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 SV(u32 x) { return __builtin_bit_cast(h16x2, x); }
__device__ __forceinline__ u32 WV(h16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return WV(SV(a) + SV(b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return WV(__builtin_elementwise_max(SV(a), SV(b))); }
__device__ __forceinline__ u32 pack2(int v) { const h16x2 h = {(_Float16)v, (_Float16)v}; return WV(h); }
__device__ __forceinline__ int lo16(u32 x) { return (int)(float)SV(x).x; }
__device__ __forceinline__ int hi16(u32 x) { return (int)(float)SV(x).y; }
#define PC_NEG (-1000)
#define PC_ADD "v_pk_add_f16"
#define PC_MAX "v_pk_max_f16"
#else
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 SV(u32 x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u32 WV(s16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return WV(SV(a) + SV(b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return WV(__builtin_elementwise_max(SV(a), SV(b))); }
__device__ __forceinline__ u32 pack2(int v) { return ((u32)v & 0xFFFFu) | ((u32)v << 16); }
__device__ __forceinline__ int lo16(u32 x) { return (int)(short)(x & 0xFFFFu); }
__device__ __forceinline__ int hi16(u32 x) { return (int)(short)(x >> 16); }
#define PC_NEG (-16384)
#define PC_ADD "v_pk_add_u16"
#define PC_MAX "v_pk_max_i16"
#endif

struct Tile { i64 win_lo, win_hi, out_lo, out_hi; int count_lo, count_hi, adapter_lo, adapter_hi, rows, pad_; };
struct SpecArgs {
    const unsigned char *arena; const i64 *win_off; const int *win_len;
    const Tile *tiles; int ntiles;
    int *out;                 // [npairs * chunks][4]: score, I, J, 0
    uint2 *fin_scratch;       // [grid][R*64]
    const u32 *s_table;       // [256][PC_K]
    int m_lo, m_hi, gap_open, gap_extend;
    int chunks, chunk_len, span;
    u32 *err;
};

__device__ static const unsigned char COMBO[PC_R] = { PC_COMBO_INIT };

typedef u32 u32_unaligned __attribute__((aligned(1)));

extern "C" __global__ __launch_bounds__(64) void pc_spec_score(SpecArgs a)
{
    constexpr int R = PC_R, K = PC_K;
    __shared__ uint4 s_tab[256 * K / 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256 * K / 4; i += 64) s_tab[i] = ((const uint4 *)a.s_table)[i];
    __syncthreads();
    const u32 E2 = pack2(a.gap_extend), O2 = pack2(a.gap_open), NEG2 = pack2(PC_NEG);
    const int pad_lo = R - a.m_lo, pad_hi = R - a.m_hi;
    uint2 *fin = a.fin_scratch + (i64)blockIdx.x * R * 64;
    const int nchunks = a.chunks > 1 ? a.chunks : 1;

    for (int vt = blockIdx.x; vt < a.ntiles * nchunks; vt += gridDim.x) {
        const int t = vt / nchunks, chunk = vt - t * nchunks;
        const Tile tile = a.tiles[t];
        const bool one_stream = tile.win_lo == tile.win_hi;
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
        // state "row-0 start + vertical gap" (see pc_bounds.h)
        u32 T[R], U[R];
#pragma clang loop unroll(full)
        for (int r = 0; r < R; ++r) {
            const int vl = (c0_lo > 0 && r >= pad_lo) ? 2 * a.gap_open + (r - pad_lo) * a.gap_extend : a.gap_open;
            const int vh = (c0_hi > 0 && r >= pad_hi) ? 2 * a.gap_open + (r - pad_hi) * a.gap_extend : a.gap_open;
            T[r] = (pack2(vl) & 0xFFFFu) | (pack2(vh) & 0xFFFF0000u);
            U[r] = NEG2;
        }
        int bs_lo = 0, bi_lo = a.m_lo, bj_lo = 0, bs_hi = 0, bi_hi = a.m_hi, bj_hi = 0;
        if (chunk > 0) { bs_lo = -32768; bj_lo = -1; bs_hi = -32768; bj_hi = -1; }   // (m,0) belongs to chunk 0
        int nmax = n_lo > n_hi ? n_lo : n_hi;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(nmax, s); nmax = o > nmax ? o : nmax; }

        // Read bytes are fetched a dword (4 columns) at a time, one dword ahead; the substitution
        // terms S[] of column j+1 are fetched from the LDS table while column j computes (two
        // register sets, the column loop statically unrolled by 4), so neither the global load nor
        // the LDS read latency sits on the column's critical path.
        auto load_dw = [&](const unsigned char *w, int n, int col) -> u32 {   // dword holding 0-based columns col..col+3
            const int k = (col < n) ? col : (n > 0 ? ((n - 1) & ~3) : 0);      // finished streams re-read their last dword
            return *(const u32_unaligned *)(w + k);
        };
        auto fetch_S = [&](u32 (&S)[K], u32 bl, u32 bh) {
            const uint4 *row = s_tab + bl * (K / 4);
#pragma clang loop unroll(full)
            for (int q = 0; q < K / 4; ++q) { const uint4 v = row[q]; S[4*q] = v.x; S[4*q+1] = v.y; S[4*q+2] = v.z; S[4*q+3] = v.w; }
            if (!one_stream) {
                const uint4 *rowh = s_tab + bh * (K / 4);
#pragma clang loop unroll(full)
                for (int q = 0; q < K / 4; ++q) {
                    const uint4 v = rowh[q];
                    S[4*q]   = (S[4*q]   & 0xFFFFu) | (v.x & 0xFFFF0000u);
                    S[4*q+1] = (S[4*q+1] & 0xFFFFu) | (v.y & 0xFFFF0000u);
                    S[4*q+2] = (S[4*q+2] & 0xFFFFu) | (v.z & 0xFFFF0000u);
                    S[4*q+3] = (S[4*q+3] & 0xFFFFu) | (v.w & 0xFFFF0000u);
                }
            }
        };
        auto column = [&](const int j, const u32 (&S)[K]) {
            const bool fin_lo = (j == n_lo) && tail_lo, fin_hi = (j == n_hi) && tail_hi;
            const bool any_fin = __any(fin_lo || fin_hi);
            if (any_fin && (fin_lo || fin_hi)) {
#pragma clang loop unroll(full)
                for (int r = 0; r < R; ++r) fin[lane * R + r] = make_uint2(T[r], U[r]);
            }
            // ---- the column: 8 packed ops per row, hand-scheduled.  gfx950 needs a wait state
            // between dependent packed ops; each asm block issues row r's vertical chain
            // (Vx, Vs, M, T') interleaved with row r+2's chain-independent half (d, Hx, Hs,
            // max(d,Hs)) so that no two dependent ops are adjacent -- not within a block and not
            // across consecutive blocks -- and hipcc has nothing to pad.
            {
                constexpr int KP = 2;
                u32 dh[R];     // int16: max(d, H) of the rows in flight;  fp16: d of the rows in flight
                // H of the new column is written straight into U[q] (its old value is dead once
                // Hx is formed), T' straight into T[r] (last read two blocks earlier): no copies
                auto ind_only = [&](int q, u32 diag) {
                    u32 dq, hx;
#if PC_F16
                    asm volatile(PC_ADD " %[hx], %[uq], %[e2]\n\t"
                                 PC_ADD " %[dq], %[diag], %[s]\n\t"
                                 PC_MAX " %[uq], %[hx], %[tq]"
                                 : [hx] "=&v"(hx), [dq] "=&v"(dh[q]), [uq] "+v"(U[q])
                                 : [e2] "s"(E2), [diag] "v"(diag), [s] "v"(S[COMBO[q]]), [tq] "v"(T[q]));
                    (void)dq;
#else
                    asm volatile(PC_ADD " %[hx], %[uq], %[e2]\n\t"
                                 PC_ADD " %[dq], %[diag], %[s]\n\t"
                                 PC_MAX " %[uq], %[hx], %[tq]\n\t"
                                 "s_nop 0\n\t"
                                 PC_MAX " %[dhq], %[dq], %[uq]"
                                 : [hx] "=&v"(hx), [dq] "=&v"(dq), [uq] "+v"(U[q]), [dhq] "=&v"(dh[q])
                                 : [e2] "s"(E2), [diag] "v"(diag), [s] "v"(S[COMBO[q]]), [tq] "v"(T[q]));
#endif
                };
#pragma clang loop unroll(full)
                for (int q = 0; q < KP && q < R; ++q) ind_only(q, q == 0 ? O2 : T[q - 1]);
                u32 Tup = O2, Vprev = NEG2;
                // two rows per asm statement (hipcc pads a wait state between dependent asm
                // statements it cannot see into, so fewer, larger statements)
#if PC_F16
                // 7 ops: Vx, Hx', Vs, d', M = max3(d, H, Vs), H', T'   (primed = row r+2)
#define PC_ROW_FULL(VP, TU, DIAG, SS, UQ, TQ, DHR, UR, TN, DHQ, VS)                  \
    PC_ADD " %[vx], " VP ", %[e2]\n\t"                                                \
    PC_ADD " %[hx], " UQ ", %[e2]\n\t"                                                \
    PC_MAX " " VS ", %[vx], " TU "\n\t"                                               \
    PC_ADD " " DHQ ", " DIAG ", " SS "\n\t"                                           \
    "v_pk_maximum3_f16 %[mn], " DHR ", " UR ", " VS "\n\t"                            \
    PC_MAX " " UQ ", %[hx], " TQ "\n\t"                                               \
    PC_ADD " " TN ", %[mn], %[o2]\n\t"
#define PC_ROW_TAIL                                                                    \
    PC_ADD " %[vx], %[vprev], %[e2]\n\t" "s_nop 0\n\t"                                 \
    PC_MAX " %[vs0], %[vx], %[tup]\n\t" "s_nop 0\n\t"                                  \
    "v_pk_maximum3_f16 %[mn], %[dhr0], %[ur0], %[vs0]\n\t" "s_nop 0\n\t"               \
    PC_ADD " %[tn0], %[mn], %[o2]"
#else
                // 8 ops: Vx, d', Vs, Hx', M = max(max(d,H), Vs), H', T', max(d',H')
#define PC_ROW_FULL(VP, TU, DIAG, SS, UQ, TQ, DHR, UR, TN, DHQ, VS)                  \
    PC_ADD " %[vx], " VP ", %[e2]\n\t"                                                \
    PC_ADD " %[dq], " DIAG ", " SS "\n\t"                                             \
    PC_MAX " " VS ", %[vx], " TU "\n\t"                                               \
    PC_ADD " %[hx], " UQ ", %[e2]\n\t"                                                \
    PC_MAX " %[mn], " DHR ", " VS "\n\t"                                              \
    PC_MAX " " UQ ", %[hx], " TQ "\n\t"                                               \
    PC_ADD " " TN ", %[mn], %[o2]\n\t"                                                \
    PC_MAX " " DHQ ", %[dq], " UQ "\n\t"
#define PC_ROW_TAIL                                                                    \
    PC_ADD " %[vx], %[vprev], %[e2]\n\t" "s_nop 0\n\t"                                 \
    PC_MAX " %[vs0], %[vx], %[tup]\n\t" "s_nop 0\n\t"                                  \
    PC_MAX " %[mn], %[dhr0], %[vs0]\n\t" "s_nop 0\n\t"                                 \
    PC_ADD " %[tn0], %[mn], %[o2]"
#endif
#pragma clang loop unroll(full)
                for (int r = 0; r < R; r += 2) {
                    u32 vx, mn, dq, hx, vs0, vs1;
                    (void)dq;
                    if (r + 1 + KP < R) {
                        const int q = r + KP;
                        asm volatile(
                            PC_ROW_FULL("%[vprev]", "%[tup]", "%[d0]", "%[s0]", "%[u0]", "%[t0]", "%[dhr0]", "%[ur0]", "%[tn0]", "%[dhq0]", "%[vs0]")
                            PC_ROW_FULL("%[vs0]", "%[tn0]", "%[t0]", "%[s1]", "%[u1]", "%[t1]", "%[dhr1]", "%[ur1]", "%[tn1]", "%[dhq1]", "%[vs1]")
                            : [vx] "=&v"(vx), [dq] "=&v"(dq), [hx] "=&v"(hx), [mn] "=&v"(mn), [vs0] "=&v"(vs0), [vs1] "=&v"(vs1),
                              [u0] "+v"(U[q]), [u1] "+v"(U[q + 1]), [tn0] "=&v"(T[r]), [tn1] "=&v"(T[r + 1]),
                              [dhq0] "=&v"(dh[q]), [dhq1] "=&v"(dh[q + 1])
                            : [vprev] "v"(Vprev), [tup] "v"(Tup), [e2] "s"(E2), [o2] "s"(O2), [d0] "v"(T[q - 1]),
                              [t0] "v"(T[q]), [t1] "v"(T[q + 1]), [s0] "v"(S[COMBO[q]]), [s1] "v"(S[COMBO[q + 1]]),
                              [dhr0] "v"(dh[r]), [dhr1] "v"(dh[r + 1]), [ur0] "v"(U[r]), [ur1] "v"(U[r + 1]));
                        Tup = T[r + 1]; Vprev = vs1;
                    } else {
                        // tail rows (and an odd last row): one row at a time
#pragma clang loop unroll(full)
                        for (int rr = r; rr < r + 2 && rr < R; ++rr) {
                            if (rr + KP < R) {
                                const int q = rr + KP;
                                asm volatile(
                                    PC_ROW_FULL("%[vprev]", "%[tup]", "%[d0]", "%[s0]", "%[u0]", "%[t0]", "%[dhr0]", "%[ur0]", "%[tn0]", "%[dhq0]", "%[vs0]")
                                    : [vx] "=&v"(vx), [dq] "=&v"(dq), [hx] "=&v"(hx), [mn] "=&v"(mn), [vs0] "=&v"(vs0),
                                      [u0] "+v"(U[q]), [tn0] "=&v"(T[rr]), [dhq0] "=&v"(dh[q])
                                    : [vprev] "v"(Vprev), [tup] "v"(Tup), [e2] "s"(E2), [o2] "s"(O2), [d0] "v"(T[q - 1]),
                                      [t0] "v"(T[q]), [s0] "v"(S[COMBO[q]]), [dhr0] "v"(dh[rr]), [ur0] "v"(U[rr]));
                            } else {
                                asm volatile(PC_ROW_TAIL
                                             : [vx] "=&v"(vx), [vs0] "=&v"(vs0), [mn] "=&v"(mn), [tn0] "=&v"(T[rr])
                                             : [vprev] "v"(Vprev), [e2] "s"(E2), [tup] "v"(Tup), [dhr0] "v"(dh[rr]),
                                               [ur0] "v"(U[rr]), [o2] "s"(O2));
                            }
                            Tup = T[rr]; Vprev = vs0;
                        }
                    }
                }
#undef PC_ROW_FULL
#undef PC_ROW_TAIL
            }
            if (any_fin) {
                // last column of a pair: rolled re-run from the saved previous column, tracked
                // cells visited top to bottom with strict '>' (dp_scout.h:165-179)
                u32 diag = O2, Tup = O2, Vprev = NEG2;
                const u32 *srow = (const u32 *)s_tab;
                // the bytes of THIS column were consumed above: recover them from the streams
                const int bl = (j - 1 < n_lo && n_lo > 0) ? w_lo[j - 1] : 0;
                const int bh = one_stream ? bl : ((j - 1 < n_hi && n_hi > 0) ? w_hi[j - 1] : 0);
#pragma unroll 1
                for (int r = 0; r < R; ++r) {
                    const uint2 old = (fin_lo || fin_hi) ? fin[lane * R + r] : make_uint2(0u, 0u);
                    const u32 s = (srow[bl * K + COMBO[r]] & 0xFFFFu) | (srow[bh * K + COMBO[r]] & 0xFFFF0000u);
                    const u32 d = pk_add(diag, s);
                    const u32 Hs = pk_max(pk_add(old.y, E2), old.x);
                    const u32 Vs = pk_max(pk_add(Vprev, E2), Tup);
                    const u32 Tn = pk_add(pk_max(pk_max(d, Hs), Vs), O2);
                    diag = old.x; Tup = Tn; Vprev = Vs;
                    const int il = r - pad_lo + 1, ih = r - pad_hi + 1;
                    const int cl = lo16(Tn) - a.gap_open, ch = hi16(Tn) - a.gap_open;
                    if (fin_lo && il >= 1 && cl > bs_lo) { bs_lo = cl; bi_lo = il; bj_lo = j; }
                    if (fin_hi && ih >= 1 && ch > bs_hi) { bs_hi = ch; bi_hi = ih; bj_hi = j; }
                }
            }
            {
                const int cl = lo16(T[R - 1]) - a.gap_open, ch = hi16(T[R - 1]) - a.gap_open;
                const bool tr_lo = j > tf_lo && (tail_lo ? j < n_lo : j <= n_lo);
                const bool tr_hi = j > tf_hi && (tail_hi ? j < n_hi : j <= n_hi);
                if (tr_lo && cl > bs_lo) { bs_lo = cl; bi_lo = a.m_lo; bj_lo = j; }
                if (tr_hi && ch > bs_hi) { bs_hi = ch; bi_hi = a.m_hi; bj_hi = j; }
            }
        };
        u32 cur_lo = load_dw(w_lo, n_lo, 0), cur_hi = one_stream ? 0u : load_dw(w_hi, n_hi, 0);
        u32 nxt_lo = load_dw(w_lo, n_lo, 4), nxt_hi = one_stream ? 0u : load_dw(w_hi, n_hi, 4);
        u32 SA[K], SB[K];
        fetch_S(SA, cur_lo & 0xFF, cur_hi & 0xFF);
        for (int j0 = 1; j0 <= nmax; j0 += 4) {
            fetch_S(SB, (cur_lo >> 8) & 0xFF, (cur_hi >> 8) & 0xFF);
            column(j0, SA);
            fetch_S(SA, (cur_lo >> 16) & 0xFF, (cur_hi >> 16) & 0xFF);
            column(j0 + 1, SB);
            fetch_S(SB, cur_lo >> 24, cur_hi >> 24);
            column(j0 + 2, SA);
            fetch_S(SA, nxt_lo & 0xFF, nxt_hi & 0xFF);
            column(j0 + 3, SB);
            cur_lo = nxt_lo; cur_hi = nxt_hi;
            nxt_lo = load_dw(w_lo, n_lo, j0 + 7);
            if (!one_stream) nxt_hi = load_dw(w_hi, n_hi, j0 + 7);
        }
        if (have_lo) { int4 o = {bs_lo, bi_lo, bj_lo + c0_lo, 0}; *(int4 *)(a.out + (p_lo * nchunks + chunk) * 4) = o; }
        if (have_hi) { int4 o = {bs_hi, bi_hi, bj_hi + c0_hi, 0}; *(int4 *)(a.out + (p_hi * nchunks + chunk) * 4) = o; }
    }
}
)PCJIT";

}  // namespace pcj
