// Micro-benchmark: issue rate of the specialised kernel's ROW pattern (5 packed-fp16 ops per row, a
// 3-op serial chain V -> M -> T' through consecutive rows) on gfx950, to separate what the
// instruction mix itself can reach from the kernel's other overheads.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_row.hip -o /tmp/ubench_row && /tmp/ubench_row
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 32
// KIND 0: kernel pattern (H', V, d', max3, s_nop, T)      1: same without the s_nop (hazard ignored: timing only)
// KIND 2: two independent chains interleaved, no nops       3: 6-op pattern, chain M -> T only (dh precomputed)
template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned *out, int iters, unsigned oe, unsigned s)
{
    unsigned T[ROWS], U[ROWS], T2[ROWS], U2[ROWS];
    for (int i = 0; i < ROWS; ++i) { T[i] = threadIdx.x + i; U[i] = threadIdx.x * 3 + i; T2[i] = T[i] ^ 5; U2[i] = U[i] ^ 9; }
    unsigned vp = 0, tu = 0, vp2 = 0, tu2 = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned dh[ROWS], dh2[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { dh[r] = T[r] + 1; dh2[r] = T2[r] + 1; }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int q = (r + 2) % ROWS;
            unsigned mn, vs, mn2, vs2;
            if (KIND == 0)
                asm volatile("v_pk_max_f16 %[uq], %[uq], %[tq]\n\tv_pk_max_f16 %[vs], %[vp], %[tu]\n\tv_pk_add_f16 %[dq], %[tq], %[s]\n\t"
                             "v_pk_maximum3_f16 %[mn], %[dr], %[ur], %[vs]\n\ts_nop 0\n\tv_pk_add_f16 %[tn], %[mn], %[oe]"
                             : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [mn] "=&v"(mn), [tn] "=&v"(T[r])
                             : [tq] "v"(T[q]), [vp] "v"(vp), [tu] "v"(tu), [s] "v"(s), [dr] "v"(dh[r]), [ur] "v"(U[r]), [oe] "s"(oe));
            if (KIND == 1)
                asm volatile("v_pk_max_f16 %[uq], %[uq], %[tq]\n\tv_pk_max_f16 %[vs], %[vp], %[tu]\n\tv_pk_add_f16 %[dq], %[tq], %[s]\n\t"
                             "v_pk_maximum3_f16 %[mn], %[dr], %[ur], %[vs]\n\tv_pk_add_f16 %[tn], %[mn], %[oe]"
                             : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [mn] "=&v"(mn), [tn] "=&v"(T[r])
                             : [tq] "v"(T[q]), [vp] "v"(vp), [tu] "v"(tu), [s] "v"(s), [dr] "v"(dh[r]), [ur] "v"(U[r]), [oe] "s"(oe));
            if (KIND == 2) {
                asm volatile("v_pk_max_f16 %[vs], %[vp], %[tu]\n\tv_pk_max_f16 %[vs2], %[vp2], %[tu2]\n\t"
                             "v_pk_max_f16 %[uq], %[uq], %[tq]\n\t"
                             "v_pk_maximum3_f16 %[mn], %[dr], %[ur], %[vs]\n\tv_pk_maximum3_f16 %[mn2], %[dr2], %[ur2], %[vs2]\n\t"
                             "v_pk_add_f16 %[dq], %[tq], %[s]\n\t"
                             "v_pk_add_f16 %[tn], %[mn], %[oe]\n\tv_pk_add_f16 %[tn2], %[mn2], %[oe]\n\t"
                             "v_pk_max_f16 %[uq2], %[uq2], %[tq2]\n\tv_pk_add_f16 %[dq2], %[tq2], %[s]"
                             : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [mn] "=&v"(mn), [tn] "=&v"(T[r]),
                               [uq2] "+v"(U2[q]), [vs2] "=&v"(vs2), [dq2] "=&v"(dh2[q]), [mn2] "=&v"(mn2), [tn2] "=&v"(T2[r])
                             : [tq] "v"(T[q]), [vp] "v"(vp), [tu] "v"(tu), [s] "v"(s), [dr] "v"(dh[r]), [ur] "v"(U[r]), [oe] "s"(oe),
                               [tq2] "v"(T2[q]), [vp2] "v"(vp2), [tu2] "v"(tu2), [dr2] "v"(dh2[r]), [ur2] "v"(U2[r]));
                vp2 = vs2; tu2 = T2[r];
            }
            if (KIND == 3)   // M = max3(dh, Vprev, Tup) ; T = M + oe ; off-chain: H', d', dh' = max(d',H'), V = max(Vprev, Tup)
                asm volatile("v_pk_max_f16 %[uq], %[uq], %[tq]\n\t"
                             "v_pk_maximum3_f16 %[mn], %[dr], %[vp], %[tu]\n\t"
                             "v_pk_add_f16 %[dq], %[tq], %[s]\n\t"
                             "v_pk_max_f16 %[vs], %[vp], %[tu]\n\t"
                             "v_pk_add_f16 %[tn], %[mn], %[oe]\n\t"
                             "v_pk_max_f16 %[dq], %[dq], %[uq]"
                             : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [mn] "=&v"(mn), [tn] "=&v"(T[r])
                             : [tq] "v"(T[q]), [vp] "v"(vp), [tu] "v"(tu), [s] "v"(s), [dr] "v"(dh[r]), [oe] "s"(oe));
            vp = vs; tu = T[r];
        }
    }
    unsigned x = vp ^ tu ^ vp2 ^ tu2;
    for (int i = 0; i < ROWS; ++i) x ^= T[i] ^ U[i] ^ T2[i] ^ U2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int KIND> void run(const char *name, int waves_per_simd, int cells_per_row)
{
    int blocks = 1024 * waves_per_simd;
    unsigned *d; hipMalloc(&d, (size_t)blocks * 64 * 4);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 64>>>(d, 10, 0x3c003c00u, 0x40004000u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(d, iters, 0x3c003c00u, 0x40004000u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double rows = (double)blocks * iters * ROWS * cells_per_row;
    printf("%-44s waves/SIMD=%d  %.3f ms  cycles per wave-row per SIMD @2.4GHz = %.2f\n", name, waves_per_simd, ms,
           2.4e9 * (ms * 1e-3) * 1024.0 / rows);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 3, 4}) {
        run<0>("5-op row, s_nop", w, 1);
        run<1>("5-op row, no nop (hazard ignored)", w, 1);
        run<2>("2 chains interleaved, 10 ops per 2 rows", w, 2);
        run<3>("6-op row, 2-op chain, no nop", w, 1);
    }
    return 0;
}
