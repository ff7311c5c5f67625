// Micro-benchmark: issue rate of the VALU ops the scan kernels are made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OPS_PER_ITER 64
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s0, unsigned s1)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < OPS_PER_ITER / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "s"(s0));
                if (KIND == 1) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "s"(s1));
                if (KIND == 3) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(a[(i + 1) & 7]));
                if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "s"(s0));
                if (KIND == 5) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 7) asm volatile("v_pk_max_i16 %0, %0, %0" : "+v"(a[0]));   // fully dependent chain
                if (KIND == 8) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 9) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 10) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 11) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "s"(s0));
                if (KIND == 12) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            }
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    unsigned *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 100, 3, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 3, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * OPS_PER_ITER;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s/SIMD-> cycles/instr @2.4GHz = %.2f\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_add_u16 (v,s)", w);
        run<1>("v_pk_max_i16 (v,v)", w);
        run<2>("v_pk_min_u16 (v,s)", w);
        run<6>("v_pk_sub_i16 (v,v)", w);
        run<3>("v_pk_mad_u16 (v,s,v)", w);
        run<4>("v_add_u32 (v,s)", w);
        run<5>("v_max_i32 (v,v)", w);
        run<8>("v_max3_i32", w);
        run<9>("v_pk_fma_f16", w);
        run<10>("v_pk_maximum3_f16", w);
        run<11>("v_pk_add_f16 (v,s)", w);
        run<12>("v_pk_max_f16 (v,v)", w);
        run<7>("v_pk_max_i16 dependent chain", w);
    }
    return 0;
}
