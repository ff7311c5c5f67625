// Micro-benchmark: issue rate of the VALU ops the scan kernels are made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OPS_PER_ITER 64
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s0, unsigned s1, unsigned long long *ticks)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i;
    const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime: shader-clock ticks
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
                if (KIND == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 14) {   // packed fp32: 64-bit register pairs (two FMAs per lane per instruction)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(unsigned long long *)&a[(2 * i) & 6])
                                 : "v"(*(unsigned long long *)&a[(2 * i + 2) & 6]), "v"(*(unsigned long long *)&a[(2 * i + 4) & 6]));
                }
                if (KIND == 15) asm volatile("v_pk_add_f16 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1] clamp" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(s0));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    unsigned *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    int iters = 20000;
    unsigned long long *dt; hipMalloc(&dt, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 100, 3, 1, dt);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 3, 1, dt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> ht((size_t)blocks * 4);
    hipMemcpy(ht.data(), dt, ht.size() * 8, hipMemcpyDeviceToHost);
    double mean_ticks = 0;
    for (auto t : ht) mean_ticks += (double)t;
    mean_ticks /= (double)ht.size();
    double wave_instr = (double)blocks * 4 * iters * OPS_PER_ITER;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    // per-wave shader-clock ticks (s_memtime) per instruction, divided by the waves sharing the SIMD = SIMD
    // cycles per wave64 instruction at the clock the chip actually ran; ticks / wall = that clock
    printf("%-34s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s/SIMD  cycles/instr @2.4GHz nominal = %.2f | by s_memtime = %.2f (clock %.2f GHz)\n",
           name, waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s,
           mean_ticks / ((double)iters * OPS_PER_ITER) / waves_per_simd, mean_ticks / (ms * 1e-3) / 1e9);
    hipFree(d); hipFree(dt);
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
        run<13>("v_fma_f32", w);
        run<14>("v_pk_fma_f32 (2 FMA/lane)", w);
        run<15>("v_pk_add_f16 neg clamp", w);
        run<16>("v_perm_b32", w);
        run<7>("v_pk_max_i16 dependent chain", w);
    }
    return 0;
}
