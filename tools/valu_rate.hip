// Experiment: sustained issue rate of common VALU instructions on gfx950, wave-instructions per SIMD-cycle at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP> __global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8]; uint32_t u[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
                if (OP == 5) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 4) & 7]));
                if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&a[i & 6]) : "v"(*(double *)&a[(i + 2) & 6]));
                if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(*(unsigned long long *)&u[i & 6]), "v"(*(unsigned long long *)&u[(i + 2) & 6]) : "vcc");
                if (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 10) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double *)&a[i & 6]) : "v"(*(double *)&a[(i + 2) & 6]));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256 threads = 4 waves = one per SIMD
    float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 32;
    printf("%-22s %d waves/SIMD: %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_f32", w); run<1>("v_xor_b32", w); run<3>("v_med3_f32", w); run<4>("v_cndmask_b32", w); run<5>("s_nop1+v_mov_b32_dpp", w);
        run<6>("v_pk_add_f32", w); run<7>("v_lshl_add_u32", w); run<8>("v_cmp_gt_u64", w); run<9>("v_alignbit_b32", w); run<10>("v_fma_f64", w);
    }
    return 0;
}
