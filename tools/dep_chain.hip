// Experiment: cycles per DEPENDENT v_add_f32 / v_sub_f32 in one wave alone on its SIMD (the float chain of the EMS walk),
// alone and with k independent instructions between two links of the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FILL> __global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, int iters, float a, float b)
{
    float s = a, f0 = b, f1 = b * 2, f2 = b * 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(b));
            if (FILL >= 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f0) : "v"(b));
            if (FILL >= 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f1) : "v"(b));
            if (FILL >= 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f2) : "v"(b));
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s) : "v"(b));
            if (FILL >= 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f0) : "v"(b));
            if (FILL >= 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f1) : "v"(b));
            if (FILL >= 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f2) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = s + f0 + f1 + f2;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int FILL> void run()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int blocks : {1, 1024}) { // 1024 blocks of one wave: one wave per SIMD on every CU
        hipLaunchKernelGGL(k<FILL>, dim3(blocks), dim3(64), 0, 0, out, cyc, 2000, 1.0f, 0.5f);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("dependent add/sub chain, %d independent v_xor between links, %4d waves: %.2f cycles per chain link\n", FILL, blocks, (double)h / (2000.0 * 32));
    }
}
int main() { run<0>(); run<1>(); run<2>(); run<3>(); return 0; }
