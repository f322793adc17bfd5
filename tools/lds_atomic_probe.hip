// lds_atomic_probe.hip -- what k_qcr's variable-node phase could use instead of read / add / write per edge:
//   (1) numerics of ds_add_f32 against v_add_f32 (signed zeros, denormals, huge, random bit patterns): are the bits the same?
//   (2) issue rates per CU of ds_add_f32, ds_read_b32, ds_read_addtid_b32, ds_write_b32 with lane-linear addresses
//   (3) whether M0 of the ADDTID instructions reaches beyond 64 KB (S of J15_L30_Z1280 is 150 KB)
// Build: hipcc -O2 --offload-arch=gfx950 -o lds_atomic_probe tools/lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((address_space(3))) char lds_byte;

__global__ void k_numerics(const uint32_t *a, const uint32_t *b, uint32_t *out_ds, uint32_t *out_v, int n)
{
    extern __shared__ float lds[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lds[threadIdx.x] = __uint_as_float(a[i]);
    __syncthreads();
    const float bv = __uint_as_float(b[i]);
    asm volatile("ds_add_f32 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"((lds_byte *)(lds + threadIdx.x)), "v"(bv) : "memory");
    __syncthreads();
    out_ds[i] = __float_as_uint(lds[threadIdx.x]);
    out_v[i] = __float_as_uint(__uint_as_float(a[i]) + bv);
}

enum { R_ADD, R_RD32, R_RD_ADDTID, R_WR32, R_RD32_ROT, R_ADD_ROT };

template <int OP> __global__ __launch_bounds__(256) void k_rate(float *out, unsigned long long *stamps, int iters, int rot)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(wave * 4096)) : "memory");
    // lane-linear address inside this wave's 4 KB, optionally rotated by `rot` lanes (what a cyclic shift looks like)
    const int base = wave * 4096 + (((lane + rot) & 63) << 2);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = 1.0f + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == R_ADD || OP == R_ADD_ROT) {
                asm volatile("ds_add_f32 %0, %1 offset:0" : : "v"(base), "v"(v[i]) : "memory");
                asm volatile("ds_add_f32 %0, %1 offset:512" : : "v"(base), "v"(v[i]) : "memory");
            }
            if (OP == R_RD32 || OP == R_RD32_ROT) {
                asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(v[i]) : "v"(base) : "memory");
                asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(v[i]) : "v"(base) : "memory");
            }
            if (OP == R_RD_ADDTID) {
                asm volatile("ds_read_addtid_b32 %0 offset:0" : "=v"(v[i]) : : "memory");
                asm volatile("ds_read_addtid_b32 %0 offset:512" : "=v"(v[i]) : : "memory");
            }
            if (OP == R_WR32) {
                asm volatile("ds_write_b32 %0, %1 offset:0" : : "v"(base), "v"(v[i]) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:512" : : "v"(base), "v"(v[i]) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

// (3): fill 150 KB with the word index, read through ADDTID with M0 = base for bases up to 149 KB
__global__ __launch_bounds__(64) void k_m0(uint32_t *out, int nb)
{
    extern __shared__ float lds[];
    uint32_t *w = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < 150 * 256; i += 64) w[i] = (uint32_t)i;
    __syncthreads();
    for (int b = 0; b < nb; b++) {
        const int base = b * 1024 * 10; // bytes
        uint32_t r;
        asm volatile("s_mov_b32 m0, %1\n s_nop 1\n ds_read_addtid_b32 %0 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "s"(base) : "memory");
        out[b * 64 + threadIdx.x] = r;
    }
}

template <int OP> void rate(const char *name, int rot)
{
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w, iters = 4000;
        float *out; unsigned long long *st;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
        (void)hipMalloc(&st, (size_t)blocks * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, 50, rot);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, iters, rot);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 2);
        (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0); // shader cycles per ns (real-time counter: 100 MHz)
        // a CU hosts 4 w waves; CU-cycles per wave-instruction = launch time / (iters * 16 instructions * 4 w waves)
        printf("%-34s rot %2d, %d waves/SIMD: %6.2f CU-cycles per wave-instruction (%.3f ms, %.2f GHz)\n", name, rot, w,
               ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * 4 * w), ms, ghz);
        (void)hipFree(out); (void)hipFree(st);
    }
}

int main()
{
    // (1)
    std::vector<uint32_t> a, b;
    const uint32_t sp[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x807fffffu, 0x00800000u, 0x80800000u, 0x00800001u,
                           0x3f800000u, 0xbf800000u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x33800000u, 0xb3800000u, 0x3f800001u,
                           0x00400000u, 0x80400000u, 0x01000000u, 0x81000000u, 0x00c00000u, 0x80c00000u};
    for (uint32_t x : sp)
        for (uint32_t y : sp) { a.push_back(x); b.push_back(y); }
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    for (int i = 0; i < 4000000; i++) {
        uint32_t x = rnd(), y = rnd();
        if ((i & 3) == 1) y = (x ^ 0x80000000u) + (rnd() % 64) - 32;                       // near-cancellation
        if ((i & 3) == 2) { x &= 0x80ffffffu; y &= 0x80ffffffu; }                           // tiny and denormal
        if ((i & 3) == 3) { x = (x & 0x807fffffu) | 0x3f000000u; y = (y & 0x807fffffu) | ((0x7e + rnd() % 3) << 23); } // around 1
        if (((x >> 23) & 255) == 255 || ((y >> 23) & 255) == 255) continue;              // no NaN / inf inputs
        a.push_back(x); b.push_back(y);
    }
    const int n = (int)a.size();
    uint32_t *da, *db, *d1, *d2;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&d1, n * 4); (void)hipMalloc(&d2, n * 4);
    (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_numerics, dim3((n + 255) / 256), dim3(256), 1024, 0, da, db, d1, d2, n);
    std::vector<uint32_t> o1(n), o2(n);
    (void)hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_host = 0;
    for (int i = 0; i < n; i++) {
        float fa, fb; memcpy(&fa, &a[i], 4); memcpy(&fb, &b[i], 4);
        volatile float hs = fa + fb; float hv = hs; uint32_t hu; memcpy(&hu, &hv, 4);
        if (o2[i] != hu && hu == hu) bad_host++;
        if (o1[i] != o2[i]) {
            if (bad < 12) printf("  ds_add_f32(%08x, %08x) = %08x, v_add_f32 = %08x\n", a[i], b[i], o1[i], o2[i]);
            bad++;
        }
    }
    printf("(1) ds_add_f32 vs v_add_f32: %d pairs, %ld differ; v_add_f32 vs host IEEE add: %ld differ\n", n, bad, bad_host);
    // (2)
    rate<R_ADD>("ds_add_f32", 0); rate<R_ADD_ROT>("ds_add_f32", 17);
    rate<R_RD32>("ds_read_b32", 0); rate<R_RD32_ROT>("ds_read_b32", 17);
    rate<R_RD_ADDTID>("ds_read_addtid_b32", 0);
    rate<R_WR32>("ds_write_b32", 0);
    // (3)
    uint32_t *dm;
    (void)hipMalloc(&dm, 15 * 64 * 4);
    (void)hipFuncSetAttribute((const void *)k_m0, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k_m0, dim3(1), dim3(64), 150 * 1024, 0, dm, 15);
    std::vector<uint32_t> hm(15 * 64);
    (void)hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost);
    for (int b2 = 0; b2 < 15; b2++) {
        const uint32_t want = (uint32_t)(b2 * 10240 / 4);
        printf("(3) M0 = %6d B: lane 0 reads word %6u, lane 63 word %6u (%s)\n", b2 * 10240, hm[b2 * 64], hm[b2 * 64 + 63],
               (hm[b2 * 64] == want && hm[b2 * 64 + 63] == want + 63) ? "ok" : "NOT the addressed words");
    }
    return 0;
}
