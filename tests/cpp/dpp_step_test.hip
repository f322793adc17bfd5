// GPU test (test_nbldpc_gpu.py::test_bitonic_steps): the DPP form of a bitonic compare-exchange step against the ds_swizzle form, for every (K, J) of the network.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NB_ABLATE 0
#include "../../cuda_ldpc_amd/csrc/nbldpc_kernel.hpp"
using namespace cldpc;

template <int K, int J> __device__ void ref_step(uint32_t &hi, uint32_t &lo)
{
    constexpr unsigned long long KM = nb_keepmax_mask(K, J);
    const int lane = threadIdx.x & 63;
    const uint32_t phi = (uint32_t)__shfl_xor((int)hi, J, 64), plo = (uint32_t)__shfl_xor((int)lo, J, 64);
    const unsigned long long mine = ((unsigned long long)hi << 32) | lo, part = ((unsigned long long)phi << 32) | plo;
    const bool keepmax = (KM >> lane) & 1;
    const bool take = (part > mine) == keepmax;
    if (take) { hi = phi; lo = plo; }
}
template <int K, int J> __device__ void one(uint32_t hi, uint32_t lo, int *bad, int id)
{
    uint32_t h1 = hi, l1 = lo, h2 = hi, l2 = lo;
    // two independent copies through the product's step so that the interleaving matches the kernel's use
    uint32_t h3 = hi ^ 0x55u, l3 = lo;
    nb_bitonic_step<K, J>(h1, l1);
    nb_bitonic_step<K, J>(h3, l3);
    ref_step<K, J>(h2, l2);
    if (h1 != h2 || l1 != l2) atomicAdd(&bad[id], 1);
}
__global__ void k(const uint32_t *hi, const uint32_t *lo, int *bad)
{
    const uint32_t h = hi[threadIdx.x], l = lo[threadIdx.x];
    int id = 0;
#define T(K, J) one<K, J>(h, l, bad, id++);
    T(2, 1) T(4, 2) T(4, 1) T(8, 4) T(8, 2) T(8, 1) T(16, 8) T(16, 4) T(16, 2) T(16, 1) T(32, 16) T(32, 8) T(32, 4) T(32, 2) T(32, 1)
    T(64, 32) T(64, 16) T(64, 8) T(64, 4) T(64, 2) T(64, 1)
}
// the whole network, four sorts in flight as in the kernel; keys with heavy ties in the high word
__global__ void ksort(const uint32_t *hi, uint32_t *out_hi, uint32_t *out_lo)
{
    uint32_t h[4], l[4];
    for (int i = 0; i < 4; i++) { h[i] = hi[i * 64 + threadIdx.x]; l[i] = 63u - threadIdx.x; }
    nb_bitonic_sort<64, 4>(h, l);
    for (int i = 0; i < 4; i++) { out_hi[i * 64 + threadIdx.x] = h[i]; out_lo[i * 64 + threadIdx.x] = l[i]; }
}
static int check_sort()
{
    uint32_t h[256], oh[256], ol[256];
    srand(7);
    for (int i = 0; i < 256; i++) h[i] = (i < 64) ? (uint32_t)(rand() % 3) : (i < 128) ? 5u : (i < 192) ? (uint32_t)rand() : (uint32_t)(rand() % 7) * 0x20000000u;
    uint32_t *dh, *doh, *dol;
    hipMalloc(&dh, 1024); hipMalloc(&doh, 1024); hipMalloc(&dol, 1024);
    hipMemcpy(dh, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ksort, dim3(1), dim3(64), 0, 0, dh, doh, dol);
    hipMemcpy(oh, doh, 1024, hipMemcpyDeviceToHost); hipMemcpy(ol, dol, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int a = 0; a < 4; a++) {
        // expected: stable descending order = descending (hi, 63 - index)
        unsigned long long k[64];
        for (int i = 0; i < 64; i++) k[i] = ((unsigned long long)h[a * 64 + i] << 32) | (unsigned)(63 - i);
        for (int i = 0; i < 64; i++) for (int j = i + 1; j < 64; j++) if (k[j] > k[i]) { unsigned long long t = k[i]; k[i] = k[j]; k[j] = t; }
        int b = 0;
        for (int i = 0; i < 64; i++) b += (oh[a * 64 + i] != (uint32_t)(k[i] >> 32)) || (ol[a * 64 + i] != (uint32_t)k[i]);
        printf("full sort %d: %d positions differ\n", a, b);
        bad += b;
    }
    return bad;
}
int main()
{
    uint32_t h[64], l[64];
    srand(1);
    for (int i = 0; i < 64; i++) { h[i] = (rand() % 7) * 0x10000000u + (rand() & 3); l[i] = 63 - i; }
    uint32_t *dh, *dl; int *db;
    hipMalloc(&dh, 256); hipMalloc(&dl, 256); hipMalloc(&db, 21 * 4);
    hipMemcpy(dh, h, 256, hipMemcpyHostToDevice); hipMemcpy(dl, l, 256, hipMemcpyHostToDevice); hipMemset(db, 0, 84);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dh, dl, db);
    int bad[21];
    hipMemcpy(bad, db, 84, hipMemcpyDeviceToHost);
    const char *names[21] = {"2,1","4,2","4,1","8,4","8,2","8,1","16,8","16,4","16,2","16,1","32,16","32,8","32,4","32,2","32,1","64,32","64,16","64,8","64,4","64,2","64,1"};
    int total = check_sort();
    for (int i = 0; i < 21; i++) { printf("(K,J)=(%s): %d lanes differ\n", names[i], bad[i]); total += bad[i]; }
    printf("total %d\n", total);
    return total != 0;
}
