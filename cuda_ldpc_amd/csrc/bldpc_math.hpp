// bldpc_math.hpp -- device-side min-sum arithmetic shared by the table and QC kernels.
//
// Restates, bit-exactly, the per-node arithmetic of the reference kernels
// (bldpc_实习/LDPC_Decoder.cu): the variable-node sum of :188-210 and the
// check-node update of :279-314 + sortQ :374-398, without their structure.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cldpc {

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __uint_as_float(x); }

// Running (min1, min2, sign parity) of a check row; one step per incoming Q.
//   min2' = median(min1, min2, |q|), min1' = min(min1, |q|)  == the two smallest
//   magnitudes with multiplicity, which is what sortQ's two bubble passes leave
//   in Q[w-1], Q[w-2] (LDPC_Decoder.cu:374-398).
struct CnAcc {
    float m1, m2;
    uint32_t sgn;
    __device__ __forceinline__ void init()
    {
        m1 = __builtin_inff();
        m2 = __builtin_inff();
        sgn = 0u;
    }
    __device__ __forceinline__ void add(float q)
    {
        float a = __builtin_fabsf(q);
        sgn ^= f2u(q);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
        m1 = __builtin_fminf(m1, a);
    }
    // XOR key applied by out(): (min1 ^ min2) selects the other magnitude, the
    // top bit carries the row's sign product (Sign[25], LDPC_Decoder.cu:293-296).
    __device__ __forceinline__ uint32_t key() const { return __builtin_amdgcn_bitop3_b32(f2u(m1), f2u(m2), sgn & 0x80000000u, 0x96); } // a ^ b ^ c in one full-rate instruction
};

// The two smallest magnitudes (with multiplicity) and the XOR of the sign bits of N values at once: what N calls of
// CnAcc::add leave in (m1, m2, sgn), with 3-input instructions -- min3 / med3 of a triple, then
// (m1, m2) <- (min(m1, lo), min3(max(m1, lo), m2, mid)): 5 instructions per 3 values instead of 6 (14 instead of 20 for the
// ten edges of half a J4_L24_Z96 row; min / max / med3 all issue every 4.2 cycles per SIMD, profiles/r02_micro_rates.txt).
// Selection only, no arithmetic: the same bits whatever the order.
template <int N, int STRIDE> __device__ __forceinline__ void cn_two_smallest(const float *q, float &m1, float &m2, uint32_t &sgn)
{
    static_assert(N >= 2, "a check row half has at least two edges");
    auto A = [&](int i) { return __builtin_fabsf(q[i * STRIDE]); };
    int i = 0;
    if (N >= 3) {
        m1 = __builtin_fminf(__builtin_fminf(A(0), A(1)), A(2));
        m2 = __builtin_amdgcn_fmed3f(A(0), A(1), A(2));
        i = 3;
    } else {
        m1 = __builtin_fminf(A(0), A(1));
        m2 = __builtin_fmaxf(A(0), A(1));
        i = 2;
    }
#pragma unroll
    for (; i + 3 <= N; i += 3) {
        const float lo = __builtin_fminf(__builtin_fminf(A(i), A(i + 1)), A(i + 2));
        const float mid = __builtin_amdgcn_fmed3f(A(i), A(i + 1), A(i + 2));
        m2 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(m1, lo), m2), mid);
        m1 = __builtin_fminf(m1, lo);
    }
#pragma unroll
    for (; i < N; i++) {
        m2 = __builtin_amdgcn_fmed3f(m1, m2, A(i));
        m1 = __builtin_fminf(m1, A(i));
    }
    // the sign product two values at a time: v_bitop3_b32 (any 3-input bit function; 0x96 = a ^ b ^ c) issues at the full rate on
    // gfx950, unlike v_or3 / v_and_or / v_bfi (tools/micro_rates.hip -> profiles/r03_micro_rates_bitop3.txt: 2.4 against 4.2 SIMD-cycles):
    // 5 instructions instead of 10 for ten edges -- J4_L24_Z96 12.67 -> 13.03 M codewords/s, J32_L64_Z64 6.34 -> 6.53 M
#ifdef QC_XOR2 /* experiment: one v_xor_b32 per value */
    sgn = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) sgn ^= f2u(q[k * STRIDE]);
#else
    sgn = (N % 2) ? f2u(q[0]) : (f2u(q[0]) ^ f2u(q[STRIDE]));
#pragma unroll
    for (int k = 2 - (N % 2); k + 1 < N; k += 2) sgn = __builtin_amdgcn_bitop3_b32(sgn, f2u(q[k * STRIDE]), f2u(q[(k + 1) * STRIDE]), 0x96);
#endif
}

// R_i = Sign[25]*Sign[i] * (i == Index_minQ ? SubMinQ : MinQ)   (LDPC_Decoder.cu:298-312)
//   clamp(q, -m2, +m2) = sign(q) * min(|q|, m2), which is sign(q)*m1 exactly for the
//   edge(s) holding the minimum and sign(q)*m2 for every other edge; XOR with
//   (m1^m2) swaps the two magnitudes.  When the minimum is duplicated m1 == m2 and the
//   reference's "first index" rule makes no difference.  q is never -0.0f or NaN here
//   (q = S - R with S accumulated from +0.0f), so the sign bit equals (q < 0).
__device__ __forceinline__ float cn_out(float q, float m2, uint32_t key)
{
    return u2f(key ^ f2u(__builtin_amdgcn_fmed3f(q, -m2, m2)));
}

} // namespace cldpc
