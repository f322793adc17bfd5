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
    __device__ __forceinline__ uint32_t key() const { return (f2u(m1) ^ f2u(m2)) ^ (sgn & 0x80000000u); }
};

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
