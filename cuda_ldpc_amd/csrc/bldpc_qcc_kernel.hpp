// bldpc_qcc_kernel.hpp -- fused QC min-sum kernel with COMPRESSED check-node state (BLDPC_KERNEL_QC_LDS,
// second tier): for codes whose per-edge messages (4E bytes per frame) do not fit LDS but whose check
// states do -- the reference's Z = 160 family (and any other Z = 256 / 512 shape the register-state kernel of
// bldpc_qcr_kernel.hpp has no instantiation for).
//
// A check row's outputs take only two magnitudes: R_p = +-min1, except +-min2 on the (first) edge that holds
// the minimum (bldpc_实习/LDPC_Decoder.cu:298-312).  So instead of one message per edge the CN phase publishes
// 12 bytes per check: (min1, min2) and one word holding the index of that edge and the sign bit of every
// output.  The VN phase rebuilds R for each of its edges from the check's state (same bits as the stored
// message would have had), sums in the reference's order and publishes the a-posteriori value S; the CN
// phase rebuilds its own previous outputs the same way, forms Q = S - R (LDPC_Decoder.cu:206-209) and
// runs min-sum.  LDS per frame: 12 (M + Z) + 4 (N + Z) bytes instead of 4 E + 4 N (Z extra states stay zero: the padding
// entries of a column's edge list point at them and add R = +0; one extra column stays +inf for the padding slots of a row).
//
// One frame per workgroup; lanes run along the circulant dimension; a thread group of U = Z rounded up to
// whole waves makes the group index wave-uniform, so the per-edge tables (block column / row, position,
// shift) are scalar loads and rows / columns are plain loops: no per-thread address registers, any J and L.
#pragma once
#include "bldpc_math.hpp"

namespace cldpc {

struct QcArgs; // bldpc_qc_kernel.hpp

template <int Z_, int U_, int G_, int CPT_, int WCS_> struct QccGeom {
    static constexpr int Z = Z_, U = U_, G = G_, CPT = CPT_, WCS = WCS_, TPB = G * U;
    static_assert(U % 64 == 0 && U >= Z && Z % 32 == 0 && TPB <= 1024, "geometry out of range");
    static_assert(WCS <= 27, "sign bits and the 5-bit index share one word");
};

// meta words.  CN slot: col | shift << 8 (col == L: the +inf padding column).  VN edge: row | pos << 6 | shift << 11.
__host__ __device__ inline unsigned qcc_cn_meta(int col, int shift) { return (unsigned)col | ((unsigned)shift << 8); }
__host__ __device__ inline unsigned qcc_vn_meta(int row, int pos, int shift) { return (unsigned)row | ((unsigned)pos << 6) | ((unsigned)shift << 11); }

// R of edge `pos` from a check's state: magnitude min2 on the minimum edge else min1, sign bit `pos` of w2.
__device__ __forceinline__ float qcc_recon(float m1, float m2, unsigned w2, int pos)
{
    const float mag = ((int)(w2 >> 27) == pos) ? m2 : m1;
    return u2f(f2u(mag) | ((w2 >> pos) << 31));
}

} // namespace cldpc
