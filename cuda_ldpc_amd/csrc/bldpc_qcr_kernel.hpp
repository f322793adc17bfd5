// bldpc_qcr_kernel.hpp -- fused QC min-sum kernel for LONG blocks (BLDPC_KERNEL_QC_LDS, third tier):
// codes whose compressed check states (12 M bytes) do not fit LDS next to the a-posteriori values either,
// i.e. BASELINE config 4, J15_L30_Z1280 (N = 38 400: 4 N = 150 KB of the CU's 160 KB).
//
// Where the state lives.  LDS holds ONE array, the a-posteriori value S of every variable (4 N bytes).  The
// check states -- (min1, min2) and a word with the index of the minimum edge and the sign bit of every output,
// from which R_p = +-min1 / +-min2 is rebuilt exactly (bldpc_qcc_kernel.hpp, bldpc_实习/LDPC_Decoder.cu:298-312)
// -- live in REGISTERS of the thread that owns the check: thread tid owns rows (j, tid + z*TPB) for every block
// row j, J*Z/TPB states of 3 VGPRs.  Nothing but the channel values (re-read once per iteration, L2-resident) and
// the packed hard bits touches HBM: 4 N bytes per frame and iteration instead of the 16 E + 8 N of the
// reference schedule (2.66 MB -> 154 KB at config 4).
//
// Schedule of one flooding iteration, bit-identical to the reference's two kernels:
//   phase 1 (check nodes, LDPC_Decoder.cu:279-314): every check reads S of its neighbours through the rotation
//           (r + s) mod Z, rebuilds its own previous outputs, forms Q = S - R (the value the reference's VN
//           kernel stored, :206-209), runs min-sum and keeps the new state in its registers.  S is read-only.
//   phase 2 (variable nodes, :188-210): S_new = (((0 + R_0) + R_1) + ...) + y in ascending block-row order.  No
//           thread can gather the states held in other threads' registers, so the sum is built check-side: block
//           rows are visited in ascending order with a workgroup barrier after each; inside one block row every
//           variable is met by at most one edge, so `S[v] = S[v] + R` is race-free and happens in exactly the
//           reference's order.  A column's first edge stores 0 + R (overwriting the old S, which phase 1 no
//           longer needs); one aligned pass adds the channel values at the end.
// The per-edge tables (column, shift, first/last) are wave-uniform: scalar loads.
#pragma once
#include "bldpc_math.hpp"
#include "bldpc_qcc_kernel.hpp"

namespace cldpc {

template <int J_, int Z_, int TPB_, int WCS_, int MINW_> struct QcrGeom {
    static constexpr int J = J_, Z = Z_, TPB = TPB_, WCS = WCS_, ZR = Z / TPB, MINW = MINW_; // every row has >= MINW edges
    static_assert(Z % TPB == 0 && TPB % 64 == 0 && TPB <= 1024, "threads must tile the circulant in whole waves");
    static_assert(WCS <= 27, "sign bits and the 5-bit index share one word");
};

// CN slot word: column | shift << 8 | first edge of its column << 19 | last edge of its column << 20 (unused)
__host__ __device__ inline unsigned qcr_cn_meta(int col, int shift, int first, int last)
{
    return (unsigned)col | ((unsigned)shift << 8) | ((unsigned)first << 19) | ((unsigned)last << 20);
}

} // namespace cldpc
