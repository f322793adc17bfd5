// bldpc_qc_kernel.hpp -- fused QC-LDPC flooding min-sum kernel (BLDPC_KERNEL_QC_LDS).
//
// One workgroup decodes NF*FP frames for ALL iterations without touching HBM in
// between: channel values live in registers, every message lives in LDS.  Same
// arithmetic, in the same order, as the reference's two kernels per iteration
// (bldpc_实习/LDPC_Decoder.cu:172-372), none of their structure.
//
// Mapping.  A block (j,l) with shift s connects check row r to variable column
// (r+s) mod Z.  Lanes run along the circulant dimension: thread (g, p, t) owns
//   * check rows   (j, t) for j = g, g+G, ...   of frame group p   (CN phase)
//   * variables    (l, t) for l = g, g+G, ...   of frame group p   (VN phase)
// and every lane carries NF (1 or 2) frames side by side, so LDS traffic is
// ds_read_b64/ds_write_b64 and the VN sums are packed adds.  A group of U lanes
// (FP*Z rounded up to whole waves; surplus lanes idle) makes g wave-uniform, so
// every table look-up is a scalar load.  The geometry (J, L, Z, FP, G, U) is a
// template parameter: block and column offsets inside LDS become immediate
// offsets of the ds_ instructions; only the shifts and the block positions are
// run-time data (read once, in the prologue).
//
// Message exchange ("APP exchange", bit-identical to the reference's in-place
// R/Q memory, SURVEY Appendix A note):
//   VN phase: S = ((0+R_0)+R_1+...)+y in ascending block-row order (A.2); the
//             thread reads R through the rotation (c-s) mod Z and publishes S
//             aligned: one LDS write per VARIABLE instead of one per EDGE.
//   CN phase: the check thread reads S of its neighbours through the rotation
//             (r+s) mod Z and its own previous outputs R_p (aligned), forms
//             Q_p = S - R_p -- the value the reference's VN kernel would have
//             stored (LDPC_Decoder.cu:206-209) -- runs min-sum, publishes R aligned.
// LDS is read-cheap / write-expensive on gfx950 (ds_read_b64 256 B/clk,
// ds_write_b64 ~85 B/clk per CU): this trades E writes for E extra reads.
// Rotations are LDS addresses computed once and kept as 16-bit slot indices
// packed two per VGPR.  The iteration loop is branch-free: rows lighter than WC
// read a slot that always holds +inf (neutral for min1/min2/sign) and write into
// the row's own padding blocks; columns lighter than WV read a slot that always
// holds +0.0f (adding +0.0f to a sum that is never -0.0f is exact).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bldpc.h"
#include "bldpc_math.hpp"
#include "bldpc_qc_assign.hpp"
#include "bldpc_qcc_kernel.hpp"
#include "common.hpp"

// Wave priorities of the half-row kernel's stages (s_setprio): the stages that only move messages through LDS (variable-node
// phase, output stores) run above the arithmetic of the check-node phase.  Two workgroups share a CU and the hardware serves
// the older one first: without this the younger workgroup crawls through its LDS-bound stages behind the older one's VALU work
// (first 11 iterations at 8 000 cycles each against 3 000 later, profiles/r02_qc2_phase_probe.txt); with it 11.6 -> 12.3 M codewords/s.
#ifndef QC_PRIO_VN
#define QC_PRIO_VN 1
#define QC_PRIO_CN 0
#define QC_PRIO_WR 1
#endif
// with local edges the output stage holds 7 stores and 20 of its 40 vector instructions feed no store: it stays at the arithmetic's
// priority (12.59 -> 12.64 M codewords/s; 1 / 0 / 1 as above: 12.59; VN 2, CN 1, WR 0: 11.8; profiles/r03_qc2_variants.txt)
#ifndef QC_PRIO_WR_LOC
#define QC_PRIO_WR_LOC 0
#endif

namespace cldpc {


struct QcArgs {
    const float *y;             // [ceil(F/NF)][N][NF]  channel values regrouped per workgroup (k_regroup_y)
    const float *y_raw;         // or, k_qc / k_qc2 with NF = 2 and F even: the reference's own layout [N][F], read in place (8 bytes
                                // per variable: a lane's two frames are neighbours there); nullptr = use y
    int *D;                     // [N+1][F]  (the kernel itself writes only the flag row N)
    unsigned *bits;             // [F][N/32] packed hard bits
    float *app;                 // [N][F] or nullptr
    unsigned long long *hist;   // [F] flag history or nullptr
    const QcCnEdge *cn_edges;   // [nnz]
    const unsigned short *rowptr; // [J+1]
    const QcVnEdge *vn_edges;   // [L][WV]
    const unsigned char *wv;    // [L]
    int F, nWG, max_iter, length;
    // compressed-state kernel (k_qcc) only:
    const unsigned *cn_meta;    // [J][WCS] padded row slots
    const unsigned *vn_meta;    // [L][WVS] column edges, top -> bottom
    int J, L, WVS;
    int lc;                     // register-state kernel (k_qcr): the block column kept in registers
    // per-frame exit (HIST instantiations only): a frame stops at the first iteration at which ITS flag is set --
    // the reference rule (LDPC_Decoder.cu:134-153) with Num_Frames_OneTime = 1 -- its outputs are those of that
    // iteration, and the workgroup leaves when all of its frames have stopped
    int per_frame;
    int *work = nullptr;        // k_qc2p: one frame-pair counter per XCD (8 ints, zeroed by the host)
    int *iters;                 // [F] iterations executed per frame (per_frame only)
#ifdef QC_STAMPS
    unsigned long long *stamps; // tools/qc_phase_probe.hip only: [nWG][QC_STAMPS] s_memtime stamps of wave 0
    int stagger;                // tools only: cycles of s_sleep for workgroups with an odd TG_ID before the loop
#endif
};

template <int NF> struct Msg;
template <> struct Msg<1> { using T = float; };
typedef float v2f32 __attribute__((ext_vector_type(2)));
template <> struct Msg<2> { using T = v2f32; };

// LDS access of one message (NF floats) by absolute LDS byte address.  The kernel owns all of its LDS
// through the dynamic segment and declares no static __shared__ object, so that segment starts at LDS
// address 0; addressing it with plain integers (address space 3) avoids the `v_add base, offset` the
// compiler otherwise emits in front of every ds_ instruction for the (link-time) segment base.
typedef __attribute__((address_space(3))) char lds_char;
template <int NF> __device__ __forceinline__ void lds_ld(float (&d)[NF], const char *, int byte_off)
{
    typedef __attribute__((address_space(3))) const typename Msg<NF>::T lds_msg;
    typename Msg<NF>::T v = *reinterpret_cast<lds_msg *>(static_cast<unsigned>(byte_off));
    __builtin_memcpy(d, &v, sizeof(v));
}
template <int NF> __device__ __forceinline__ void lds_st(char *, int byte_off, const float (&s)[NF])
{
    typedef __attribute__((address_space(3))) typename Msg<NF>::T lds_msg;
    typename Msg<NF>::T v;
    __builtin_memcpy(&v, s, sizeof(v));
    *reinterpret_cast<lds_msg *>(static_cast<unsigned>(byte_off)) = v;
}

// Compile-time geometry of one kernel variant: G*Z threads decode NF frames.
// LOC: local edges as in QcGeom2 (see there): every column handed to one block row that contains it, L / J columns per row; the
// variable of a local block lives in the check thread, its R and S never touch LDS.  Needs wave-uniform thread groups (Z whole
// waves): the place of the local edge in its column's ascending order is a scalar per column.
template <int NF_, int J_, int L_, int Z_, int WC_, int WV_, int G_, int MINW_, bool LOC_ = false> struct QcGeom {
    static constexpr int NF = NF_, J = J_, L = L_, Z = Z_, WC = WC_, WV = WV_, G = G_, MINW = MINW_;
    static constexpr bool LOC = LOC_;
    static constexpr int NLR = L / J; // LOC: local edges per row = its first NLR slots
    static_assert(!LOC || (Z % 64 == 0 && L % J == 0 && L / J >= 1 && L / J <= WC_ && WV >= 2), "local edges: wave-uniform groups, L / J columns per row");
    static constexpr int RPT = (J + G - 1) / G, CPT = (L + G - 1) / G, TPB = G * Z;
    static constexpr int MSG = NF * 4;              // bytes per message slot
    static constexpr int Sslot = J * WC * Z;        // R blocks (row-padded to WC), then S columns
    static constexpr int zero_slot = Sslot + L * Z; // always +0.0f
    static constexpr int inf_slot = zero_slot + 1;  // always +inf
    static constexpr int flag_byte = (inf_slot + 1) * MSG;
    static constexpr int lds_bytes = flag_byte + NF * 8; // flag words: [2][NF], see flags_publish
    static constexpr int NW = (L * Z) / 32;         // 32-bit words of hard bits per frame
    static_assert(TPB % 64 == 0 && TPB <= 1024, "workgroup must be whole waves");
    static_assert(Z % 32 == 0, "half-waves must not straddle a thread group");
    static_assert(J % G == 0 && L % G == 0, "groups must tile the block rows and columns");
    static_assert(inf_slot < 65536, "slot indices are packed into 16 bits");
};

// Thread tid = g*Z + t owns check rows (j, t), j = g + rr*G and variables (l, t), l = g + cc*G.
// g is NOT wave-uniform (Z need not be a multiple of 64): nothing in the iteration loop depends on g
// except per-lane base addresses, so no lane ever idles; the per-edge tables are gathered per lane,
// once, in the prologue.
// Store of one message (NF floats) at base + OFF with OFF in the instruction's offset field, as inline assembly: the compiler
// merges two such stores into one ds_write2_b64 (13.5 LDS cycles against 2 x 6.2, profiles/r02_micro_rates.txt) when it sees
// them.  Its lgkmcnt bookkeeping does not see these either: the caller drains the counter before a barrier.
template <int NF, int OFF> __device__ __forceinline__ void lds_st_imm(int base, const float (&v)[NF])
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_write has a 16-bit offset");
    if constexpr (NF == 2) {
        v2f32 x = {v[0], v[1]};
        asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base), "v"(x), "n"(OFF) : "memory");
    } else {
        asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(base), "v"(v[0]), "n"(OFF) : "memory");
    }
}
// The same accesses on whole messages (float or v2f32), and the components of one.
template <int NF> __device__ __forceinline__ typename Msg<NF>::T lds_ldm(int byte_off)
{
    typedef __attribute__((address_space(3))) const typename Msg<NF>::T lds_msg;
    return *reinterpret_cast<lds_msg *>(static_cast<unsigned>(byte_off));
}
template <int NF, int OFF> __device__ __forceinline__ void lds_stm_imm(int base, typename Msg<NF>::T v)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_write has a 16-bit offset");
    if constexpr (NF == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base), "v"(v), "n"(OFF) : "memory");
    else asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(base), "v"(v), "n"(OFF) : "memory");
}
template <int NF> __device__ __forceinline__ float msg_get(const typename Msg<NF>::T &m, int v)
{
    if constexpr (NF == 2) return v ? m.y : m.x;
    else return m;
}
template <int NF> __device__ __forceinline__ void msg_set(typename Msg<NF>::T &m, int v, float x)
{
    if constexpr (NF == 2) { if (v) m.y = x; else m.x = x; }
    else m = x;
}
template <int... Is, typename Fn> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// Row kernel with local edges: the places (0 .. WV-1) of a thread group's CPT local blocks in their columns' orders, as one number
// (digit cc = column cc, base WV).  The host sorts a row's local blocks by place and the rows of a group by their digit tuples, so only
// the sorted codes occur; the kernel has one iteration loop per such code (and row weight), chosen once per wave.
template <int WV> constexpr int qc_kdigit(int code, int cc)
{
    for (int i = 0; i < cc; i++) code /= WV;
    return code % WV;
}
template <int WV, int RPT, int NLR> constexpr bool qc_kcode_sorted(int code)
{
    for (int rr = 0; rr < RPT; rr++)
        for (int pp = 0; pp + 1 < NLR; pp++)
            if (qc_kdigit<WV>(code, rr * NLR + pp) > qc_kdigit<WV>(code, rr * NLR + pp + 1)) return false;
    for (int rr = 0; rr + 1 < RPT; rr++) { // rows of the group in lexicographic order of their tuples
        for (int pp = 0; pp < NLR; pp++) {
            const int a = qc_kdigit<WV>(code, rr * NLR + pp), b = qc_kdigit<WV>(code, (rr + 1) * NLR + pp);
            if (a < b) break;
            if (a > b) return false;
        }
    }
    return true;
}
constexpr int qc_ipow(int b, int e) { return e == 0 ? 1 : b * qc_ipow(b, e - 1); }

#define QC1_NAME k_qc
#define QC1_PERSIST 0
#include "bldpc_qc1_body.inc"
#undef QC1_NAME
#undef QC1_PERSIST
#define QC1_NAME k_qcp
#define QC1_PERSIST 1
#include "bldpc_qc1_body.inc"
#undef QC1_NAME
#undef QC1_PERSIST

// ---------------------------------------------------------------------------------------------
// Split-row variant: each check row is shared by the two half-waves of one wave (lanes i and i+32 own the
// two halves of the row's WC edges for the same t), each half-wave also owns its own set of variable
// columns.  Twice the threads per frame at about half the registers each, i.e. twice the waves per SIMD for
// the same LDS footprint -- on gfx950 one wave issues at most one VALU instruction per 4 cycles, a SIMD one
// per 2, so the fused kernel needs >= 4-6 resident waves per SIMD to keep the VALU and the LDS pipe busy
// through barriers.  The halves' (min1, min2, sign) meet through three v_permlane32_swap per frame.
//
// LOC ("local edges", round 3): a block column may be stored rotated -- which lane holds which variable of a column is free as long
// as everybody addresses S the same way -- so for ONE of its blocks the variable (l, (t+s) mod Z) can live in the very thread that
// owns check (j, t).  With every column handed to a half-row that contains it (CPT columns per half-row, a matching found on the
// host) that edge never touches LDS: its R stays in the check thread's register for the variable-node sum, its S in the same
// thread's register for the next check-node phase -- 3 of a thread's 12 R reads, 3 of its 10 S reads and 3 of its 10 R stores
// per iteration at J4_L24_Z96.  The variable-node sum keeps the reference's order (ascending block row): with WV == J a column's
// slot k is block row k (absent rows read the +0.0f slot: adding +0.0f to a sum that is never -0.0f is exact), the local edge of a
// thread of block row j is slot j, and j is wave-uniform: a scalar switch picks one of J bodies.
template <int NF_, int J_, int L_, int Z_, int WC_, int WV_, int GJ_, int MINW_, bool LOC_ = false> struct QcGeom2 {
    static constexpr int NF = NF_, J = J_, L = L_, Z = Z_, WC = WC_, WV = WV_, GJ = GJ_, MINW = MINW_;
    static constexpr bool LOC = LOC_;
    static constexpr int ZB = Z / 32, WCH = WC / 2, NCG = 2 * GJ;
    static constexpr int RPT = J / GJ, CPT = L / NCG, TPB = GJ * ZB * 64;
    static constexpr int MSG = NF * 4;
    static constexpr int Sslot = J * WC * Z;
    static constexpr int zero_slot = Sslot + L * Z;
    static constexpr int inf_slot = zero_slot + 1;
    static constexpr int flag_byte = (inf_slot + 1) * MSG;
    static constexpr int lds_bytes = flag_byte + NF * 8;
    static constexpr int NW = (L * Z) / 32;
    static_assert(Z % 32 == 0 && WC % 2 == 0, "half-waves own 32 circulant positions and half a row each");
    static_assert(J % GJ == 0 && L % NCG == 0, "groups must tile the block rows and columns");
    static_assert(TPB <= 1024 && inf_slot < 65536, "geometry out of range");
    static_assert(!LOC || (GJ == J && WV == J && L / NCG <= WC / 2 && J <= 4), "local edges: one block row per wave, a column's slot k is block row k");
};

// v_permlane32_swap a, b: a <- [a.lo, b.lo], b <- [a.hi, b.hi] (lo = lanes 0-31, hi = lanes 32-63).
__device__ __forceinline__ void swap32(uint32_t &a, uint32_t &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// LDS access of a frame pair by absolute LDS byte address (see lds_ld): one ds_read_b64 / ds_write_b64.
typedef __attribute__((address_space(3))) v2f32 lds_v2;
__device__ __forceinline__ v2f32 lds_ld2(int byte_off) { return *reinterpret_cast<const lds_v2 *>(static_cast<unsigned>(byte_off)); }
__device__ __forceinline__ void lds_st2(int byte_off, v2f32 v) { *reinterpret_cast<lds_v2 *>(static_cast<unsigned>(byte_off)) = v; }
// The same store at base + OFF with OFF in the instruction's offset field, kept out of the compiler's sight so that two of
// them are not merged into one ds_write2_b64 (13.5 LDS cycles against 2 x 6.2).  The caller drains lgkmcnt before a barrier.
template <int OFF> __device__ __forceinline__ void lds_st2_imm(int base, v2f32 v)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_write_b64 has a 16-bit offset");
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base), "v"(v), "n"(OFF) : "memory");
}

// What bounds this kernel (profiles/r02_micro_rates.txt, measured on MI355X): per CU one ds_read_b64 costs 2.1 cycles, one
// ds_write_b64 6.2, one ds_read2_b64 8.1 (twice two single reads); per SIMD add / sub / xor / mov issue every 2.5 cycles,
// min / max / med3 / compares / packed f32 every 4.2, v_permlane32_swap every 8.3.  Hence:
//   * a thread keeps its own previous outputs R_p in REGISTERS (it wrote them): the check-node phase reads only S from LDS
//     (the aligned R reads of the earlier version were merged into ds_read2_b64 by the compiler: 40 LDS cycles per wave and
//     iteration for what 10 registers hold);
//   * the two halves of a row merge ONE-SIDED: three swaps hand lanes 0-31 both halves' (min1, min2, sign) of frame 0 and
//     lanes 32-63 those of frame 1, every lane merges one frame, two more swaps hand both results to all lanes -- 5 swaps
//     and one merge per lane instead of 6 swaps + 6 copies and two merges;
//   * magnitudes merge as unsigned integers (same order as non-negative floats, +inf included; no canonicalising v_max).
#define QC2_NAME k_qc2
#define QC2_PERSIST 0
#include "bldpc_qc2_body.inc"
#undef QC2_NAME
#undef QC2_PERSIST
#define QC2_NAME k_qc2p
#define QC2_PERSIST 1
#include "bldpc_qc2_body.inc"
#undef QC2_NAME
#undef QC2_PERSIST

// Regroup the reference's frame-fastest Channel_Out [N][F] into per-workgroup slabs [F/NF][N][NF] so that the
// decode kernel's loads are contiguous along the circulant dimension (the reference layout would cost every
// lane its own 128-byte line for 4*NF useful bytes).  64 x 64 tiles through LDS; reads and writes coalesced.
template <int NF> __global__ __launch_bounds__(256) void k_regroup_y(const float *y, float *out, int N, int F)
{
    __shared__ float tile[64][65];
    const int f0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int n = n0 + i * 4 + w, f = f0 + lane;
        tile[i * 4 + w][lane] = (n < N && f < F) ? y[(size_t)n * F + f] : 0.0f;
    }
    __syncthreads();
    const int n = n0 + lane;
    for (int q = w; q < 64 / NF; q += 4) { // q-th frame group of this tile
        const int fg = f0 / NF + q;
        if (n < N && fg * NF < F + NF - 1 && fg < (F + NF - 1) / NF) {
            float v[NF];
#pragma unroll
            for (int k = 0; k < NF; k++) v[k] = tile[lane][q * NF + k];
            typename Msg<NF>::T o;
            __builtin_memcpy(&o, v, sizeof(o));
            *reinterpret_cast<typename Msg<NF>::T *>(out + ((size_t)fg * N + n) * NF) = o;
        }
    }
}

// Unpack the hard bits into the reference's D layout (int32 [N][F], frame-fastest): one thread per
// (word w, 4 frames); reads 4 words, writes 32 rows of int4 -- stores coalesce along the frame dimension.
// errs != nullptr: the message-bit errors of Statistic against the all-zero codeword (Simulation.cu:249-257) are counted here, from
// the packed words -- the ones among bits 0 .. length-1 of a frame -- instead of by a second pass over the 4 N F bytes of D.
__global__ __launch_bounds__(256) void k_expand_bits(const unsigned *bits, int *D, int F, int NW, int *errs, int length)
{
    const int f = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int w = blockIdx.y;
    if (f >= F) return;
    unsigned x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (f + i < F) ? bits[(size_t)(f + i) * NW + w] : 0u;
    if (errs) {
        const int lo = w * 32;
        const unsigned mask = lo + 32 <= length ? ~0u : (lo < length ? (1u << (length - lo)) - 1u : 0u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = __builtin_popcount(x[i] & mask);
            if (c) atomicAdd(&errs[f + i], c); // (x[i] = 0 beyond F)
        }
    }
    int *row = D + (size_t)w * 32 * F + f;
    if (f + 3 < F && (F & 3) == 0) {
#pragma unroll
        for (int b = 0; b < 32; b++) {
            int4 o;
            o.x = (x[0] >> b) & 1; o.y = (x[1] >> b) & 1; o.z = (x[2] >> b) & 1; o.w = (x[3] >> b) & 1;
            *reinterpret_cast<int4 *>(row + (size_t)b * F) = o;
        }
    } else {
        for (int b = 0; b < 32; b++)
            for (int i = 0; i < 4 && f + i < F; i++) row[(size_t)b * F + i] = (x[i] >> b) & 1;
    }
}


// ---------------------------------------------------------------------------------------------
// Compressed-state kernel, see bldpc_qcc_kernel.hpp.  LDS: mm float2[M] | w2 uint[M] | S float[(L+1)*Z] | flag.
// PERSIST (per-frame exit only): the grid fills the chip once and a workgroup takes frame after frame of its XCD from the counter
// a.work[xcd], as k_qc2p / k_qcr2<PERSIST> do (frames leave after 1 ... max_iter iterations: one workgroup per frame leaves the CUs far apart)
template <typename GM, bool HIST, bool PERSIST = false> __global__ __launch_bounds__(GM::TPB) void k_qcc(QcArgs a)
{
    constexpr int Z = GM::Z, U = GM::U, G = GM::G, CPT = GM::CPT, WCS = GM::WCS;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int chunk = (a.nWG + 7) >> 3; // XCD-aware workgroup id, see k_qc
    int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (!PERSIST && wg >= a.nWG) return;
    const int J = a.J, L = a.L, M = J * Z, N = L * Z, F = a.F;
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid / U); // wave-uniform
    const int t0 = tid - g * U;
    const bool lane_on = t0 < Z;      // the surplus lanes of the padded group idle
    const int t = lane_on ? t0 : 0;   // ... on valid addresses
    // check states 0 .. M-1, then Z states that stay zero (R = +0: what the padding entries of a column's edge list add)
    const int off_w2 = (M + Z) * 8, off_S = (M + Z) * 12;
    int *lds_flag = reinterpret_cast<int *>(lds + off_S + (L + 1) * Z * 4);
    // per-edge tables, wave-uniform; read through the constant address space: scalar loads whatever else the kernel stores
    typedef __attribute__((address_space(4))) const unsigned qcc_cu32;
    const qcc_cu32 *cmeta = (const qcc_cu32 *)a.cn_meta; // [J][WCS] row slots, then [J] row weights
    const qcc_cu32 *vmeta = (const qcc_cu32 *)a.vn_meta; // [L][WVS] column edges, padded to a multiple of 2 with zero-state entries
    for (;;) { // (one pass unless PERSIST)
    if (PERSIST) {
        const int xcd = (int)(blockIdx.x & 7);
        if (tid == 0) lds_flag[2] = atomicAdd(&a.work[xcd], 1);
        __syncthreads();
        const int ord = __builtin_amdgcn_readfirstlane(lds_flag[2]);
        if (ord >= chunk || xcd * chunk + ord >= a.nWG) break;
        wg = xcd * chunk + ord;
    }
    const int f = wg; // one frame per workgroup

    // ---- prologue -----------------------------------------------------------------------------------------
    for (int j = g; j < J + 1; j += G)
        if (lane_on) {
            const float zero2[2] = {0.0f, 0.0f}; // min1 = min2 = 0, no signs: every R starts as +0 (LDPC_Decoder.cu:82)
            lds_st<2>(lds, (j * Z + t) * 8, zero2);
            const float zw[1] = {0.0f};
            lds_st<1>(lds, off_w2 + (j * Z + t) * 4, zw);
        }
    if (g == 0 && lane_on) {
        const float inf[1] = {__builtin_inff()};
        lds_st<1>(lds, off_S + (L * Z + t) * 4, inf); // padding column: neutral for min1/min2/sign
    }
    float yreg[CPT];
    int wcol[CPT];
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        const int l = g + cc * G;
        wcol[cc] = (l < L) ? a.wv[l] : 0;
        yreg[cc] = (l < L && lane_on) ? a.y[(size_t)f * N + l * Z + t] : 0.0f;
    }
    if (tid == 0) lds_flag[0] = lds_flag[1] = 0; // two flag words, used by odd and even iterations in turn (see k_qc)
    unsigned long long hist = 0;
    int stop = 0; // per-frame exit: the iteration at which the frame's flag came up
    __syncthreads();

    constexpr int VR = 2; // column edges per straight-line round (lists are padded to a multiple of it with zero-state entries)
    auto vn_phase = [&](bool &bad) {
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            const int l = g + cc * G;
            if (l < L) { // wave-uniform
                float S = 0.0f;
                const qcc_cu32 *vm = vmeta + l * a.WVS;
                for (int k0 = 0; k0 < wcol[cc]; k0 += VR) { // VR edges per round, straight-line (the list is padded)
                    float mm[VR][2], w2f[VR][1];
                    int pos[VR];
#pragma unroll
                    for (int i = 0; i < VR; i++) {
                        const unsigned m = vm[k0 + i]; // scalar load
                        const int j = m & 63, sh = m >> 11;
                        pos[i] = (m >> 6) & 31;
                        const unsigned r1 = (unsigned)(t - sh);
                        const int r = (int)min(r1, r1 + (unsigned)Z); // t - sh wraps to a huge value when negative
                        const int sidx = j * Z + r;
                        lds_ld<2>(mm[i], lds, sidx * 8);
                        lds_ld<1>(w2f[i], lds, off_w2 + sidx * 4);
                    }
#pragma unroll
                    for (int i = 0; i < VR; i++) S += qcc_recon(mm[i][0], mm[i][1], f2u(w2f[i][0]), pos[i]); // ascending block row = the reference's edge order
                }
                S += yreg[cc];
                if (lane_on) {
                    const float sv[1] = {S};
                    lds_st<1>(lds, off_S + (l * Z + t) * 4, sv);
                    if (HIST) bad = bad || ((l * Z + t) < a.length && S < 0);
                }
            }
        }
    };
    auto flags_collect = [&](int it) -> int {
        int flag = 0;
        if (tid == 0) {
            flag = lds_flag[it & 1] ? 0 : 1;
            lds_flag[(it + 1) & 1] = 0;
            if (flag && it <= 64) hist |= (1ull << (it - 1));
        }
        return flag;
    };

    for (int it = 1; it < a.max_iter; it++) {
        bool bad = false;
        vn_phase(bad);
        if (HIST && bad) lds_flag[it & 1] = 1;
        __syncthreads();
        if (HIST) {
            (void)flags_collect(it);
            if (a.per_frame && !lds_flag[it & 1]) { // the frame stops with the values of this iteration
                stop = it;
                break;
            }
        }

        // CN phase (LDPC_Decoder.cu:279-314) on the compressed state
        for (int j = g; j < J; j += G) { // wave-uniform
            const int sidx = j * Z + t;
            float pm[2], pw[1];
            lds_ld<2>(pm, lds, sidx * 8);
            lds_ld<1>(pw, lds, off_w2 + sidx * 4);
            const unsigned pw2 = f2u(pw[0]);
            const qcc_cu32 *cm = cmeta + j * WCS;
            const int w = (int)cmeta[J * WCS + j]; // row weight (scalar load)
            float m1 = __builtin_inff(), m2 = __builtin_inff();
            unsigned signs = 0;
            int idx = 0;
            constexpr int CHK = 4; // slots per round; rounds beyond the row's weight are skipped by a scalar branch
#pragma unroll
            for (int p0 = 0; p0 < WCS; p0 += CHK)
                if (p0 < w) {
                    float Sv[CHK];
#pragma unroll
                    for (int i = 0; i < CHK; i++) {
                        const unsigned m = cm[p0 + i]; // scalar load; padding slots point at the +inf column with shift 0
                        const int col = m & 255, sh = m >> 8;
                        const unsigned c1 = (unsigned)(t + sh);
                        const int c = (int)min(c1, c1 - (unsigned)Z); // c1 - Z wraps to a huge value unless c1 >= Z
                        float sv[1];
                        lds_ld<1>(sv, lds, off_S + (col * Z + c) * 4);
                        Sv[i] = sv[0];
                    }
#pragma unroll
                    for (int i = 0; i < CHK; i++) {
                        const int p = p0 + i;
                        const float q = Sv[i] - qcc_recon(pm[0], pm[1], pw2, p); // Q = S - R  (LDPC_Decoder.cu:206-209)
                        const float aq = __builtin_fabsf(q);
                        idx = (aq < m1) ? p : idx;                       // first edge holding the minimum (:298-305)
                        m2 = __builtin_amdgcn_fmed3f(m1, m2, aq);
                        m1 = __builtin_fminf(m1, aq);
                        signs |= (f2u(q) >> 31) << p;
                    }
                }
            // R_p = Sign[25]*Sign[p] * magnitude: output sign bit p = parity of all signs XOR sign p
            if (__builtin_popcount(signs) & 1) signs ^= (1u << w) - 1u;
            if (lane_on) {
                const float nm[2] = {m1, m2};
                lds_st<2>(lds, sidx * 8, nm);
                const float nw[1] = {u2f(signs | ((unsigned)idx << 27))};
                lds_st<1>(lds, off_w2 + sidx * 4, nw);
            }
        }
        __syncthreads();
    }

    {
        bool bad = false;
        if (!stop) vn_phase(bad); // final iteration: VN only
        bad = false;
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            const int l = g + cc * G;
            if (l < L) {
                const int n = l * Z + t;
                float sv[1];
                lds_ld<1>(sv, lds, off_S + n * 4);
                const bool neg = lane_on && sv[0] < 0;
                bad = bad || (n < a.length && neg);
                const unsigned long long m = __ballot(neg); // one 32-bit word per half-wave (Z % 32 == 0)
                if ((tid & 31) == 0 && lane_on) a.bits[(size_t)f * (N / 32) + (n >> 5)] = (unsigned)(m >> (tid & 32));
                if (a.app && lane_on) a.app[(size_t)n * F + f] = sv[0];
            }
        }
        const int last = stop ? stop : a.max_iter;
        if (!stop && bad) lds_flag[last & 1] = 1; // (a stopped frame's verdict is in already)
        __syncthreads();
        const int flag = stop ? 1 : flags_collect(last);
        if (tid == 0) {
            a.D[(size_t)N * F + f] = flag;
            if (HIST && a.hist) a.hist[f] = hist;
            if (HIST && a.per_frame) a.iters[f] = last;
        }
    }
    if (!PERSIST) break;
    __syncthreads(); // the states, S, the flags and the frame word are reused by the next frame
    } // next frame
}

#include "bldpc_qcr_kernel.hpp" // k_qcr: check states in registers, S in LDS (long blocks)
#include "bldpc_qcr2_kernel.hpp" // k_qcr2: the same with hardware-addressed (M0 + lane) LDS accesses

// AND of all frames' flag histories -> first iteration at which every frame's flag is set.
__global__ __launch_bounds__(256) void k_hist_and(const unsigned long long *hist, int F, unsigned long long *out)
{
    unsigned long long x = ~0ull;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) x &= hist[f];
    for (int off = 32; off > 0; off >>= 1) x &= __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAnd(out, x);
}

// Largest per-frame iteration count of a batch.
__global__ __launch_bounds__(256) void k_iters_max(const int *iters, int F, int *out)
{
    int x = 0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) x = max(x, iters[f]);
    for (int off = 32; off > 0; off >>= 1) x = max(x, __shfl_down(x, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, x);
}

// ---------------------------------------------------------------------------------------------
using QcKernel = void (*)(QcArgs);
struct QcVariant { int NF, J, L, Z, WC, WV, G, MINW, threads, lds_bytes; QcKernel fn, fn_hist; const char *tag; int U, CPT, regstate; QcKernel fn_pf = nullptr; int loc = 0; };
// fn_pf: persistent kernel for the per-frame exit (k_qc2p), or null; loc: half-row kernel with local edges (QcGeom2<..., true>)
// tag "compressed" (U != 0): J = L = 0 (any), WC = row slots, lds_bytes computed per code; tag "regstate": L = 0 (any)

// Ahead-of-time variants: one per block-matrix geometry of the reference's matrix set whose message
// state fits one CU's LDS (shifts are run-time data, so every code of the same J x L x Z shape and
// weights <= WC/WV shares a variant).  Everything else runs on the table kernels.
//        NF   J   L    Z  WC WV   G MINW
#define QC_VARIANTS(X)                                                                          \
    X(2,   4, 24,  96, 20, 4,  4, 4) /* J4_L24_Z96 (BASELINE config 2): 384 thr, 81 KB, 2 WG/CU */ \
    X(2,  32, 64,  64,  7, 3, 16, 4) /* J32_L64_Z64 (BASELINE config 3): 1024 thr, 148 KB       */ \
    X(2,   8, 24,  96, 10, 6,  4, 3) /* J8_L24_Z96                                              */ \
    X(2,  12, 24,  96,  7, 6,  6, 2) /* J12_L24_Z96                                             */ \
    X(2,   6, 24,  96, 15, 4,  6, 2) /* J6_L24_Z96                                              */ \
    X(1,   4, 24, 256, 20, 4,  4, 4) /* J4_L24_Z256 (one frame per lane)                        */

inline const QcVariant *qc_variants(int *count)
{
#define X(NF, J, L, Z, WC, WV, G, MINW)                                                                  \
    {NF, J, L, Z, WC, WV, G, MINW, QcGeom<NF, J, L, Z, WC, WV, G, MINW>::TPB,                              \
     QcGeom<NF, J, L, Z, WC, WV, G, MINW>::lds_bytes, k_qc<QcGeom<NF, J, L, Z, WC, WV, G, MINW>, false>,    \
     k_qc<QcGeom<NF, J, L, Z, WC, WV, G, MINW>, true>, "row", 0, 0, 0, k_qcp<QcGeom<NF, J, L, Z, WC, WV, G, MINW>, true>},
#define X2(NF, J, L, Z, WC, WV, GJ, MINW)                                                                 \
    {NF, J, L, Z, WC, WV, GJ, MINW, QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW>::TPB,                            \
     QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW>::lds_bytes, k_qc2<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW>, false>, \
     k_qc2<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW>, true>, "halfrow", 0, 0, 0, k_qc2p<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW>, true>},
#define X1L(NF, J, L, Z, WC, WV, G, MINW)                                                                      \
    {NF, J, L, Z, WC, WV, G, MINW, QcGeom<NF, J, L, Z, WC, WV, G, MINW, true>::TPB,                                \
     QcGeom<NF, J, L, Z, WC, WV, G, MINW, true>::lds_bytes, k_qc<QcGeom<NF, J, L, Z, WC, WV, G, MINW, true>, false>,  \
     k_qc<QcGeom<NF, J, L, Z, WC, WV, G, MINW, true>, true>, "row-local", 0, 0, 0, QC1L_PF(NF, J, L, Z, WC, WV, G, MINW), 2},
#ifdef QC_LOCAL_PER_FRAME_ROW /* experiment: the per-frame exit of the ROW kernel on its local-edge form too (persistent form instantiated) */
#define QC1L_PF(NF, J, L, Z, WC, WV, G, MINW) k_qcp<QcGeom<NF, J, L, Z, WC, WV, G, MINW, true>, true>
#else
#define QC1L_PF(NF, J, L, Z, WC, WV, G, MINW) nullptr
#endif
#define X2L(NF, J, L, Z, WC, WV, GJ, MINW)                                                                      \
    {NF, J, L, Z, WC, WV, GJ, MINW, QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW, true>::TPB,                              \
     QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW, true>::lds_bytes, k_qc2<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW, true>, false>, \
     k_qc2<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW, true>, true>, "halfrow-local", 0, 0, 0,                            \
     k_qc2p<QcGeom2<NF, J, L, Z, WC, WV, GJ, MINW, true>, true>, 1},
#define XC(Z, U, G, CPT, WCS)                                                                              \
    {1, 0, 0, Z, WCS, 31, G, 0, QccGeom<Z, U, G, CPT, WCS>::TPB, 0, k_qcc<QccGeom<Z, U, G, CPT, WCS>, false>,   \
     k_qcc<QccGeom<Z, U, G, CPT, WCS>, true>, "compressed", U, CPT, 0, k_qcc<QccGeom<Z, U, G, CPT, WCS>, true, true>},
#define XR(J, L, Z, TPB, WCS, MINW, YB)                                                                     \
    {1, J, L, Z, WCS, 31, 0, MINW, TPB, 0, k_qcr<QcrGeom<J, L, Z, TPB, WCS, MINW, YB>, false>,               \
     k_qcr<QcrGeom<J, L, Z, TPB, WCS, MINW, YB>, true>, "regstate", 0, 0, 1, k_qcr<QcrGeom<J, L, Z, TPB, WCS, MINW, YB>, true, true>},
#define XR2(J, L, Z, TPB, WCS, YB, NG) /* CPT carries NG here */                                              \
    {1, J, L, Z, WCS, 31, 0, 2, TPB, 0, k_qcr2<Qcr2Geom<J, L, Z, TPB, WCS, YB, NG>, false>,                    \
     k_qcr2<Qcr2Geom<J, L, Z, TPB, WCS, YB, NG>, true>, "regstate-halo", 0, NG, 2, k_qcr2<Qcr2Geom<J, L, Z, TPB, WCS, YB, NG>, true, true>},
    static const QcVariant v[] = {
        X2L(2, 4, 24, 96, 20, 4, 4, 6) /* J4_L24_Z96 (BASELINE config 2) with local edges: taken when every row is full and the matching exists */
        X2(2, 4, 24, 96, 20, 4, 4, 6) /* J4_L24_Z96 (BASELINE config 2): 768 thr, 80 KB, 2 WG/CU, 6 waves/SIMD */
        X2(2, 8, 24, 96, 10, 6, 4, 6) /* J8_L24_Z96: 768 thr, 80 KB, 2 WG/CU                                     */
        X2(2, 12, 24, 96, 8, 6, 4, 3) /* J12_L24_Z96 (rows padded 7 -> 8): 768 thr, 92 KB                        */
        X2(2, 6, 24, 96, 16, 4, 3, 3) /* J6_L24_Z96 (rows padded 15 -> 16): 576 thr, 92 KB                       */
        X1L(2, 32, 64, 64, 7, 3, 16, 4) /* J32_L64_Z64 (BASELINE config 3) with local edges: every column of weight WV, the matching exists */
        QC_VARIANTS(X)
        /* check states in registers, S in LDS (bldpc_qcr_kernel.hpp), several workgroups per CU */
        XR(12, 69, 256, 256, 23, 22, 3) /* PON_LDPC J12_L69_Z256 (the reference's default, define.cuh:20-22): 69 KB, 2 WG/CU */
        XR2(4, 24, 512, 512, 20, 8, 5)   /* J4_L24_Z512: 55 KB, 2 WG/CU; at most 5 wrapped blocks per (row, tile) */
        XR(4, 24, 512, 512, 20, 20, 8)  /* J4_L24_Z512: 48 KB, 3 WG/CU                                                      */
        /* compressed check state (bldpc_qcc_kernel.hpp): any J, L with ceil(L/G) <= CPT and row weight <= WCS */
        XC(256, 256, 4, 18, 24) /* PON_LDPC J12_L69_Z256 (the reference's default, define.cuh:20-22)       */
        XC(160, 192, 5, 12, 24) /* the Z = 160 family, J10 ... J48, L60                                   */
        XC(512, 512, 2, 12, 24) /* J4_L24_Z512                                                            */
        /* lifting sizes outside the reference's matrix set (its users swap matrices by editing define.cuh): the same
         * generic kernel, any J <= 62 and L <= CPT*G whose states fit LDS; everything else runs on the table kernels */
        XC(64, 64, 16, 8, 24)   /* Z =  64: L <= 128 */
        XC(96, 128, 8, 12, 24)  /* Z =  96: L <=  96 */
        XC(128, 128, 8, 12, 24) /* Z = 128: L <=  96 */
        XC(192, 192, 5, 16, 24) /* Z = 192: L <=  80 */
        XC(320, 320, 3, 24, 24) /* Z = 320: L <=  72 */
        XC(384, 384, 2, 32, 24) /* Z = 384: L <=  64 */
        XC(640, 640, 1, 40, 24) /* Z = 640: L <=  40 */
        XC(1024, 1024, 1, 32, 24) /* Z = 1024: L <= 32 */
        /* check states in registers, S with halos in LDS, hardware-addressed accesses (bldpc_qcr2_kernel.hpp) */
        XR2(15, 30, 1280, 768, 8, 10, 2) /* J15_L30_Z1280 (BASELINE config 4): 157.5 KB; at most 2 wrapped blocks per (row, tile) */
        XR2(15, 30, 1280, 768, 8, 10, 3) /* the same shape with other shifts: at most 3 */
        /* check states in registers, S in LDS (bldpc_qcr_kernel.hpp): long blocks with 4 N <= LDS */
        XR(15, 30, 1280, 768, 8, 7, 10) /* J15_L30_Z1280 (BASELINE config 4): 12 waves, 8 of them cover 2 tiles of Z (5 tiles per SIMD) */
    };
#undef X
#undef X2
#undef X2L
#undef X1L
#undef XC
#undef XR
#undef XR2
    *count = (int)(sizeof(v) / sizeof(v[0]));
    return v;
}

constexpr size_t kLdsBytes = 160 * 1024;
#ifdef QC_STAMPS
static unsigned long long *g_qc_stamps = nullptr;
static int g_qc_stagger = 0;
#endif

struct QcPlan {
    int J = 0, L = 0, Z = 0, nnz = 0, Wc = 0, Wv = 0;
    int variant = -1;
    int frames_per_wg = 0; // 0 = unavailable
    QcCnEdge *d_cn = nullptr;
    unsigned short *d_rowptr = nullptr;
    QcVnEdge *d_vn = nullptr;
    unsigned char *d_wv = nullptr;
    unsigned *d_cn_meta = nullptr, *d_vn_meta = nullptr; // compressed-state kernel
    int WVS = 0, lds_bytes = 0, lc = 0;
    char name[96] = "qc_lds(unavailable)";
    mutable int ran_to_max = 0; // BATCH_GLOBAL: the previous batch did not stop before max_iter (a performance hint, never a result)
    int persist_grid = 0; // k_qc2p: workgroups that fill the chip once (a multiple of 8)
    // experiment / test switches, read ONCE when the plan is built (never per decode call):
    bool no_persist = false;    // BLDPC_NO_PERSIST: one workgroup per frame group even where the persistent form exists
    bool force_regroup = false; // BLDPC_REGROUP: k_regroup_y in front of the row / half-row kernels instead of reading in place
                                // (BLDPC_NO_LOCAL, also read there: the half-row kernel without local edges)
    // A ROW-kernel plan with local edges carries the plain plan of the same code for the per-frame exit: its flag-tracking
    // instantiation keeps the branching variable-node phase (one loop per place code would be 210 KB there) and the plain kernel is the
    // faster one for frames that leave after 3 ... 10 iterations (J32_L64_Z64 per-frame 24.5 against 23.4 M codewords/s).  The HALF-ROW
    // kernel serves the per-frame exit with its local-edge form (persistent, k_qc2p<LOC>): 49 / 66 / 71 M against 44 / 59 / 66 M
    // codewords/s at 3.0 / 3.6 / 4.2 dB (it was the other way round, 40 / 53 / 59 M, before the block-row choice left the loop).
    QcPlan *pf = nullptr;
};

// bldpc_decode_statistic: per-frame error counts wanted from the pass that unpacks the hard bits (single-launch modes only).
// Per CALL state (it used to live in the plan, where a concurrent decode on another host thread could pick it up).
struct QcStat {
    int *errs = nullptr; // device int32 [F], all zero on entry
    int length = 0;
    bool done = false;   // set when the unpack pass has accumulated into errs
};

inline void qc_plan_release(QcPlan *q)
{
    if (q->d_cn) (void)hipFree(q->d_cn);
    if (q->d_rowptr) (void)hipFree(q->d_rowptr);
    if (q->d_vn) (void)hipFree(q->d_vn);
    if (q->d_wv) (void)hipFree(q->d_wv);
    if (q->d_cn_meta) (void)hipFree(q->d_cn_meta);
    if (q->d_vn_meta) (void)hipFree(q->d_vn_meta);
    q->d_cn_meta = nullptr; q->d_vn_meta = nullptr;
    q->d_cn = nullptr; q->d_rowptr = nullptr; q->d_vn = nullptr; q->d_wv = nullptr;
    q->frames_per_wg = 0;
    if (q->pf) {
        qc_plan_release(q->pf);
        delete q->pf;
        q->pf = nullptr;
    }
}

// Pick the first variant whose geometry matches the code, upload its block lists.  Leaves
// frames_per_wg == 0 (not an error) when none does.  BLDPC_QC_VARIANT=<index> pins one (experiments).
inline int qc_plan_build(QcPlan *q, int J, int L, int Z, const int *H, bool plain = false)
{
    q->J = J; q->L = L; q->Z = Z;
    std::vector<QcCnEdge> cn;
    std::vector<unsigned short> rowptr(J + 1, 0);
    std::vector<int> wv(L, 0);
    int Wc = 0, Wcmin = 1 << 30;
    if (Z > 65535 || L > 65535) return BLDPC_OK;
    for (int j = 0; j < J; j++) {
        for (int l = 0; l < L; l++)
            if (H[j * L + l] != -1) {
                cn.push_back({(unsigned short)l, (unsigned short)H[j * L + l]});
                wv[l]++;
            }
        if (cn.size() > 65535) return BLDPC_OK;
        rowptr[j + 1] = (unsigned short)cn.size();
        Wc = std::max(Wc, (int)(rowptr[j + 1] - rowptr[j]));
        Wcmin = std::min(Wcmin, (int)(rowptr[j + 1] - rowptr[j]));
    }
    const int nnz = (int)cn.size();
    const int Wv = *std::max_element(wv.begin(), wv.end());
    q->nnz = nnz; q->Wc = Wc; q->Wv = Wv;
    int nvar = 0;
    const QcVariant *vars = qc_variants(&nvar);
    const char *pin = getenv("BLDPC_QC_VARIANT");
    const bool no_halo = getenv("BLDPC_NO_HALO") != nullptr; // tests: k_qcr on a code k_qcr2 takes
    q->no_persist = getenv("BLDPC_NO_PERSIST") != nullptr;
    q->force_regroup = getenv("BLDPC_REGROUP") != nullptr;
    const bool no_local = plain || getenv("BLDPC_NO_LOCAL") != nullptr;
    std::vector<int> owner; // half-row kernel with local edges: the block row every column is handed to
    // k_qcr2 gives per-lane addresses to two slots per (block row, tile): no more than two of a row's blocks may wrap past Z in
    // the same tile of 64 circulant positions (shifts taken relative to the register-resident column, as the kernel sees them)
    auto qcr2_fits = [&](int ng) -> bool {
        int lc = -1;
        for (int l = 0; l < L && lc < 0; l++)
            if (wv[l] == J) lc = l;
        if (lc < 0 || Z % 64 != 0) return false;
        for (int j = 0; j < J; j++) {
            int rot = 0;
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++)
                if (cn[e].col == lc) rot = cn[e].shift;
            for (int t = 0; t < Z / 64; t++) {
                int nw = 0;
                for (int e = rowptr[j]; e < rowptr[j + 1]; e++)
                    if (cn[e].col != lc && (64 * t + (cn[e].shift - rot + Z) % Z) % Z > Z - 64) nw++;
                if (nw > ng) return false;
            }
        }
        return true;
    };
    for (int vi = 0; vi < nvar && q->variant < 0; vi++) {
        const QcVariant &v = vars[vi];
        if (pin && atoi(pin) != vi) continue;
        if (v.regstate) { // register-state kernel
            if (v.regstate == 2 && (!qcr2_fits(v.CPT) || no_halo)) continue; // three blocks of one row wrap in the same tile: k_qcr takes the code (env: tests)
            const size_t lds = v.regstate == 2 ? (size_t)L * (Z + 64) * 4 + 272 : (size_t)L * Z * 4 + 16;
            if (v.J != J || v.L != L || v.Z != Z || v.WC < Wc || v.MINW > Wcmin || L > 255 || Z > 2047 || lds > kLdsBytes) continue;
            q->lds_bytes = (int)lds;
        } else if (v.U) { // compressed-state kernel: geometry-generic
            const size_t lds = (size_t)(J + 1) * Z * 12 + (size_t)(L + 1) * Z * 4 + 16;
            if (v.Z != Z || (L + v.G - 1) / v.G > v.CPT || v.WC < Wc || Wv > 28 || J > 62 || L > 254 || Z > 2047 || lds > kLdsBytes) continue;
            q->lds_bytes = (int)lds;
        } else {
            if (v.J != J || v.L != L || v.Z != Z || v.WC < Wc || v.WV < Wv) continue;
            if ((size_t)v.lds_bytes > kLdsBytes) continue;
            if (v.loc == 1 && (no_local || Wcmin != v.WC || L % (2 * J) != 0 || !qc2_local_assign(J, L, rowptr, cn, owner))) continue; // every row full, every column placed
            if (v.loc == 2 && (no_local || Z % 64 != 0 || *std::min_element(wv.begin(), wv.end()) != v.WV || Wcmin < L / J ||
                               !qc2_local_assign(J, L, rowptr, cn, owner))) continue; // every column full, every column placed
            q->lds_bytes = v.lds_bytes;
        }
        q->variant = vi;
    }
    if (q->variant < 0) return BLDPC_OK;
    const QcVariant &v = vars[q->variant];
    if (v.regstate) { // row slots with first-edge-of-column marks
        // the register-resident column LC: one that meets every block row (prefer the heaviest traffic saved = any such);
        // every block row is rotated until its LC block has shift 0, and lists that block first
        int lc = -1;
        for (int l = 0; l < L && lc < 0; l++)
            if (wv[l] == J) lc = l;
        bool ok = lc >= 0;
        for (int l = 0; l < L; l++) ok = ok && wv[l] > 0; // an unconnected column would keep a stale S
        if (!ok) { q->variant = -1; return BLDPC_OK; }
        std::vector<unsigned> cm((size_t)J * v.WC, qcr_cn_meta(0, 0, 0, 1));
        std::vector<int> seen(L, 0);
        for (int j = 0; j < J; j++) {
            int rot = 0;
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++)
                if (cn[e].col == lc) rot = cn[e].shift;
            int pos = 1;
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++) {
                const int l = cn[e].col;
                seen[l]++;
                cm[(size_t)j * v.WC + (l == lc ? 0 : pos++)] = qcr_cn_meta(l, (cn[e].shift - rot + Z) % Z, seen[l] == 1, 0);
            }
        }
        q->lc = lc;
        if (v.regstate == 2) { // k_qcr2: per (block row, tile, slot) the byte offset of the wave's 64 rotated positions, and the flags
            const int NT = Z / 64, ZH = Z + 64, WCS = v.WC;
            const unsigned inf_base = (unsigned)(L * ZH * 4); // 64 words of +inf: what a padding slot addresses
            std::vector<unsigned> ta((size_t)J * NT * WCS, inf_base), tx((size_t)J * NT * WCS, inf_base);
            bool fits = true;
            for (int j = 0; j < J && fits; j++)
                for (int t = 0; t < NT && fits; t++) {
                    // slots of this (block row, tile): the blocks whose 64 positions wrap past Z go last (the last NG slots take
                    // per-lane addresses in phase 2), the others first, padding in between (plain + wrapped <= WCS - 1: no overlap)
                    std::vector<std::pair<int, int>> plain, wrapped; // (column, rb)
                    for (int p = 1; p < WCS; p++) {
                        const unsigned m = cm[(size_t)j * WCS + p];
                        if ((m >> 21) & 1u) continue;
                        const int col = (int)(m & 255u), rb = (64 * t + (int)((m >> 8) & 2047u)) % Z;
                        (rb > Z - 64 ? wrapped : plain).push_back({col, rb});
                    }
                    if ((int)wrapped.size() > v.CPT) { fits = false; break; }
                    unsigned *a1 = &ta[((size_t)j * NT + t) * WCS], *a2 = &tx[((size_t)j * NT + t) * WCS];
                    int slot = 1;
                    for (auto &b : plain) {
                        a1[slot] = a2[slot] = (unsigned)((b.first * ZH + b.second) * 4);
                        slot++;
                    }
                    for (int g = 0; g < v.CPT; g++) // the per-lane slots, whatever they hold: offset | first wrapped lane << 18 (64: no lane wraps)
                        a2[WCS - 1 - g] = (a2[WCS - 1 - g] & 0x3ffffu) | (64u << 18);
                    for (size_t k = 0; k < wrapped.size(); k++) {
                        const int gs = WCS - 1 - (int)k;
                        a1[gs] = (unsigned)((wrapped[k].first * ZH + wrapped[k].second) * 4);
                        a2[gs] = a1[gs] | ((unsigned)(Z - wrapped[k].second) << 18); // lanes from Z - rb on wrap
                    }
                }
            if (!fits) { q->variant = -1; return BLDPC_OK; } // (the selection loop has checked: cannot happen)
            if (L * ZH * 4 + 272 >= (1 << 18)) { q->variant = -1; return BLDPC_OK; } // offsets are 18-bit fields
            CLDPC_HIP(hipMalloc((void **)&q->d_vn_meta, tx.size() * sizeof(unsigned)), BLDPC_ENOMEM); // one table for both phases (ta = tx without the lane tags)
            CLDPC_HIP(hipMemcpy(q->d_vn_meta, tx.data(), tx.size() * sizeof(unsigned), hipMemcpyHostToDevice), BLDPC_EHIP);
        } else {
            CLDPC_HIP(hipMalloc((void **)&q->d_cn_meta, cm.size() * sizeof(unsigned)), BLDPC_ENOMEM);
            CLDPC_HIP(hipMemcpy(q->d_cn_meta, cm.data(), cm.size() * sizeof(unsigned), hipMemcpyHostToDevice), BLDPC_EHIP);
        }
    }
    if (v.U) { // meta tables of the compressed-state kernel
        const int WVS = (Wv + 1) / 2 * 2; // column edge lists padded to whole rounds of 2 with entries of the zero state (row J)
        q->WVS = WVS;
        std::vector<unsigned> cm((size_t)J * v.WC + J, qcc_cn_meta(L, 0)), vm((size_t)L * WVS, qcc_vn_meta(J, 0, 0));
        std::vector<int> fillc(L, 0);
        for (int j = 0; j < J; j++) {
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++) {
                const int pos = e - rowptr[j], l = cn[e].col;
                cm[(size_t)j * v.WC + pos] = qcc_cn_meta(l, cn[e].shift);
                vm[(size_t)l * WVS + fillc[l]++] = qcc_vn_meta(j, pos, cn[e].shift); // ascending j = the reference's edge order
            }
            cm[(size_t)J * v.WC + j] = (unsigned)(rowptr[j + 1] - rowptr[j]); // row weights behind the slots
        }
        CLDPC_HIP(hipMalloc((void **)&q->d_cn_meta, cm.size() * sizeof(unsigned)), BLDPC_ENOMEM);
        CLDPC_HIP(hipMalloc((void **)&q->d_vn_meta, vm.size() * sizeof(unsigned)), BLDPC_ENOMEM);
        CLDPC_HIP(hipMemcpy(q->d_cn_meta, cm.data(), cm.size() * sizeof(unsigned), hipMemcpyHostToDevice), BLDPC_EHIP);
        CLDPC_HIP(hipMemcpy(q->d_vn_meta, vm.data(), vm.size() * sizeof(unsigned), hipMemcpyHostToDevice), BLDPC_EHIP);
    }
    const bool generic = v.U || v.regstate;
    const int vnw = generic ? 1 : v.WV;
    // Row variant with wave-uniform rows (Z whole waves) and several rows per thread: the kernel runs ONE body for all of a
    // thread's rows, sized for the heaviest of them, so that their loads are in flight together.  Which check rows a thread owns is
    // free (a row is only a name for a set of R slots): hand every thread group rows of equal weight where the weights allow it --
    // "virtual" row i + rr*G = the (i*RPT + rr)-th row in descending weight order.  The ORDER of a column's edges stays the
    // reference's, ascending real block row: only the slot a block's messages live in changes.
    std::vector<int> virt_of(J);
    for (int j = 0; j < J; j++) virt_of[j] = j;
    if (!generic && std::string(v.tag).compare(0, 3, "row") == 0 && Z % 64 == 0 && J / v.G >= 2) {
        const int G = v.G, RPT = J / G;
        std::vector<int> order(J);
        for (int j = 0; j < J; j++) order[j] = j;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rowptr[x + 1] - rowptr[x] > rowptr[y + 1] - rowptr[y]; });
        if (v.loc == 2) { // local edges: the rows of a thread group in lexicographic order of their local blocks' (sorted) places, see qc_kcode_sorted
            auto place = [&](int l) { int o = 0; for (int j2 = 0; j2 < owner[l]; j2++) o += (H[j2 * L + l] != -1) ? 1 : 0; return o; };
            auto tuple_of = [&](int j) { std::vector<int> t; for (int l = 0; l < L; l++) if (owner[l] == j) t.push_back(place(l)); std::sort(t.begin(), t.end()); return t; };
            for (int k0 = 0; k0 + RPT <= J; k0 += RPT)
                std::stable_sort(order.begin() + k0, order.begin() + k0 + RPT, [&](int x, int y) { return tuple_of(x) < tuple_of(y); });
        }
        for (int k = 0; k < J; k++) virt_of[order[k]] = (k / RPT) + (k % RPT) * G;
        std::vector<int> real_of(J);
        for (int j = 0; j < J; j++) real_of[virt_of[j]] = j;
        std::vector<QcCnEdge> cn2;
        std::vector<unsigned short> rowptr2(J + 1, 0);
        for (int vj = 0; vj < J; vj++) {
            const int j = real_of[vj];
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++) cn2.push_back(cn[e]);
            rowptr2[vj + 1] = (unsigned short)cn2.size();
        }
        cn = cn2;
        rowptr = rowptr2;
    }
    std::vector<int> virt_col(L);
    for (int l = 0; l < L; l++) virt_col[l] = l;
    if (v.loc == 2) {
        // Row kernel: virtual row vj = g + rr*G lists its NLR local blocks first (the order of a row's slots is free); the column of
        // local block i is the thread group's virtual column g + (rr*NLR + i)*G.
        const int NLR = L / J, G = v.G;
        std::vector<int> real_of(J);
        for (int j = 0; j < J; j++) real_of[virt_of[j]] = j;
        for (int vj = 0; vj < J; vj++) {
            std::vector<QcCnEdge> loc, oth;
            for (int e = rowptr[vj]; e < rowptr[vj + 1]; e++) (owner[cn[e].col] == real_of[vj] ? loc : oth).push_back(cn[e]);
            auto place = [&](int l) { int o = 0; for (int j2 = 0; j2 < owner[l]; j2++) o += (H[j2 * L + l] != -1) ? 1 : 0; return o; };
            std::stable_sort(loc.begin(), loc.end(), [&](const QcCnEdge &x, const QcCnEdge &y) { return place(x.col) < place(y.col); }); // by place in the column's order
            for (int i = 0; i < NLR; i++) {
                cn[rowptr[vj] + i] = loc[i];
                virt_col[loc[i].col] = (vj % G) + ((vj / G) * NLR + i) * G;
            }
            for (size_t i = 0; i < oth.size(); i++) cn[rowptr[vj] + NLR + i] = oth[i];
        }
    }
    if (v.loc == 1) {
        // Half-row (j, h) lists its CPT local blocks first, then its share of the row's other blocks (the order of a row's slots is
        // free: min1 / min2 / sign product are symmetric, a duplicated minimum gives min1 == min2); the column of local block cc is the
        // thread group's virtual column (2j + h) + cc * 2J.
        const int CPT = L / (2 * J), WCH = v.WC / 2;
        std::vector<QcCnEdge> cn2(cn.size());
        for (int j = 0; j < J; j++) {
            std::vector<QcCnEdge> loc, oth;
            for (int e = rowptr[j]; e < rowptr[j + 1]; e++) (owner[cn[e].col] == j ? loc : oth).push_back(cn[e]);
            for (int h = 0; h < 2; h++) {
                QcCnEdge *dst = &cn2[rowptr[j] + h * WCH];
                for (int cc = 0; cc < CPT; cc++) {
                    dst[cc] = loc[h * CPT + cc];
                    virt_col[loc[h * CPT + cc].col] = (2 * j + h) + cc * 2 * J;
                }
                for (int i = CPT; i < WCH; i++) dst[i] = oth[h * (WCH - CPT) + (i - CPT)];
            }
        }
        cn = cn2;
    }
    std::vector<QcVnEdge> vn((size_t)L * vnw, v.loc == 1 ? QcVnEdge{0xffff, 0} : QcVnEdge{0, 0});
    std::vector<int> fill(L, 0), fill_nl(L, 0), kloc(L, 0);
    for (int j = 0; j < J; j++) { // ascending REAL block row = the reference's edge order
        const int vj = virt_of[j];
        for (int e = rowptr[vj]; e < rowptr[vj + 1]; e++) {
            const int l = cn[e].col;
            // .e = padded block index (virtual row)*WC + position
            if (v.loc == 2) { // the column's other blocks in ascending real block row; kloc = where its local block stands among them
                if (owner[l] == j) kloc[virt_col[l]] = fill[l];
                else vn[(size_t)virt_col[l] * v.WV + fill_nl[l]++] = {(unsigned short)(vj * v.WC + (e - rowptr[vj])), cn[e].shift};
                fill[l]++;
            } else if (v.loc == 1) vn[(size_t)virt_col[l] * v.WV + j] = {(unsigned short)(j * v.WC + (e - rowptr[j])), cn[e].shift}; // slot k = block row k
            else if (!generic) vn[(size_t)l * v.WV + fill[l]++] = {(unsigned short)(vj * v.WC + (e - rowptr[vj])), cn[e].shift};
        }
    }
    std::vector<unsigned char> wvb(L);
    for (int l = 0; l < L; l++) wvb[l] = (unsigned char)(v.loc == 2 ? kloc[l] : wv[l]); // row kernel with local edges: per VIRTUAL column, the place of its local block
    CLDPC_HIP(hipMalloc((void **)&q->d_cn, cn.size() * sizeof(QcCnEdge)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_rowptr, rowptr.size() * sizeof(unsigned short)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_vn, vn.size() * sizeof(QcVnEdge)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_wv, wvb.size()), BLDPC_ENOMEM);
    CLDPC_HIP(hipMemcpy(q->d_cn, cn.data(), cn.size() * sizeof(QcCnEdge), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_rowptr, rowptr.data(), rowptr.size() * sizeof(unsigned short), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_vn, vn.data(), vn.size() * sizeof(QcVnEdge), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_wv, wvb.data(), wvb.size(), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, generic ? (int)kLdsBytes : v.lds_bytes), BLDPC_EHIP);
    CLDPC_HIP(hipFuncSetAttribute((const void *)v.fn_hist, hipFuncAttributeMaxDynamicSharedMemorySize, generic ? (int)kLdsBytes : v.lds_bytes), BLDPC_EHIP);
    if (v.fn_pf) {
        CLDPC_HIP(hipFuncSetAttribute((const void *)v.fn_pf, hipFuncAttributeMaxDynamicSharedMemorySize, generic ? (int)kLdsBytes : v.lds_bytes), BLDPC_EHIP);
        int occ = 0, dev = 0, ncu = 0;
        CLDPC_HIP(hipGetDevice(&dev), BLDPC_EHIP);
        CLDPC_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev), BLDPC_EHIP);
        CLDPC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)v.fn_pf, v.threads, (size_t)q->lds_bytes), BLDPC_EHIP);
        q->persist_grid = std::max(8, ncu * std::max(1, occ) / 8 * 8);
    }
    q->frames_per_wg = v.NF;
    if (v.loc == 2 && !plain && !getenv("BLDPC_LOCAL_PER_FRAME")) { // the per-frame exit's plan of the ROW kernel (see QcPlan::pf); without it the local-edge kernels serve that mode too
        q->pf = new QcPlan();
        const int rp = qc_plan_build(q->pf, J, L, Z, H, true);
        if (rp || q->pf->frames_per_wg != v.NF || qc_variants(&nvar)[q->pf->variant].loc) {
            qc_plan_release(q->pf);
            delete q->pf;
            q->pf = nullptr;
            if (rp) return rp;
        }
    }
    snprintf(q->name, sizeof(q->name), "qc_lds_%s<nf%d,J%d,L%d,Z%d,wc%d,wv%d,g%d,w%d>t%d_lds%d", v.tag, v.NF, J, L, v.Z, v.WC, generic ? Wv : v.WV,
             v.G, v.MINW, v.threads, q->lds_bytes);
    return BLDPC_OK;
}

inline bool qc_reads_in_place(const QcPlan *q)
{
    int nvar = 0;
    const QcVariant &v = qc_variants(&nvar)[q->variant];
    return v.NF == 2 && !v.U && !v.regstate; // the row and half-row kernels
}

inline int qc_regroup(const QcPlan *q, const float *y, float *yg, int F, hipStream_t st)
{
    int nvar = 0;
    const QcVariant &v = qc_variants(&nvar)[q->variant];
    const int N = q->L * q->Z;
    const dim3 grid((unsigned)((F + 63) / 64), (unsigned)((N + 63) / 64));
    if (v.NF == 2) hipLaunchKernelGGL(k_regroup_y<2>, grid, dim3(256), 0, st, y, yg, N, F);
    else hipLaunchKernelGGL(k_regroup_y<1>, grid, dim3(256), 0, st, y, yg, N, F);
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    return BLDPC_OK;
}

// y here is the regrouped buffer produced by qc_regroup.
inline int qc_launch(const QcPlan *q, const float *y, int F, int max_iter, int length, int *D, float *app,
                     unsigned long long *hist, unsigned *bits, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                     int *iters = nullptr, bool expand = true, int *work = nullptr, QcStat *stat = nullptr, bool y_in_place = false)
{
    int nvar = 0;
    const QcVariant &v = qc_variants(&nvar)[q->variant];
    QcArgs a;
    a.y = y; a.y_raw = y_in_place ? y : nullptr; a.D = D; a.bits = bits; a.app = app; a.hist = hist;
    a.per_frame = (iters && hist) ? 1 : 0; a.iters = iters; // per-frame exit lives in the flag-tracking instantiation
    a.cn_edges = q->d_cn; a.rowptr = q->d_rowptr; a.vn_edges = q->d_vn; a.wv = q->d_wv;
    a.F = F;
    a.nWG = (F + q->frames_per_wg - 1) / q->frames_per_wg;
    a.max_iter = max_iter; a.length = length;
    a.cn_meta = q->d_cn_meta; a.vn_meta = q->d_vn_meta; a.J = q->J; a.L = q->L; a.WVS = q->WVS; a.lc = q->lc;
#ifdef QC_STAMPS
    a.stamps = g_qc_stamps; a.stagger = g_qc_stagger;
#endif
    unsigned grid = (unsigned)((a.nWG + 7) / 8 * 8);
    QcKernel fn = hist ? v.fn_hist : v.fn;
    if (a.per_frame && v.fn_pf && work && q->persist_grid > 0 && grid > (unsigned)q->persist_grid && !q->no_persist) {
        // per-frame exit on the half-row kernel: persistent workgroups, one frame-pair counter per XCD (k_qc2p)
        CLDPC_HIP(hipMemsetAsync(work, 0, 8 * sizeof(int), st), BLDPC_EHIP);
        a.work = work;
        fn = v.fn_pf;
        grid = (unsigned)q->persist_grid;
    }
    if (ev0) (void)hipEventRecord(ev0, st);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(v.threads), q->lds_bytes, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    const int NW = q->L * q->Z / 32;
    if (expand) {
        hipLaunchKernelGGL(k_expand_bits, dim3((unsigned)((F + 1023) / 1024), (unsigned)NW), dim3(256), 0, st, bits, D, F, NW, stat ? stat->errs : nullptr,
                           stat ? stat->length : 0);
        if (stat && stat->errs) stat->done = true;
    }
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    return BLDPC_OK;
}

// hist_ws: device uint64[F] workspace; and_ws: device uint64; bits: device uint32 [F][N/32] workspace;
// yg: device float [ceil(F/NF)*NF][N] workspace for the regrouped channel values.
inline int qc_decode(const QcPlan *q, const float *y, int F, int max_iter, int length, int exit_mode, int *D, float *app,
                     unsigned long long *flag_hist, unsigned long long *hist_ws, unsigned long long *and_ws, unsigned *bits,
                     float *yg, int *itera, int *iters, int *iters_ws, hipStream_t st, hipEvent_t ev0 = nullptr,
                     hipEvent_t ev1 = nullptr, QcStat *stat = nullptr, const char **used = nullptr)
{
    const QcPlan *qf = q->pf ? q->pf : q; // the plan of the per-frame passes
    if (used) *used = (exit_mode == BLDPC_EXIT_PER_FRAME) ? qf->name : q->name;
    // k_qc / k_qc2 carry two frames per lane: with F even (and the frame-fastest rows 8-byte aligned) a lane's pair of channel
    // values is 8 contiguous bytes of the reference's own layout and the kernels read it in place -- every 64-byte sector is
    // shared by the 4 workgroups of 8 neighbouring frames, which the XCD-aware block order puts on one L2 -- instead of paying a
    // separate pass that reads and writes the whole input (0.24 ms of a 5.9 ms step at config 2).
    const bool in_place = qc_reads_in_place(q) && (F % 2 == 0) && ((uintptr_t)y % 8 == 0) && !q->force_regroup;
    if (!in_place) {
        int rr = qc_regroup(q, y, yg, F, st);
        if (rr) return rr;
        y = yg;
    }
    if (exit_mode == BLDPC_EXIT_FIXED) {
        *itera = max_iter;
        return qc_launch(q, y, F, max_iter, length, D, app, flag_hist, bits, st, ev0, ev1, nullptr, true, nullptr, stat, in_place);
    }
    if (exit_mode == BLDPC_EXIT_PER_FRAME) { // every workgroup leaves when its own frames have stopped; nothing to wait for
        *itera = max_iter;
        return qc_launch(qf, y, F, max_iter, length, D, app, flag_hist ? flag_hist : hist_ws, bits, st, ev0, ev1, iters, true, (int *)and_ws, stat, in_place);
    }
    // Reference rule (LDPC_Decoder.cu:150-153): stop after the first iteration at which ALL frames are flagged.  No
    // workgroup can know that iteration while it runs, so it is found first and the batch then decoded with exactly that
    // many iterations:
    //   pass 1  per-frame exit (cheap: every workgroup leaves when its own frames are flagged) -> m = the latest
    //           first-flag iteration of any frame (max_iter for a frame that never flags).  The batch cannot stop before m.
    //   pass 2  `run` = m iterations with the flag history on: if every frame is flagged at some iteration <= run (usually
    //           exactly at m) that is the stop iteration -- replayed if it is not `run` itself.  Otherwise a frame has lost
    //           its flag again: double `run` and repeat; `run` = max_iter ends the search.
    if (max_iter > 64) return fail(BLDPC_EUNSUPPORTED, "QC_LDS with BATCH_GLOBAL exit supports max_iter <= 64 (got %d)", max_iter);
    unsigned long long *hist = flag_hist ? flag_hist : hist_ws;
    // A batch that holds a frame which never passes costs the per-frame pass for nothing (its answer is max_iter); sweeps
    // stay in that regime for many batches in a row, so after such a batch the full run comes first.
    auto all_flagged = [&](int iters_run, unsigned long long *all) -> int { // AND of the histories, first iters_run bits
        CLDPC_HIP(hipMemsetAsync(and_ws, 0xFF, sizeof(unsigned long long), st), BLDPC_EHIP);
        hipLaunchKernelGGL(k_hist_and, dim3(std::min((F + 255) / 256, 1024)), dim3(256), 0, st, hist, F, and_ws);
        CLDPC_HIP(hipMemcpyAsync(all, and_ws, sizeof(*all), hipMemcpyDeviceToHost, st), BLDPC_EHIP);
        CLDPC_HIP(hipStreamSynchronize(st), BLDPC_EHIP);
        if (iters_run < 64) *all &= ((1ull << iters_run) - 1);
        return BLDPC_OK;
    };
    int r;
    int run = max_iter;
    if (!q->ran_to_max) {
        r = qc_launch(qf, y, F, max_iter, length, D, nullptr, hist, bits, st, nullptr, nullptr, iters_ws, /*expand=*/false, (int *)and_ws, nullptr, in_place);
        if (r) return r;
        int m = 0;
        CLDPC_HIP(hipMemsetAsync(and_ws, 0, sizeof(unsigned long long), st), BLDPC_EHIP);
        hipLaunchKernelGGL(k_iters_max, dim3(std::min((F + 255) / 256, 1024)), dim3(256), 0, st, iters_ws, F, (int *)and_ws);
        CLDPC_HIP(hipMemcpyAsync(&m, and_ws, sizeof(int), hipMemcpyDeviceToHost, st), BLDPC_EHIP);
        CLDPC_HIP(hipStreamSynchronize(st), BLDPC_EHIP);
        if (m < 1 || m > max_iter) return fail(BLDPC_EHIP, "per-frame pass returned iteration count %d", m);
        run = m;
    }
    for (;; run = std::min(max_iter, std::max(run + 4, 2 * run))) {
        if ((r = qc_launch(q, y, F, run, length, D, app, hist, bits, st, ev0, ev1, nullptr, true, nullptr, nullptr, in_place))) return r;
        unsigned long long all = 0; // bit it-1: every frame flagged after iteration it
        if ((r = all_flagged(run, &all))) return r;
        if (all) {
            const int stop = __builtin_ctzll(all) + 1;
            *itera = stop;
            q->ran_to_max = (stop == max_iter);
            return stop < run ? qc_launch(q, y, F, stop, length, D, app, flag_hist, bits, st, ev0, ev1, nullptr, true, nullptr, nullptr, in_place) : BLDPC_OK;
        }
        if (run == max_iter) {
            *itera = max_iter;
            q->ran_to_max = 1;
            return BLDPC_OK;
        }
    }
}

} // namespace cldpc
