// bldpc_qc_kernel.hpp -- fused QC-LDPC flooding min-sum kernel (BLDPC_KERNEL_QC_LDS).
//
// One workgroup decodes NF*FP frames for ALL iterations without touching HBM in
// between: channel values and the check-node outputs live in registers, the
// messages in flight live in LDS.  Same arithmetic, in the same order, as the
// reference's two kernels per iteration (bldpc_实习/LDPC_Decoder.cu:172-372),
// none of their structure.
//
// Mapping.  A block (j,l) with shift s connects check row r to variable column
// (r+s) mod Z.  Lanes run along the circulant dimension: thread (g, p, t) owns
//   * check rows   (j, t) for j = g, g+G, ...   of frame group p   (CN phase)
//   * variables    (l, t) for l = g, g+G, ...   of frame group p   (VN phase)
// and every lane carries NF (1 or 2) frames side by side, so LDS traffic is
// ds_read_b64/ds_write_b64 and the VN sums are packed adds.  U = FP*Z is a
// multiple of 64, hence g is wave-uniform and all table look-ups are scalar.
//
// Message exchange ("APP exchange", bit-identical to the reference's in-place
// R/Q memory, SURVEY Appendix A note):
//   VN phase: S = ((0+R_0)+R_1+...)+y in ascending block-row order (A.2); the
//             thread reads R through the rotation (c-s) mod Z and publishes S
//             aligned (one LDS write per VARIABLE instead of one per EDGE).
//   CN phase: the check thread keeps its own last outputs R_p in registers,
//             reads S of its neighbours through the rotation (r+s) mod Z, forms
//             Q_p = S - R_p (the value the reference's VN kernel would have stored,
//             LDPC_Decoder.cu:206-209), runs min-sum, publishes R aligned.
// LDS per lane-slot: (E + N) * 4 * NF bytes per frame group instead of HBM
// traffic of 16E + 8N bytes per frame and iteration.
#pragma once
#include <algorithm>
#include <vector>

#include "../../include/bldpc.h"
#include "bldpc_math.hpp"
#include "common.hpp"

namespace cldpc {

struct QcCnEdge { unsigned short col, shift; };  // block-row-major list of non-zero blocks
struct QcVnEdge { unsigned short e, shift; };    // per column, top->bottom: block index e into the CN list

struct QcArgs {
    const float *y;             // [N][F]
    int *D;                     // [N+1][F]
    float *app;                 // [N][F] or nullptr
    unsigned long long *hist;   // [F] flag history or nullptr
    const QcCnEdge *cn_edges;   // [nnz]
    const unsigned short *rowptr; // [J+1]
    const QcVnEdge *vn_edges;   // [L][WV]
    const unsigned char *wv;    // [L]
    int J, L, Z, F, FP, G, U, nWG, max_iter, length, nnz;
};

template <int NF> struct Msg;
template <> struct Msg<1> { using T = float; };
template <> struct Msg<2> { using T = float2; };

template <int NF> __device__ __forceinline__ void lds_ld(float (&d)[NF], const float *lds, int idx)
{
    typename Msg<NF>::T v = *reinterpret_cast<const typename Msg<NF>::T *>(lds + idx);
    __builtin_memcpy(d, &v, sizeof(v));
}
template <int NF> __device__ __forceinline__ void lds_st(float *lds, int idx, const float (&s)[NF])
{
    typename Msg<NF>::T v;
    __builtin_memcpy(&v, s, sizeof(v));
    *reinterpret_cast<typename Msg<NF>::T *>(lds + idx) = v;
}

// Packed pairs of 16-bit LDS slot indices (one slot = NF floats); halves the address registers.
template <int NF> __device__ __forceinline__ int slot_lo(unsigned pk) { return (int)(pk & 0xffffu) * NF; }
template <int NF> __device__ __forceinline__ int slot_hi(unsigned pk) { return (int)(pk >> 16) * NF; }

// NF frames per lane, RPT check rows per thread (max), WC max row weight,
// CPT variable columns per thread (max), WV max column weight, TPB threads per workgroup,
// HIST: record the per-iteration termination flags (flag history / batch-global exit).
//
// The iteration loop is branch-free below row/column granularity: rows lighter than WC are padded
// with dummy edges that read a +inf slot (neutral for min1/min2/sign) and write into the row's own
// padding blocks; columns lighter than WV read a slot that always holds +0.0f (adding +0.0f to a sum
// that is never -0.0f is exact), so the compiler can keep all LDS reads of a phase in flight.
template <int NF, int RPT, int WC, int CPT, int WV, int TPB, bool HIST>
__global__ __launch_bounds__(TPB) void k_qc(QcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-aware workgroup id: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
    // contiguous range of frames so the 4-byte-per-frame rows of y / D are completed in one L2.
    const int chunk = (a.nWG + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (wg >= a.nWG) return;

    const int Z = a.Z, FP = a.FP, G = a.G, U = a.U, F = a.F;
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid / U); // wave-uniform (U % 64 == 0)
    const int u = tid - g * U;
    const int p = u / Z, t = u - p * Z;
    const int f0 = (wg * FP + p) * NF;      // first frame carried by this lane
    const int strideS = FP * Z;              // slots between consecutive blocks / columns
    const int lane_slot = p * Z + t;         // aligned position of this lane inside a block
    const int Sslot = a.J * WC * strideS;    // Rbuf: J rows x WC blocks (row-padded), then Sbuf: L columns
    const int zero_slot = Sslot + a.L * strideS; // always +0.0f
    const int inf_slot = zero_slot + 1;          // always +inf
    const int trash_slot = inf_slot + 1;         // one block: S of padding columns (l >= L) lands here
    int *lds_flag = reinterpret_cast<int *>(lds + (trash_slot + strideS) * NF); // [FP*NF]

    // ---- prologue: per-thread edge addresses, channel values, zeroed R ------------------
    float Rreg[RPT][WC][NF];
    unsigned saddr[RPT][(WC + 1) / 2];
    int rbase[RPT];
    bool rvalid[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
        const int j = g + rr * G;
        rvalid[rr] = j < a.J;
        rbase[rr] = (j * WC * strideS + lane_slot) * NF;
        const int e0 = rvalid[rr] ? a.rowptr[j] : 0;
        const int w = rvalid[rr] ? a.rowptr[j + 1] - e0 : 0;
#pragma unroll
        for (int pp = 0; pp < WC; pp++) {
            int slot = inf_slot;
            if (pp < w) {
                const QcCnEdge ed = a.cn_edges[e0 + pp];
                int c = t + ed.shift;
                c = (c >= Z) ? c - Z : c;
                slot = Sslot + ed.col * strideS + p * Z + c;
            }
            if (pp & 1) saddr[rr][pp / 2] |= (unsigned)slot << 16;
            else saddr[rr][pp / 2] = (unsigned)slot;
#pragma unroll
            for (int v = 0; v < NF; v++) Rreg[rr][pp][v] = 0.0f;
            if (rvalid[rr]) {
                const float zero[NF] = {};
                lds_st<NF>(lds, rbase[rr] + pp * strideS * NF, zero); // Memory_RQ = 0 (LDPC_Decoder.cu:82)
            }
        }
    }
    float yreg[CPT][NF];
    unsigned raddr[CPT][(WV + 1) / 2];
    bool cvalid[CPT];
    int swrite[CPT]; // float index of this lane's S slot per column (trash block for padding columns)
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        const int l = g + cc * G;
        cvalid[cc] = l < a.L;
        const int w = cvalid[cc] ? a.wv[l] : 0;
        swrite[cc] = ((cvalid[cc] ? Sslot + l * strideS : trash_slot) + lane_slot) * NF;
#pragma unroll
        for (int k = 0; k < WV; k++) {
            int slot = zero_slot;
            if (k < w) {
                const QcVnEdge ed = a.vn_edges[l * WV + k]; // ed.e = row * WC + position
                int r = t - ed.shift;
                r = (r < 0) ? r + Z : r;
                slot = ed.e * strideS + p * Z + r;
            }
            if (k & 1) raddr[cc][k / 2] |= (unsigned)slot << 16;
            else raddr[cc][k / 2] = (unsigned)slot;
        }
#pragma unroll
        for (int v = 0; v < NF; v++) {
            yreg[cc][v] = 0.0f;
            if (cvalid[cc] && f0 + v < F) yreg[cc][v] = a.y[(size_t)(l * Z + t) * F + f0 + v];
        }
    }
    if (tid < NF) {
        lds[zero_slot * NF + tid] = 0.0f;
        lds[inf_slot * NF + tid] = __builtin_inff();
    }
    if (tid < FP * NF) lds_flag[tid] = 0;
    unsigned long long hist = 0; // used by threads tid < FP*NF
    __syncthreads();

    // VN phase (LDPC_Decoder.cu:188-210): S = ((0+R_0)+...+R_{w-1})+y, published aligned.
    auto vn_phase = [&](bool (&bad)[NF]) {
        float R[CPT][WV][NF];
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
#pragma unroll
            for (int k = 0; k < WV; k++)
                lds_ld<NF>(R[cc][k], lds, (k & 1) ? slot_hi<NF>(raddr[cc][k / 2]) : slot_lo<NF>(raddr[cc][k / 2]));
        }
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            float S[NF];
#pragma unroll
            for (int v = 0; v < NF; v++) S[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < WV; k++) {
#pragma unroll
                for (int v = 0; v < NF; v++) S[v] += R[cc][k][v];
            }
#pragma unroll
            for (int v = 0; v < NF; v++) S[v] += yreg[cc][v];
            lds_st<NF>(lds, swrite[cc], S);
            if (HIST) {
                const bool in_len = cvalid[cc] && ((g + cc * G) * Z + t) < a.length;
#pragma unroll
                for (int v = 0; v < NF; v++) bad[v] = bad[v] || (in_len && S[v] < 0);
            }
        }
    };

    // per-iteration flag bookkeeping (LDPC_Decoder.cu:137-147); call between VN phase and barrier / after it
    auto flags_publish = [&](const bool (&bad)[NF]) {
#pragma unroll
        for (int v = 0; v < NF; v++)
            if (bad[v]) lds_flag[p * NF + v] = 1; // same value from every writer
    };
    auto flags_collect = [&](int it) -> int {
        int flag = 0;
        if (tid < FP * NF) {
            flag = lds_flag[tid] ? 0 : 1;
            lds_flag[tid] = 0;
            if (flag && it <= 64) hist |= (1ull << (it - 1));
        }
        return flag;
    };

    // ---- iterations 1 .. max_iter-1: VN, CN ----------------------------------------------
    for (int it = 1; it < a.max_iter; it++) {
        // keep the 16-bit-packed slot indices packed across iterations (the unpacked form costs 2x the VGPRs)
#pragma unroll
        for (int rr = 0; rr < RPT; rr++)
#pragma unroll
            for (int i = 0; i < (WC + 1) / 2; i++) asm volatile("" : "+v"(saddr[rr][i]));
#pragma unroll
        for (int cc = 0; cc < CPT; cc++)
#pragma unroll
            for (int i = 0; i < (WV + 1) / 2; i++) asm volatile("" : "+v"(raddr[cc][i]));
        bool bad[NF];
#pragma unroll
        for (int v = 0; v < NF; v++) bad[v] = false;
        vn_phase(bad);
        if (HIST) flags_publish(bad);
        __syncthreads();
        if (HIST) (void)flags_collect(it);

        // CN phase (LDPC_Decoder.cu:279-314)
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
            if (rvalid[rr]) {
                float Q[WC][NF];
#pragma unroll
                for (int pp = 0; pp < WC; pp++)
                    lds_ld<NF>(Q[pp], lds, (pp & 1) ? slot_hi<NF>(saddr[rr][pp / 2]) : slot_lo<NF>(saddr[rr][pp / 2]));
                CnAcc acc[NF];
#pragma unroll
                for (int v = 0; v < NF; v++) acc[v].init();
#pragma unroll
                for (int pp = 0; pp < WC; pp++) {
#pragma unroll
                    for (int v = 0; v < NF; v++) {
                        Q[pp][v] = Q[pp][v] - Rreg[rr][pp][v]; // Q = S - R  (LDPC_Decoder.cu:206-209)
                        acc[v].add(Q[pp][v]);
                    }
                }
                uint32_t key[NF];
#pragma unroll
                for (int v = 0; v < NF; v++) key[v] = acc[v].key();
#pragma unroll
                for (int pp = 0; pp < WC; pp++) {
#pragma unroll
                    for (int v = 0; v < NF; v++) Rreg[rr][pp][v] = cn_out(Q[pp][v], acc[v].m2, key[v]);
                    lds_st<NF>(lds, rbase[rr] + pp * strideS * NF, Rreg[rr][pp]);
                }
            }
        }
        __syncthreads();
    }

    // ---- final iteration: VN only (the CN pass after it is unobservable), then outputs -------
    {
        bool bad[NF];
#pragma unroll
        for (int v = 0; v < NF; v++) bad[v] = false;
        vn_phase(bad);
#pragma unroll
        for (int cc = 0; cc < CPT; cc++) {
            if (cvalid[cc]) {
                const int n = (g + cc * G) * Z + t;
                float S[NF];
                lds_ld<NF>(S, lds, swrite[cc]); // own value, just written
#pragma unroll
                for (int v = 0; v < NF; v++) {
                    if (!HIST) bad[v] = bad[v] || (n < a.length && S[v] < 0);
                    if (f0 + v < F) {
                        a.D[(size_t)n * F + f0 + v] = (S[v] < 0) ? 1 : 0;
                        if (a.app) a.app[(size_t)n * F + f0 + v] = S[v];
                    }
                }
            }
        }
        flags_publish(bad);
        __syncthreads();
        const int flag = flags_collect(a.max_iter);
        const int f = wg * FP * NF + tid;
        if (tid < FP * NF && f < F) {
            a.D[(size_t)a.L * Z * F + f] = flag;
            if (HIST && a.hist) a.hist[f] = hist;
        }
    }
}

// AND of all frames' flag histories -> first iteration at which every frame's flag is set.
__global__ __launch_bounds__(256) void k_hist_and(const unsigned long long *hist, int F, unsigned long long *out)
{
    unsigned long long x = ~0ull;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) x &= hist[f];
    for (int off = 32; off > 0; off >>= 1) x &= __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAnd(out, x);
}

// ---------------------------------------------------------------------------------------------
struct QcPlan {
    int J = 0, L = 0, Z = 0, nnz = 0, Wc = 0, Wv = 0;
    int NF = 0, FP = 0, G = 0, U = 0, RPT = 0, CPT = 0, threads = 0, variant = -1;
    int frames_per_wg = 0; // 0 = unavailable
    size_t lds_bytes = 0;
    QcCnEdge *d_cn = nullptr;
    unsigned short *d_rowptr = nullptr;
    QcVnEdge *d_vn = nullptr;
    unsigned char *d_wv = nullptr;
    char name[64] = "qc_lds(unavailable)";
};

using QcKernel = void (*)(QcArgs);
struct QcVariant { int NF, RPT, WC, CPT, WV, TPB; QcKernel fn, fn_hist; };

// Ahead-of-time instantiations.  A code runs on the first variant whose bounds
// cover it; codes outside all of them use the table kernels.
//                                   NF RPT WC CPT WV  TPB
#define QC_VARIANTS(X)             \
    X(2, 1, 20, 6, 4, 768)         /* J4_L24_Z96 (BASELINE config 2)      */ \
    X(2, 2, 7, 4, 3, 1024)         /* J32_L64_Z64 (BASELINE config 3)     */ \
    X(2, 2, 10, 6, 6, 768)         /* J8_L24_Z96                          */ \
    X(2, 3, 7, 6, 6, 768)          /* J12_L24_Z96                         */ \
    X(1, 2, 15, 8, 4, 768)         /* J6_L24_Z96 (one frame per lane)     */ \
    X(1, 1, 20, 6, 4, 1024)        /* J4_L24_Z256 (one frame per lane)    */

inline const QcVariant *qc_variants(int *count)
{
#define X(NF, RPT, WC, CPT, WV, TPB) {NF, RPT, WC, CPT, WV, TPB, k_qc<NF, RPT, WC, CPT, WV, TPB, false>, k_qc<NF, RPT, WC, CPT, WV, TPB, true>},
    static const QcVariant v[] = {QC_VARIANTS(X)};
#undef X
    *count = (int)(sizeof(v) / sizeof(v[0]));
    return v;
}

constexpr size_t kLdsBytes = 160 * 1024;

inline void qc_plan_release(QcPlan *q)
{
    if (q->d_cn) (void)hipFree(q->d_cn);
    if (q->d_rowptr) (void)hipFree(q->d_rowptr);
    if (q->d_vn) (void)hipFree(q->d_vn);
    if (q->d_wv) (void)hipFree(q->d_wv);
    q->d_cn = nullptr; q->d_rowptr = nullptr; q->d_vn = nullptr; q->d_wv = nullptr;
    q->frames_per_wg = 0;
}

static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// Choose (NF, FP, G, variant) for the code, upload its block lists.  Leaves
// frames_per_wg == 0 (not an error) when no variant / LDS budget fits.
inline int qc_plan_build(QcPlan *q, int J, int L, int Z, const int *H)
{
    q->J = J; q->L = L; q->Z = Z;
    std::vector<QcCnEdge> cn;
    std::vector<unsigned short> rowptr(J + 1, 0);
    std::vector<int> wv(L, 0);
    int Wc = 0;
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
    }
    const int nnz = (int)cn.size();
    const int Wv = *std::max_element(wv.begin(), wv.end());
    q->nnz = nnz; q->Wc = Wc; q->Wv = Wv;
    const int FPmin = 64 / gcd_i(Z, 64);
    int nvar = 0;
    const QcVariant *vars = qc_variants(&nvar);
    for (int NF = 2; NF >= 1 && q->frames_per_wg == 0; NF--) {
        const int FP = FPmin, U = FP * Z;
        if (U > 1024) continue;
        for (int vi = 0; vi < nvar && q->frames_per_wg == 0; vi++) {
            const QcVariant &v = vars[vi];
            if (v.NF != NF || v.WC < Wc || v.WV < Wv) continue;
            // LDS: R blocks (J rows padded to the variant's WC), S columns, the +0 / +inf slots, the flags
            const size_t slots = (size_t)(J * v.WC + L + 1) * U + 2; // + one trash block
            const size_t lds = (slots * NF + FP * NF) * sizeof(float);
            if (lds > kLdsBytes || slots > 65535) continue; // slot indices are packed into 16 bits
            const int G = std::min(v.TPB / U, std::max(J, 1));
            if (G < 1) continue;
            const int RPT = (J + G - 1) / G, CPT = (L + G - 1) / G;
            if (RPT > v.RPT || CPT > v.CPT) continue;
            q->NF = NF; q->FP = FP; q->U = U; q->G = G; q->RPT = RPT; q->CPT = CPT;
            q->threads = G * U; q->variant = vi; q->lds_bytes = lds; q->frames_per_wg = NF * FP;
        }
    }
    if (q->frames_per_wg == 0) return BLDPC_OK;
    const QcVariant &v = vars[q->variant];
    std::vector<QcVnEdge> vn((size_t)L * v.WV, QcVnEdge{0, 0});
    std::vector<int> fill(L, 0);
    for (int j = 0; j < J; j++)
        for (int e = rowptr[j]; e < rowptr[j + 1]; e++) {
            const int l = cn[e].col;
            // ascending j = the reference's edge order; .e = padded block index row*WC + position
            vn[(size_t)l * v.WV + fill[l]++] = {(unsigned short)(j * v.WC + (e - rowptr[j])), cn[e].shift};
        }
    std::vector<unsigned char> wvb(L);
    for (int l = 0; l < L; l++) wvb[l] = (unsigned char)wv[l];
    CLDPC_HIP(hipMalloc((void **)&q->d_cn, cn.size() * sizeof(QcCnEdge)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_rowptr, rowptr.size() * sizeof(unsigned short)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_vn, vn.size() * sizeof(QcVnEdge)), BLDPC_ENOMEM);
    CLDPC_HIP(hipMalloc((void **)&q->d_wv, wvb.size()), BLDPC_ENOMEM);
    CLDPC_HIP(hipMemcpy(q->d_cn, cn.data(), cn.size() * sizeof(QcCnEdge), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_rowptr, rowptr.data(), rowptr.size() * sizeof(unsigned short), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_vn, vn.data(), vn.size() * sizeof(QcVnEdge), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipMemcpy(q->d_wv, wvb.data(), wvb.size(), hipMemcpyHostToDevice), BLDPC_EHIP);
    CLDPC_HIP(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes), BLDPC_EHIP);
    CLDPC_HIP(hipFuncSetAttribute((const void *)v.fn_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes), BLDPC_EHIP);
    snprintf(q->name, sizeof(q->name), "qc_lds<nf%d,rpt%d,wc%d,cpt%d,wv%d>g%d_fp%d_t%d", v.NF, v.RPT, v.WC, v.CPT, v.WV, q->G, q->FP,
             q->threads);
    return BLDPC_OK;
}

inline int qc_launch(const QcPlan *q, const float *y, int F, int max_iter, int length, int *D, float *app,
                     unsigned long long *hist, hipStream_t st)
{
    int nvar = 0;
    const QcVariant &v = qc_variants(&nvar)[q->variant];
    QcArgs a;
    a.y = y; a.D = D; a.app = app; a.hist = hist;
    a.cn_edges = q->d_cn; a.rowptr = q->d_rowptr; a.vn_edges = q->d_vn; a.wv = q->d_wv;
    a.J = q->J; a.L = q->L; a.Z = q->Z; a.F = F; a.FP = q->FP; a.G = q->G; a.U = q->U;
    a.nWG = (F + q->frames_per_wg - 1) / q->frames_per_wg;
    a.max_iter = max_iter; a.length = length; a.nnz = q->nnz;
    const unsigned grid = (unsigned)((a.nWG + 7) / 8 * 8);
    hipLaunchKernelGGL(hist ? v.fn_hist : v.fn, dim3(grid), dim3(q->threads), q->lds_bytes, st, a);
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    return BLDPC_OK;
}

// hist_ws: device uint64[F] workspace; and_ws: device uint64; h_word: pinned host int[>=2].
inline int qc_decode(const QcPlan *q, const float *y, int F, int max_iter, int length, int exit_mode, int *D, float *app,
                     unsigned long long *flag_hist, unsigned long long *hist_ws, unsigned long long *and_ws, int *h_word,
                     int *itera, hipStream_t st)
{
    if (exit_mode == BLDPC_EXIT_FIXED) {
        *itera = max_iter;
        return qc_launch(q, y, F, max_iter, length, D, app, flag_hist, st);
    }
    // Reference rule (LDPC_Decoder.cu:150-153): stop after the first iteration at which ALL frames are
    // flagged.  Pass 1 runs max_iter iterations on-chip recording each frame's flag history; the AND of
    // the histories gives that iteration; if it is earlier than max_iter, pass 2 replays exactly that many.
    if (max_iter > 64) return fail(BLDPC_EUNSUPPORTED, "QC_LDS with BATCH_GLOBAL exit supports max_iter <= 64 (got %d)", max_iter);
    unsigned long long *hist = flag_hist ? flag_hist : hist_ws;
    int r = qc_launch(q, y, F, max_iter, length, D, app, hist, st);
    if (r) return r;
    const unsigned long long ones = ~0ull;
    CLDPC_HIP(hipMemcpyAsync(and_ws, &ones, sizeof(ones), hipMemcpyHostToDevice, st), BLDPC_EHIP);
    hipLaunchKernelGGL(k_hist_and, dim3(std::min((F + 255) / 256, 1024)), dim3(256), 0, st, hist, F, and_ws);
    unsigned long long all = 0;
    CLDPC_HIP(hipMemcpyAsync(&all, and_ws, sizeof(all), hipMemcpyDeviceToHost, st), BLDPC_EHIP);
    CLDPC_HIP(hipStreamSynchronize(st), BLDPC_EHIP);
    (void)h_word;
    if (max_iter < 64) all &= ((1ull << max_iter) - 1);
    int stop = max_iter;
    if (all) stop = __builtin_ctzll(all) + 1;
    *itera = stop;
    if (stop < max_iter) return qc_launch(q, y, F, stop, length, D, app, flag_hist, st);
    return BLDPC_OK;
}

} // namespace cldpc
