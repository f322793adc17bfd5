// nbldpc_hbm_kernel.hpp -- GF(q) EMS decoder for the codes the fused kernels do not take: message state of one frame larger than
// a CU's LDS, or check rows heavier than the templated walk unrolls (the reference's Tanner_74_9_Z128_GF16.txt: 9472 symbols,
// 1152 checks of weight 21; LDPC_N576_K480_GF256_exp.txt: GF(256), 12 checks of weight 12).  One workgroup = one frame at a time,
// as in k_nb_ems / k_nb_ems_wide, with the frame's state in a global-memory workspace slot (it stays in the L2 / MALL: a few MB
// per frame) instead of LDS, the graph tables read from global memory, and the check-node walk the reference's recursion
// (myNBLDPC/src/LDPC_Decoder.cpp:319-359) executed literally -- an explicit stack, any row weight, any (Nm, Nc) -- rather than a
// program unrolled per row weight.  Same arithmetic in the same order as the reference's Decoding_EMS (:172-317), so the same bits.
// Round 3: the walk no longer steps through the workspace.  A (row, edge) thread keeps its max array in LDS (the workgroup walks
// the check threads in chunks of as many as fit) and, for the two heavy-row shapes of the reference's matrix set -- GF(16) rows
// of weight 21 (Tanner_74_9_Z128_GF16.txt) and GF(256) rows of weight 12 (LDPC_N576_K480_GF256_exp.txt) -- runs the straight-line
// (and 20: the Tanner code has both) -- walk of the fused kernels (nb_t0 / nb_t1 / nb_conf, nbldpc_kernel.hpp) instantiated for that weight: the first two sorted pairs
// of every neighbour in registers, the others prefetched a batch ahead of the float chain, so that a step is a dependent addition
// (~8.5 cycles), not a dependent L2 load (~1 us).  Other row weights keep the explicit-stack walk, with the max array in LDS.
//
// Workspace slot (floats): v2c[NE][q] | pairs[NE][2q] (value, premultiplied symbol) | E[M*dc][q] (EMS_L_c2v) | LLR[N][q-1] | outs[N]
#pragma once
#include "nbldpc_kernel.hpp"

namespace cldpc {

constexpr int kNbHbmMaxDc = 64;
constexpr int kNbHbmThreads = 512; // 256 VGPRs per thread: the weight-21 walk keeps 100 values of its 20 neighbours in registers

// check threads walked per pass: their max arrays (q + 1 floats each, odd stride) share the LDS with the GF table
__host__ __device__ inline int nb_hbm_chunk(int q) { return (int)min((size_t)kNbHbmThreads, ((size_t)160 * 1024 - 256 - (size_t)q * q) / ((size_t)(q + 1) * 4)); }
__host__ __device__ inline size_t nb_hbm_lds_bytes(int q) { return (((size_t)q * q + 15) & ~(size_t)15) + (size_t)nb_hbm_chunk(q) * (q + 1) * 4; }

__host__ __device__ inline size_t nb_hbm_slot_floats(int N, int M, int q, int dv, int dc)
{
    const size_t NE = (size_t)N * dv, TC = (size_t)M * dc;
    return ((NE * q + NE * 2 * q + TC * q + (size_t)N * (q - 1) + N + 3) / 4) * 4;
}

// ConstructConf(Nm, Nc, begin = 0, except = e) of LDPC_Decoder.cpp:319-359 for one (row, edge): the depth-first walk over the
// row's other positions in ascending order, `sumNonLLR` carried through additions and subtractions exactly as the reference's
// by-reference recursion carries it (the drift is part of the result), `k` of every open level on an explicit stack.
// (symbols arrive premultiplied by 4, as byte offsets into the max array: the form the straight-line walk consumes)
__host__ __device__ inline void nb_hbm_conf(const float *pairs, const int *cn_src_row, int e, int nact, int q2, float *Et, int Nm, int Nc)
{
    unsigned short ks[kNbHbmMaxDc]; // k runs to Nm <= q = 256 inclusive: not a byte
    float s = 0.0f;
    int sym = 0, diff = 0, d = 0;
    if (nact == 0) { // a row of weight 1: the leaf is reached at once (:322-325)
        if (s > Et[0]) Et[0] = s;
        return;
    }
    ks[0] = 0;
    for (;;) {
        const int k = ks[d];
        if (k < Nm) {
            const int pos = d + (d >= e ? 1 : 0); // :327-331 skips `except`
            const float2 pr = *reinterpret_cast<const float2 *>(pairs + (size_t)cn_src_row[pos] * q2 + 2 * k);
            const int m = __builtin_bit_cast(int, pr.y), dk = (k != 0) ? 1 : 0;
            sym ^= m;         // :336
            s = s + pr.x;     // :337
            diff += dk;       // :338
            if (diff <= Nc) {
                if (d + 1 < nact) { // :341 one level down
                    d++;
                    ks[d] = 0;
                    continue;
                }
                if (s > Et[sym >> 2]) Et[sym >> 2] = s; // begin > end (:322-325)
                sym ^= m;                     // :342-344
                s = s - pr.x;
                diff -= dk;
                ks[d] = (unsigned short)(k + 1);
                continue;
            }
            sym ^= m; // :348-351, then break
            s = s - pr.x;
            diff -= dk;
        }
        // this level's loop is over (k == Nm, or the break): back in the parent, after its recursive call (:342-344)
        d--;
        if (d < 0) break;
        const int kp = ks[d];
        const int pos = d + (d >= e ? 1 : 0);
        const float2 pr = *reinterpret_cast<const float2 *>(pairs + (size_t)cn_src_row[pos] * q2 + 2 * kp);
        sym ^= __builtin_bit_cast(int, pr.y);
        s = s - pr.x;
        diff -= (kp != 0) ? 1 : 0;
        ks[d] = (unsigned short)(kp + 1);
    }
}

// The straight-line walk of the fused kernels for one (row, edge) thread of weight W over GF(Q): pairs in the workspace (global
// memory), max array at Ebase in LDS, already filled with -inf (EMS_L_c2v = -DBL_MAX, :277-280: the ZS form of nb_t0, which takes
// every leaf as a maximum and so also serves the codes with a zero coefficient).
template <int W, int Q> __device__ void nb_cn_update_hbm(const NbArgs &a, const int *src, const float *pairs, int pst, char *Ebase, int e)
{
    constexpr int NACT = W - 1;
    NbCn<NACT> c;
    c.E = Ebase;
    c.pairs = pairs;
#pragma unroll
    for (int i = 0; i < NACT; i++) {
        const int pos = i + (i >= e ? 1 : 0); // ascending positions, skipping `except` (:327-331)
        c.pb[i] = src[pos] * pst;
        const float2 p0 = *reinterpret_cast<const float2 *>(pairs + c.pb[i]);
        const float2 p1 = *reinterpret_cast<const float2 *>(pairs + c.pb[i] + 2);
        c.v0[i] = p0.x; c.m0[i] = __float_as_int(p0.y);
        c.v1[i] = p1.x; c.m1[i] = __float_as_int(p1.y);
    }
    c.s = 0.0f;
    nb_t0<0, NACT, Q, true>(c, 0); // ConstructConf(GFQ, 1) :286
    c.s = 0.0f;
    int Nc = a.Nc;
    if (a.Nc == a.dcmax_cfg - 1) Nc = W - 1; // :294-297
    nb_conf<0, NACT>(c, 0, 0, a.Nm, Nc);     // ConstructConf(EMS_Nm, EMS_Nc) :300
}

__global__ __launch_bounds__(kNbHbmThreads) void k_nb_ems_hbm(NbArgs a)
{
    constexpr int NT = kNbHbmThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char mulb[]; // [q][q], then the max arrays of the check threads of one pass
    __shared__ int flag;
    const int tid = threadIdx.x;
    const int N = a.N, M = a.M, q = a.q, dv = a.dv, dc = a.dc;
    const int NE = N * dv, TC = M * dc, q2 = 2 * q;
    float *v2c = a.ws + (size_t)blockIdx.x * a.ws_stride; // [NE][q]
    float *pairs = v2c + (size_t)NE * q;                  // [NE][2q]
    float *E = pairs + (size_t)NE * q2;                   // [TC][q]
    float *LLRw = E + (size_t)TC * q;                     // [N][q-1]
    int *outs = reinterpret_cast<int *>(LLRw + (size_t)N * (q - 1)); // [N]
    for (int i = tid; i < q * q; i += NT) mulb[i] = a.mul[i];
    const int QP = q + 1, chunk = nb_hbm_chunk(q);
    float *Elds = reinterpret_cast<float *>(mulb + ((q * q + 15) & ~15)); // [chunk][q + 1]

    __shared__ int next_frame;
    for (int frame = blockIdx.x;; frame += gridDim.x) { // a.work: frames from a counter (they differ 20x in iterations), else strided
        if (a.work) {
            if (tid == 0) next_frame = atomicAdd(a.work, 1);
            __syncthreads();
            frame = __builtin_amdgcn_readfirstlane(next_frame);
        }
        if (frame >= a.B) break;
        const float *Lch = a.Lch + (size_t)frame * N * (q - 1);
        float *LLRo = a.LLR ? a.LLR + (size_t)frame * N * (q - 1) : nullptr;
        for (int i = tid; i < TC * q; i += NT) E[i] = 0.0f; // L_c2v = 0 (:185-193): (0 - 0) / 1.2 == +0
        if (tid == 0) flag = 0;
        __syncthreads();
        int it = 0, ok = 0;
        while (it < a.max_iter) {
            it++;
            // ---- A: variable nodes (:202-214, :241-251), one thread per (node, vector position) ---------------------
            for (int idx = tid; idx < N * q; idx += NT) {
                const int col = idx / q, el = idx - col * q;
                const bool active = el < q - 1; // position q - 1 = element 0 (:250)
                const int sym = active ? el + 1 : 0, w = a.vn_w[col];
                float llr = active ? Lch[col * (q - 1) + el] : 0.0f;
                float c2[kNbMaxDv];
#pragma unroll
                for (int d = 0; d < kNbMaxDv; d++) {
                    c2[d] = 0.0f;
                    if (d < w) {
                        const int thr = a.vn_thr[col * dv + d], h = a.vn_gf[col * dv + d];
                        const float c = nb_div12(E[(size_t)thr * q + mulb[sym * q + h]] - E[(size_t)thr * q]); // :309 (double division)
                        c2[d] = c;
                        llr = llr + c; // :208-213, ascending d
                    }
                }
                if (active) {
                    LLRw[col * (q - 1) + el] = llr;
                    if (LLRo) LLRo[col * (q - 1) + el] = llr;
                }
#pragma unroll
                for (int d = 0; d < kNbMaxDv; d++)
                    if (d < w) v2c[(size_t)(col * dv + d) * q + el] = active ? llr - c2[d] : 0.0f;
            }
            __syncthreads();
            // DecideLLRVector (:71-91): running maximum from 0, strict >: the first maximum wins
            for (int col = tid; col < N; col += NT) {
                float mx = 0.0f;
                int dec = 0;
                for (int k = 0; k < q - 1; k++) {
                    const float v = LLRw[col * (q - 1) + k];
                    if (v > mx) { mx = v; dec = k + 1; }
                }
                outs[col] = dec;
            }
            __syncthreads();
            // ---- S: syndrome (:218-238) ------------------------------------------------------------------------------
            for (int row = tid; row < M; row += NT) {
                int s = 0;
                for (int i = 0; i < a.cn_w[row]; i++) s ^= mulb[outs[a.cn_vn[row * dc + i]] * q + a.cn_gf[row * dc + i]];
                if (s) flag = 1;
            }
            __syncthreads();
            if (flag == 0) {
                it--; // :236
                ok = 1;
                break;
            }
            // ---- B: stable descending sort of every v2c vector (:17-36, :253-269) as a rank count: one thread per (edge, position)
            for (int idx = tid; idx < NE * q; idx += NT) {
                const int edge = idx / q, k = idx - edge * q;
                if ((edge % dv) >= a.vn_w[edge / dv]) continue; // slot d of a column is an edge when d < its weight
                const float *vv = v2c + (size_t)edge * q;
                const float val = vv[k];
                int rank = 0;
                for (int j = 0; j < q; j += 4) { // the bubble sort swaps on a strict <: equal values (-0 == +0 too) keep their order
                    const float4 o = *reinterpret_cast<const float4 *>(vv + j);
                    rank += (o.x > val || (o.x == val && j + 0 < k)) ? 1 : 0;
                    rank += (o.y > val || (o.y == val && j + 1 < k)) ? 1 : 0;
                    rank += (o.z > val || (o.z == val && j + 2 < k)) ? 1 : 0;
                    rank += (o.w > val || (o.w == val && j + 3 < k)) ? 1 : 0;
                }
                const int sym = (k < q - 1) ? k + 1 : 0;
                float2 pr;
                pr.x = val;
                pr.y = __int_as_float((int)mulb[sym * q + a.vn_gf[edge]] << 2); // GFMultiply(sort_Entr_v2c, linkVNs_GF) of :336, as a byte offset
                *reinterpret_cast<float2 *>(pairs + (size_t)edge * q2 + 2 * rank) = pr;
            }
            __syncthreads();
            if (tid == 0) flag = 0; // read by everybody before the barrier above, written again after the next ones
            // ---- C: check nodes (:272-313), one thread per (row, edge), `chunk` of them per pass (max arrays in LDS) ------------
            for (int base = 0; base < TC; base += chunk) {
                const int thr = base + tid;
                if (tid < chunk && thr < TC) {
                    const int row = thr / dc, e = thr - row * dc, w = a.cn_w[row];
                    if (e < w) {
                        float *Et = Elds + tid * QP;
                        for (int k = 0; k < q; k++) Et[k] = -__builtin_inff(); // EMS_L_c2v = -DBL_MAX as a float (:277-280)
                        const int *src = a.cn_src + row * dc;
                        if (q == 16 && w == 21) nb_cn_update_hbm<21, 16>(a, src, pairs, q2, reinterpret_cast<char *>(Et), e);
                        else if (q == 16 && w == 20) nb_cn_update_hbm<20, 16>(a, src, pairs, q2, reinterpret_cast<char *>(Et), e); // (the Tanner code has both)
                        else if (q == 256 && w == 12) nb_cn_update_hbm<12, 256>(a, src, pairs, q2, reinterpret_cast<char *>(Et), e);
                        else {
                            int Nc = a.Nc;
                            if (a.Nc == a.dcmax_cfg - 1) Nc = w - 1; // :294-297
                            nb_hbm_conf(pairs, src, e, w - 1, q2, Et, q, 1);     // conf(q, 1)  :286
                            nb_hbm_conf(pairs, src, e, w - 1, q2, Et, a.Nm, Nc); // conf(Nm, Nc) :294-300
                        }
                        float *Eg = E + (size_t)thr * q; // what phase A of the next iteration (and the L_c2v output) reads
                        for (int k = 0; k < q; k++) Eg[k] = Et[k];
                    }
                }
            }
            __syncthreads();
        }
        // ---- outputs ---------------------------------------------------------------------------------------------------
        for (int i = tid; i < N; i += NT) a.out[(size_t)frame * N + i] = outs[i];
        if (tid == 0) {
            a.iters[frame] = it;
            a.ok[frame] = ok;
        }
        if (a.c2v) {
            for (int idx = tid; idx < TC * (q - 1); idx += NT) { // L_c2v[dc][k-1] = (E[k*h] - E[0]) / 1.2 (:305-310)
                const int thr = idx / (q - 1), k = idx - thr * (q - 1) + 1;
                const int row = thr / dc, e = thr - row * dc;
                float o = 0.0f;
                if (e < a.cn_w[row]) o = nb_div12(E[(size_t)thr * q + mulb[k * q + a.cn_gf[thr]]] - E[(size_t)thr * q]);
                a.c2v[((size_t)frame * TC + thr) * (q - 1) + k - 1] = o;
            }
        }
        __syncthreads(); // the slot is reused by this workgroup's next frame
    }
}

} // namespace cldpc
