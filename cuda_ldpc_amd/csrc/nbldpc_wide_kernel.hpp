// nbldpc_wide_kernel.hpp -- fused GF(q) EMS decoder for fields larger than a wavefront (q = 128, 256): one workgroup = one
// frame, a message vector spans q / 64 waves.  Same arithmetic, in the same order, as the reference's CPU decoder
// (myNBLDPC/src/LDPC_Decoder.cpp:172-359) and as k_nb_ems (nbldpc_kernel.hpp), whose check-node walk (nb_cn_update: the
// depth-first configuration walk with its float drift) is reused as is; what changes is everything that k_nb_ems does inside
// one wave: the hard decision (maximum and FIRST maximum over q - 1 values in q / 64 waves) and the stable sort (q keys).
// First path for the reference's GF(256) code LDPC_N96_K48_GF256_d1_exp.txt (12 symbols, 6 checks, dv 2, dc 4: 139 KB of LDS, of
// which 64 KB are the multiplication table): written for exactness and simplicity -- the sort is a rank count over the q keys.
//
// Threads: NT = 1024 = NG groups of q threads; a group handles one variable node (phase A) or one edge (phase B) at a time, thread
// el of a group <-> vector position el (field element el + 1, position q - 1 = element 0, LDPC_Decoder.cpp:250).
#pragma once
#include "nbldpc_kernel.hpp"

namespace cldpc {

__host__ __device__ constexpr size_t nb_wide_lds_bytes(int N, int M, int q, int dv, int dc, int NT)
{
    return ((size_t)N * dv * (2 * q + 2) + (size_t)(q + 1) * M * dc + N + 4 + 4 * (NT / 64)) * sizeof(float) + (size_t)(NT / q) * q * 8 +
           (size_t)q * q + ((size_t)N + 2 * (size_t)N * dv + (size_t)M + 3 * (size_t)M * dc + 8) * sizeof(unsigned short) + 32;
}

template <int Q, int NT> __global__ __launch_bounds__(NT) void k_nb_ems_wide(NbArgs a)
{
    static_assert(Q % 64 == 0 && NT % Q == 0, "a vector is a whole number of waves, the workgroup a whole number of vectors");
    constexpr int GW = Q / 64, NG = NT / Q, QP = Q + 1, q = Q;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int frame = blockIdx.x; // (a persistent workgroup takes its frames from a.work instead, see k_nb_ems)
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = tid / Q, el = tid - grp * Q, wv = el >> 6; // wave of this thread inside its group
    const int N = a.N, M = a.M, dv = a.dv, dc = a.dc;
    const int NE = N * dv, TC = M * dc, PST = nb_pair_stride(q);
    float *pairs = lds;                                 // [NE][PST] (value, premultiplied symbol) pairs, sorted by phase B
    float *E = pairs + NE * PST;                        // [TC][q + 1] max arrays (EMS_L_c2v)
    int *outs = reinterpret_cast<int *>(E + TC * QP);   // [N]
    int *flag = outs + N;                               // [4]
    float *redv = reinterpret_cast<float *>(flag + 4);  // [NT/64] wave maxima
    int *redi = reinterpret_cast<int *>(redv + NT / 64); // [NT/64] position of the first maximum inside the wave
    unsigned long long *keys = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(redi + 2 * (NT / 64)) + 15) & ~(uintptr_t)15); // [NG][q]
    unsigned char *mulb = reinterpret_cast<unsigned char *>(keys + NG * q); // [q][q]
    unsigned short *t_vn_w = reinterpret_cast<unsigned short *>(mulb + q * q);
    unsigned short *t_vn_thr = t_vn_w + N, *t_vn_gf = t_vn_thr + NE, *t_cn_w = t_vn_gf + NE, *t_cn_src = t_cn_w + M;
    unsigned short *t_cn_gf = t_cn_src + TC, *t_cn_vn = t_cn_gf + TC;
    for (int i = tid; i < N; i += NT) t_vn_w[i] = (unsigned short)a.vn_w[i];
    for (int i = tid; i < NE; i += NT) { t_vn_thr[i] = (unsigned short)a.vn_thr[i]; t_vn_gf[i] = (unsigned short)a.vn_gf[i]; }
    for (int i = tid; i < M; i += NT) t_cn_w[i] = (unsigned short)a.cn_w[i];
    for (int i = tid; i < TC; i += NT) {
        t_cn_src[i] = (unsigned short)a.cn_src[i]; t_cn_gf[i] = (unsigned short)a.cn_gf[i]; t_cn_vn[i] = (unsigned short)a.cn_vn[i];
    }
    for (int i = tid; i < q * q; i += NT) mulb[i] = a.mul[i];
    __syncthreads();
    const float *Lch = nullptr; // of the frame being decoded
    float *LLRo = nullptr;
    const bool active = el < q - 1;
    const int sym = active ? el + 1 : 0;
    const int rounds_a = (N + NG - 1) / NG, rounds_b = (NE + NG - 1) / NG;
    int it = 0, ok = 0;
    for (;;) { // frames of this workgroup: its own, or (persistent) the next of the batch, see k_nb_ems
    if (a.work) {
        if (tid == 0) flag[1] = atomicAdd(a.work, 1);
        __syncthreads();
        frame = __builtin_amdgcn_readfirstlane(flag[1]);
    }
    if (frame >= a.B) break;
    Lch = a.Lch + (size_t)frame * N * (q - 1);
    LLRo = a.LLR ? a.LLR + (size_t)frame * N * (q - 1) : nullptr;
    for (int i = tid; i < TC * QP; i += NT) E[i] = 0.0f; // L_c2v = 0 (:185-193): (0-0)/1.2 == +0
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    it = 0;
    ok = 0;
    while (it < a.max_iter) {
        it++;
        // ---- A: variable nodes (:202-251), one per group and round ---------------------------------------------------
        for (int r = 0; r < rounds_a; r++) {
            const int col = r * NG + grp;
            const bool on_col = col < N;
            const int cc = on_col ? col : N - 1;
            const int w = t_vn_w[cc];
            float llr = active ? Lch[cc * (q - 1) + el] : 0.0f;
            float c2[kNbMaxDv];
#pragma unroll
            for (int d = 0; d < kNbMaxDv; d++) {
                c2[d] = 0.0f;
                if (d < w) {
                    const int thr = t_vn_thr[cc * dv + d], h = t_vn_gf[cc * dv + d];
                    const float c = nb_div12(E[thr * QP + mulb[sym * q + h]] - E[thr * QP]); // :309, double division
                    c2[d] = c;
                    llr = llr + c; // :208-213, ascending d
                }
            }
            // DecideLLRVector (:71-91): running max from 0, strict >, first maximum wins -- per wave, then over the group's waves
            const float v = active ? llr : -__builtin_inff();
            const float wmx = nb_wave_max(v);
            const unsigned long long eq = __ballot(v == wmx);
            if (lane == 0) { redv[tid >> 6] = wmx; redi[tid >> 6] = (wv << 6) + (int)__builtin_ctzll(eq); }
            __syncthreads();
            if (on_col) {
                if (el == 0) {
                    float mx = redv[grp * GW];
                    int pos = redi[grp * GW];
                    for (int g2 = 1; g2 < GW; g2++)
                        if (redv[grp * GW + g2] > mx) { mx = redv[grp * GW + g2]; pos = redi[grp * GW + g2]; } // strict: the earlier wave keeps ties
                    outs[col] = (mx > 0.0f) ? pos + 1 : 0;
                }
                if (LLRo && active) LLRo[col * (q - 1) + el] = llr;
#pragma unroll
                for (int d = 0; d < kNbMaxDv; d++)
                    if (d < w) pairs[(col * dv + d) * PST + 2 * el] = active ? llr - c2[d] : 0.0f; // :241-251
            }
            __syncthreads(); // redv / redi are reused by the next round
        }
        // ---- S: syndrome (:218-238) ----------------------------------------------------------
        if (tid < M) {
            int s = 0;
            for (int i = 0; i < t_cn_w[tid]; i++) s ^= mulb[outs[t_cn_vn[tid * dc + i]] * q + t_cn_gf[tid * dc + i]];
            if (s) flag[0] = 1;
        }
        __syncthreads();
        if (flag[0] == 0) {
            it--; // :236
            ok = 1;
            break;
        }
        // ---- B: stable descending sort of every v2c vector (:17-36, :253-269): rank = number of larger keys, the keys
        // (order-preserving image of the value, q - 1 - position) being pairwise distinct --------------------------------
        for (int r = 0; r < rounds_b; r++) {
            const int edge = r * NG + grp;
            const int ee = edge < NE ? edge : NE - 1;
            const bool on_edge = edge < NE && (ee % dv) < t_vn_w[ee / dv]; // slot d of a column is an edge when d < its weight
            const float val = pairs[ee * PST + 2 * el];
            const uint32_t b = __float_as_uint(val + 0.0f); // +0.0f folds -0 onto +0 (they compare equal)
            const unsigned long long key = ((unsigned long long)(b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) << 32) | (unsigned)(q - 1 - el);
            keys[grp * q + el] = key;
            __syncthreads();
            int rank = 0;
            const unsigned long long *kg = keys + grp * q;
            for (int j = 0; j < q; j += 2) { // both keys of a 16-byte word are the same for every lane: broadcast reads
                const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(kg + j);
                rank += (kk.x > key) ? 1 : 0;
                rank += (kk.y > key) ? 1 : 0;
            }
            if (on_edge) {
                float2 pr;
                pr.x = val;
                pr.y = __int_as_float((int)mulb[sym * q + t_vn_gf[ee]] << 2); // GFMultiply(sort_Entr_v2c, linkVNs_GF) of :334, as a byte offset
                *reinterpret_cast<float2 *>(pairs + ee * PST + 2 * rank) = pr;
            }
            __syncthreads(); // everybody has read its own value and the keys before anything is overwritten / the keys are reused
        }
        if (a.zero_coeff) { // EMS_L_c2v = -DBL_MAX (:277-280), see nb_t0; nobody reads E between phases A and C (workgroup-uniform branch)
            for (int i = tid; i < TC * QP; i += NT) E[i] = -__builtin_inff();
            __syncthreads();
        }
        // ---- C: check nodes (:272-303), one thread per (row, edge): the walk of nbldpc_kernel.hpp ---------------------
        if (tid < TC) {
            const int row = tid / dc, e = tid - row * dc, w = t_cn_w[row];
            if (e < w) {
                switch (w) {
                case 2: nb_cn_update<2, Q>(a, t_cn_src, pairs, E, QP, row, e, tid); break;
                case 3: nb_cn_update<3, Q>(a, t_cn_src, pairs, E, QP, row, e, tid); break;
                case 4: nb_cn_update<4, Q>(a, t_cn_src, pairs, E, QP, row, e, tid); break;
                case 5: nb_cn_update<5, Q>(a, t_cn_src, pairs, E, QP, row, e, tid); break;
                case 6: nb_cn_update<6, Q>(a, t_cn_src, pairs, E, QP, row, e, tid); break;
                default: break;
                }
            }
        }
        __syncthreads();
        if (tid == 0) flag[0] = 0; // next write is two barriers away, last read was two barriers ago
    }
    // ---- outputs ------------------------------------------------------------------------------
    for (int i = tid; i < N; i += NT) a.out[(size_t)frame * N + i] = outs[i];
    if (tid == 0) {
        a.iters[frame] = it;
        a.ok[frame] = ok;
    }
    if (a.c2v && tid < TC) {
        const int row = tid / dc, e = tid - row * dc;
        float *o = a.c2v + ((size_t)frame * TC + tid) * (q - 1);
        if (e < t_cn_w[row]) {
            const int h = t_cn_gf[tid];
            const float e0 = E[tid * QP];
            for (int k = 1; k < q; k++) o[k - 1] = nb_div12(E[tid * QP + mulb[k * q + h]] - e0);
        } else {
            for (int k = 1; k < q; k++) o[k - 1] = 0.0f;
        }
    }
    if (!a.work) break;
    __syncthreads(); // the max arrays, the symbols and flag[1] are reused by the next frame
    } // next frame
}

} // namespace cldpc
