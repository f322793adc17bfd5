// nbldpc_tmm_kernel.hpp -- fused GF(q) trellis min-max decoders, one workgroup = one frame, all iterations on-chip.
//
// Bit-exact restatement of the reference's CPU functions Decoding_TMM (decoder_method 1,
// myNBLDPC/src/LDPC_Decoder.cpp:361-558) and Decoding_layered_TMM (decoder_method 3, :560-702) with their helpers
// d_TMM_Get_Zn (:704-724), d_TMM_Get_deltaU (:726-744), TMM_Get_Min (:746-772), TMM_ConstructConf (:774-817) and
// d_DecideLLRVector (:92-105).  Vectors have q entries (field element 0 included), smaller = more likely; all
// arithmetic is float subtract / compare plus one double multiplication by 0.8 (:527).
//
// Lane <-> field element.  One wave owns a variable node (sum, first-minimum decision) or a check row:
//   Zn[d]   : first minimum of the incoming vector (wave minimum by DPP, lowest lane among the equal ones) times h
//   deltaU  : lane eta gathers v2c[d][h^-1 (eta ^ Zn[d])] - min
//   Min1/Min2/Col over the row's edges, sequentially with the reference's if / else-if
//   ConstructConf: lane i walks j = 0..q-1 in order (the update is a strict <, so the first best path wins);
//           (Min1[j], Col[j]) is a broadcast LDS read, (Min1[i^j], Col[i^j]) a permuted one
//   output  : lane eta scatters (float)((double)Lc2p[eta] * 0.8) to c2v[d][h^-1 (eta ^ syndrome ^ Zn[d])]
// Flooding (method 1): LLR accumulates the c2v of every iteration, as the reference's does (:425-435, it is never
// reset to the channel).  Layered (method 3): the reference visits the rows one after the other; rows that share no
// variable node commute, so the host orders them into dependency levels (a row's level = 1 + the highest level among
// the earlier rows it shares a variable with) and the waves of a workgroup run one level at a time -- same values,
// same order per variable.
// LDS: LLR [N][q], c2v [M*dc][q], (flooding) v2c [N*dv][q], per-wave (Min1, Col) pairs, GF table, graph tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nbldpc_kernel.hpp"

#ifndef TMM_ABLATE
#define TMM_ABLATE 0 // experiments only (timing, wrong results): 1 no ConstructConf walk, 2 no check rows at all
#endif

namespace cldpc {

constexpr int kTmmThreads = 1024;
constexpr int kTmmMaxW = 8; // row weights handled (registers per lane)

struct TmmArgs {
    const float *Lch; // [B][N][q-1]
    int *out;         // [B][N]
    int *iters;       // [B]
    int *ok;          // [B]
    float *LLR;       // [B][N][q] or nullptr
    float *c2v;       // [B][M][dc][q] or nullptr
    const int *vn_w, *vn_thr;                 // [N], [N][dv]
    const int *cn_w, *cn_src, *cn_gf, *cn_vn; // [M], [M][dc] x3
    const int *cn_hinv;                       // [M][dc] inverse of the edge coefficient
    const int *row_order, *level_begin;       // layered: rows sorted by level [M], first row of each level [levels+1]
    const unsigned char *mul;                 // [q][q]
    int N, M, q, dv, dc, B, max_iter, levels;
    int *work = nullptr; // != nullptr: persistent workgroups take frame after frame from this counter (zeroed by the host), see k_nb_ems
};

template <int CTRL, int ROW_MASK> __device__ __forceinline__ float nb_dpp_min(float v)
{
    const int x = __builtin_bit_cast(int, v);
    return fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false)));
}
// minimum over the 64 lanes, in every lane (see nb_wave_max)
__device__ __forceinline__ float nb_wave_min(float v)
{
    v = nb_dpp_min<0xB1, 0xf>(v);
    v = nb_dpp_min<0x4E, 0xf>(v);
    v = nb_dpp_min<0x141, 0xf>(v);
    v = nb_dpp_min<0x140, 0xf>(v);
    v = nb_dpp_min<0x142, 0xa>(v);
    v = nb_dpp_min<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__host__ __device__ inline size_t tmm_lds_bytes(int N, int M, int q, int dv, int dc, bool layered)
{
    size_t fl = (((size_t)N * q + (size_t)M * dc * q + (layered ? 0 : (size_t)N * dv * q) + 127) & ~(size_t)127) + (size_t)(kTmmThreads / 64) * 64 * 2 + N + 4;
    size_t tb = (size_t)N + (size_t)N * dv + (size_t)M + 4 * (size_t)M * dc + (size_t)M + 64 + 2;
    return fl * sizeof(float) + (size_t)q * q + tb * sizeof(unsigned short);
}

template <int Q, bool LAYERED> __global__ __launch_bounds__(kTmmThreads) void k_nb_tmm(TmmArgs a)
{
    constexpr int q = Q, NT = kTmmThreads, nwaves = NT / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int frame = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, M = a.M, dv = a.dv, dc = a.dc, NE = N * dv, TC = M * dc;
    float *LLR = lds;                               // [N][q]
    float *C2V = LLR + N * q;                       // [TC][q]
    float *V2C = C2V + TC * q;                      // [NE][q] (flooding only)
    float *PR = lds + (((LLR - lds) + N * q + TC * q + (LAYERED ? 0 : NE * q) + 127) & ~127); // [nwaves][64][2] (Min1, Col) pairs, 512-byte aligned
    int *outs = reinterpret_cast<int *>(PR + nwaves * 128); // [N]
    int *flag = outs + N;                           // [4]
    unsigned char *mulb = reinterpret_cast<unsigned char *>(flag + 4); // [q][q]
    unsigned short *t_vn_w = reinterpret_cast<unsigned short *>(mulb + q * q);
    unsigned short *t_vn_thr = t_vn_w + N;          // [N][dv] check thread row*dc+slot of each VN edge
    unsigned short *t_cn_w = t_vn_thr + NE;         // [M]
    unsigned short *t_cn_src = t_cn_w + M;          // [M][dc] VN edge vn*dv+idx of each CN slot
    unsigned short *t_cn_gf = t_cn_src + TC;
    unsigned short *t_cn_vn = t_cn_gf + TC;
    unsigned short *t_cn_hinv = t_cn_vn + TC;
    unsigned short *t_rows = t_cn_hinv + TC;        // [M] layered: rows by level
    unsigned short *t_lvl = t_rows + M;             // [levels+1] (levels <= 63)
    for (int i = tid; i < N; i += NT) t_vn_w[i] = (unsigned short)a.vn_w[i];
    for (int i = tid; i < NE; i += NT) t_vn_thr[i] = (unsigned short)a.vn_thr[i];
    for (int i = tid; i < M; i += NT) { t_cn_w[i] = (unsigned short)a.cn_w[i]; t_rows[i] = (unsigned short)(LAYERED ? a.row_order[i] : i); }
    for (int i = tid; i < TC; i += NT) {
        t_cn_src[i] = (unsigned short)a.cn_src[i]; t_cn_gf[i] = (unsigned short)a.cn_gf[i];
        t_cn_vn[i] = (unsigned short)a.cn_vn[i]; t_cn_hinv[i] = (unsigned short)a.cn_hinv[i];
    }
    if (LAYERED)
        for (int i = tid; i <= a.levels; i += NT) t_lvl[i] = (unsigned short)a.level_begin[i];
    for (int i = tid; i < q * q; i += NT) mulb[i] = a.mul[i];
    const bool act = lane < q;
    __syncthreads();

    auto uni = [&](const unsigned short *t, int i) -> int { return __builtin_amdgcn_readfirstlane((int)t[i]); }; // wave-uniform table entry
    // first minimum of a vector held one element per lane: (value, lane)
    auto first_min = [&](float v, float &m) -> int {
        m = nb_wave_min(act ? v : __builtin_inff());
        const unsigned long long eq = __ballot(act && v == m);
        return (int)__builtin_ctzll(eq | (1ull << 63));
    };

    // One check row (:489-531 / :637-680).  LAYERED: also forms v2c = LLR - c2v on the way in and LLR = v2c + c2v on the way out.
    auto do_row = [&](int row) {
        const int w = uni(t_cn_w, row);
        float vq[kTmmMaxW], dU[kTmmMaxW];
        int Zn[kTmmMaxW];
        int syn = 0;
        float *pr = PR + wave * 128;
#pragma unroll
        for (int d = 0; d < kTmmMaxW; d++)
            if (d < w) { // wave-uniform
                const int thr = row * dc + d;
                const int vbase = (LAYERED ? uni(t_cn_vn, thr) : uni(t_cn_src, thr)) * q, hinv = uni(t_cn_hinv, thr);
                if (LAYERED) vq[d] = act ? LLR[vbase + lane] - C2V[thr * q + lane] : 0.0f; // :640-646
                else vq[d] = act ? V2C[vbase + lane] : 0.0f;
                float mn;
                const int qs = first_min(vq[d], mn);                        // d_TMM_Get_Zn
                Zn[d] = __builtin_amdgcn_readfirstlane((int)mulb[qs * q + uni(t_cn_gf, thr)]);
                syn ^= Zn[d];
                // d_TMM_Get_deltaU: lane eta <- v2c[h^-1 (eta ^ Zn)] - min
                const int idx = mulb[hinv * q + ((lane ^ Zn[d]) & (q - 1))];
                float g;
                if (LAYERED) g = LLR[vbase + idx] - C2V[thr * q + idx];
                else g = V2C[vbase + idx];
                dU[d] = g - mn;
            }
        // TMM_Get_Min
        float M1 = __builtin_inff(), M2 = __builtin_inff();
        int Col = 0;
#pragma unroll
        for (int d = 0; d < kTmmMaxW; d++)
            if (d < w) {
                const bool lt1 = dU[d] < M1, lt2 = dU[d] < M2;
                M2 = lt1 ? M1 : (lt2 ? dU[d] : M2);
                Col = lt1 ? d : Col;
                M1 = lt1 ? dU[d] : M1;
            }
        // TMM_ConstructConf
        if (act) {
            pr[lane * 2] = M1;
            pr[lane * 2 + 1] = __int_as_float(Col);
        }
        // The reference's two branches (:797-810) take the larger of (Min1[j], Min1[k]) when the two differ, their columns
        // differ and it beats I.  Per step and lane: one permuted read, the candidate, three compares whose masks meet
        // on the scalar unit, two selects; which step won last is all that has to be remembered (path and E follow).
        // deltaU = v - min >= +0 and never -0 or NaN for finite inputs, so the walk compares the BIT PATTERNS as unsigned
        // integers (same order, no canonicalisation instructions); the (Min1, Col) array of a wave is 512-byte aligned, so the
        // permuted address is one XOR with a literal.
        unsigned Ib = __float_as_uint(M1);
        int jb = -1;
        unsigned long long upd = 0; // lanes whose I was replaced at least once
        const unsigned prb = (unsigned)(uintptr_t)pr; // LDS byte address of this wave's pairs (low 9 bits zero)
        const unsigned xaddr = prb + (unsigned)lane * 8u;
        typedef unsigned tmm_u2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const tmm_u2 lds_u2;
        constexpr int JB = 8; // steps whose permuted reads are in flight together
#pragma unroll
        for (int j0 = 0; j0 < ((TMM_ABLATE & 1) ? 0 : q); j0 += JB) {
            tmm_u2 pjv[JB], pkv[JB];
#pragma unroll
            for (int u = 0; u < JB; u++) {
                pjv[u] = *reinterpret_cast<lds_u2 *>(prb + 8u * (j0 + u));
                pkv[u] = *reinterpret_cast<lds_u2 *>(xaddr ^ (8u * (j0 + u)));
            }
#pragma unroll
            for (int u = 0; u < JB; u++) {
                const int j = j0 + u;
                const unsigned c = max(pjv[u].x, pkv[u].x);
                const unsigned long long t = __builtin_amdgcn_ballot_w64(pjv[u].y != pkv[u].y) & __builtin_amdgcn_ballot_w64(pjv[u].x != pkv[u].x) &
                                             __builtin_amdgcn_ballot_w64(c < Ib) & ~(1ull << j);
                const bool tk = __builtin_amdgcn_inverse_ballot_w64(t);
                Ib = tk ? c : Ib;
                jb = tk ? j : jb;
                upd |= t;
                asm volatile("" : "+v"(jb)); // select now: otherwise all 64 masks are kept (and spilled) for a later chain
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float I = __uint_as_float(Ib);
        const bool any = __builtin_amdgcn_inverse_ballot_w64(upd);
        const float E = any ? M1 : M2;
        int P0 = Col, P1 = Col;
        if (upd) { // wave-uniform
            const int jbc = max(jb, 0);
            const int c0 = __float_as_int(pr[2 * jbc + 1]), c1 = __float_as_int(pr[2 * ((lane ^ jbc) & (q - 1)) + 1]);
            P0 = any ? c0 : Col;
            P1 = any ? c1 : Col;
        }
        // outputs (:506-530)
#pragma unroll
        for (int d = 0; d < kTmmMaxW; d++)
            if (d < w) {
                const int thr = row * dc + d;
                const float L = (lane == 0) ? 0.0f : ((d != P0 && d != P1) ? I : E);
                const float val = (float)((double)L * 0.8);
                const int beta = mulb[uni(t_cn_hinv, thr) * q + ((lane ^ syn ^ Zn[d]) & (q - 1))];
                if (act) C2V[thr * q + beta] = val;
            }
        if (LAYERED) { // :682-688
#pragma unroll
            for (int d = 0; d < kTmmMaxW; d++)
                if (d < w) {
                    const int thr = row * dc + d;
                    const int vb = uni(t_cn_vn, thr) * q;
                    if (act) LLR[vb + lane] = vq[d] + C2V[thr * q + lane];
                }
        }
    };

    int it = 0, ok = 0;
    for (;;) { // frames of this workgroup: its own (one workgroup per frame) or, persistent, the next of the batch (see k_nb_ems)
    if (a.work) {
        if (tid == 0) flag[1] = atomicAdd(a.work, 1);
        __syncthreads();
        frame = __builtin_amdgcn_readfirstlane(flag[1]);
    }
    if (frame >= a.B) break;
    for (int i = tid; i < TC * q; i += NT) C2V[i] = 0.0f; // :395-404
    if (tid == 0) flag[0] = 0;
    {   // initial vectors (:363-393): max over L_ch, LLR[0] = max, LLR[k] = max - L_ch[k-1]
        const float *Lch = a.Lch + (size_t)frame * N * (q - 1);
        for (int col = wave; col < N; col += nwaves) {
            const float lc = (lane < q - 1) ? Lch[col * (q - 1) + lane] : -__builtin_inff();
            const float mx = nb_wave_max(lc);
            const float lm = (lane >= 1 && act) ? Lch[col * (q - 1) + lane - 1] : 0.0f;
            if (act) LLR[col * q + lane] = (lane == 0) ? mx : mx - lm;
        }
    }
    __syncthreads();
    it = 0;
    ok = 0;
    while (it < a.max_iter) {
        it++;
        // ---- variable nodes: (flooding) LLR += c2v in ascending d (:425-433); first-minimum decision (:434 / :602-605)
        for (int col = wave; col < N; col += nwaves) {
            float llr = act ? LLR[col * q + lane] : 0.0f;
            if (!LAYERED) {
                const int w = t_vn_w[col];
                for (int d = 0; d < w; d++) llr = llr + (act ? C2V[t_vn_thr[col * dv + d] * q + lane] : 0.0f);
                if (act) LLR[col * q + lane] = llr;
            }
            float m;
            const int dec = first_min(llr, m);
            if (lane == 0) outs[col] = dec;
            if (!LAYERED) { // v2c = LLR - c2v (:467-476); not observable when the frame leaves below
                const int w = t_vn_w[col];
                for (int d = 0; d < w; d++)
                    if (act) V2C[(col * dv + d) * q + lane] = llr - C2V[t_vn_thr[col * dv + d] * q + lane];
            }
        }
        __syncthreads();
        // ---- syndrome (:437-450)
        if (tid < M) {
            int s = 0;
            for (int i = 0; i < t_cn_w[tid]; i++) s ^= mulb[outs[t_cn_vn[tid * dc + i]] * q + t_cn_gf[tid * dc + i]];
            if (s) flag[0] = 1;
        }
        __syncthreads();
        if (flag[0] == 0) {
            it--;
            ok = 1;
            break;
        }
        // ---- check rows
        if (!LAYERED) {
            for (int row = wave; row < ((TMM_ABLATE & 2) ? 0 : M); row += nwaves) do_row(row);
            __syncthreads();
        } else {
            for (int lv = 0; lv < a.levels; lv++) {
                for (int i = t_lvl[lv] + wave; i < t_lvl[lv + 1]; i += nwaves) do_row(t_rows[i]);
                __syncthreads();
            }
        }
        if (tid == 0) flag[0] = 0;
    }
    for (int i = tid; i < N; i += NT) a.out[(size_t)frame * N + i] = outs[i];
    if (tid == 0) {
        a.iters[frame] = it;
        a.ok[frame] = ok;
    }
    if (a.LLR)
        for (int i = tid; i < N * q; i += NT) a.LLR[(size_t)frame * N * q + i] = LLR[i];
    if (a.c2v)
        for (int i = tid; i < TC * q; i += NT) {
            const int thr = i / q;
            a.c2v[(size_t)frame * TC * q + i] = (thr % dc < t_cn_w[thr / dc]) ? C2V[i] : 0.0f;
        }
    if (!a.work) break;
    __syncthreads(); // LLR, C2V, the symbols and flag[1] are reused by the next frame
    } // next frame
}

} // namespace cldpc
