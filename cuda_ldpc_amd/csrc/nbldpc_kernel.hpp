// nbldpc_kernel.hpp -- fused GF(q) EMS decoder: one workgroup = one frame, all iterations on-chip.
//
// Bit-exact restatement of the reference's CPU decoder (myNBLDPC/src/LDPC_Decoder.cpp:172-359), whose
// float operation order is the canonical one: the configuration-set walk of ConstructConf
// carries its running sum by reference, so `s + v ... s - v` rounding drift must be replayed in the
// reference's depth-first order.  That order does not depend on the data, so each (check row, edge)
// thread executes a fixed straight-line program over its neighbours' sorted messages.
//
// LDS (one frame):   pairs[NE][q] (float value, int premultiplied symbol as a byte offset), padded per edge
//                    against bank conflicts;  E[M*dc][q+1] the per-(row,edge) max arrays (EMS_L_c2v), one per check
//                    thread with an odd stride (the entry of a leaf is base + (prefix ^ symbol): one v_xad_u32);
//                    the GF multiplication table as bytes.
// Phases per iteration (LDPC_Decoder.cpp:199-313):
//   A  one wave per variable node, lane k <-> vector element k: c2v from E (the double division of
//      :309), LLR = L_ch + sum c2v (:202-214), hard decision (:71-91), v2c = LLR - c2v (:241-251)
//   S  syndrome over GF(q) (:218-238); a frame leaves as soon as it is zero
//   B  one wave per edge: stable descending sort (:17-36,253-269) as a 64-lane bitonic network over the
//      pairwise-distinct keys (order-preserving image of the value, 63 - index)
//   C  one thread per (row, edge): conf(q,1) then conf(Nm,Nc) (:272-303, :319-359) into E
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef NB_ABLATE
#define NB_ABLATE 0 // experiments only (timing, wrong results): 1 no CN phase, 2 no rank loop, 4 no double division, 8 no VN phase A math
#endif

namespace cldpc {

constexpr int kNbThreads = 1024;
__host__ __device__ constexpr int nb_threads(int q) { return 1024; } // the phases are latency chains per wave: more waves, shorter chains
constexpr int kNbMaxDv = 8;
constexpr int kNbMaxW = 6; // row weights handled by the templated walk

struct NbArgs {
    const float *Lch; // [B][N][q-1]
    int *out;         // [B][N]
    int *iters;       // [B]
    int *ok;          // [B]
    float *LLR;       // [B][N][q-1] or nullptr
    float *c2v;       // [B][M][dc][q-1] or nullptr
    const int *vn_w;   // [N]
    const int *vn_thr; // [N][dv]  CN thread (row*dc + slot) at the other end of each VN edge
    const int *vn_gf;  // [N][dv]
    const int *cn_w;   // [M]
    const int *cn_src; // [M][dc]  VN edge (vn*dv + idx) at the other end of each CN slot
    const int *cn_gf;  // [M][dc]
    const int *cn_vn;  // [M][dc]
    const unsigned char *mul; // [q][q]
    int N, M, q, dv, dc, B, Nm, Nc, max_iter, dcmax_cfg;
    int zero_coeff; // some edge coefficient is 0 (the reference's exponent-format files): the max arrays need their -inf fill
    int *work = nullptr;               // k_nb_ems: != nullptr: persistent workgroups take frame after frame from this counter (zeroed by the host)
    float *ws = nullptr;               // k_nb_ems_hbm only: workspace, one slot per workgroup
    unsigned long long ws_stride = 0;  // floats per slot
};

__host__ __device__ inline int nb_pair_stride(int q) { return 2 * q + 2; } // floats per edge (+2: bank skew)

// Maximum over the 64 lanes, in every lane: DPP butterflies inside each row of 16 (no LDS round trips -- six
// dependent ds_bpermute cost ~1 us per variable node), two row broadcasts, one v_readlane.
// (float)((double)x / 1.2) of LDPC_Decoder.cpp:309 without the ~15-instruction f64 division: with r = RN(1/1.2),
// q0 = x*r, e = fma(-1.2, q0, x) (exact residual), q1 = fma(e, r, q0) is the correctly rounded double quotient
// (Markstein's correction step); tests/c/div12_exhaustive.c checks the float result for every finite non-zero float.
// Zeros and infinities divide to themselves.
__device__ __forceinline__ float nb_div12(float x)
{
    constexpr double r = 1.0 / 1.2;
    const double xd = (double)x;
    const double q0 = xd * r;
    const double e = __builtin_fma(-1.2, q0, xd);
    const float q = (float)__builtin_fma(e, r, q0);
    return (x == 0.0f || __builtin_isinf(x)) ? x : q;
}

// Each butterfly step is ONE instruction, v_max_f32 with the DPP pattern on its first operand (through fmaxf and a separate
// v_mov_b32_dpp the compiler emits five: copy, wait states, move, and two canonicalising maxima).  The two row broadcasts write
// the rows their masks name and leave the others as they are; v_max_f32 returns one of its operands unchanged for the ordered
// values met here (no NaN reaches a decision, see nb_ds_max).
__device__ __forceinline__ float nb_wave_max(float v)
{
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t" // every lane of a row holds the row's maximum
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" // into rows 1 and 3
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" // into rows 2 and 3: lane 63 holds the maximum
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}


// 64-lane bitonic sort, descending, of the pairwise-distinct 64-bit keys (hi, lo).  One compare-exchange step:
// the partner lane's key arrives through DPP (partners 1, 2, 8 lanes away: 14 of the 21 steps) or the LDS crossbar
// (ds_swizzle / ds_bpermute: 4, 16, 32 lanes away), the compare is one v_cmp_gt_u64, the keep-max lane pattern of
// the step is a compile-time constant XNORed into the compare mask on the scalar unit.  Every step is checked
// against a plain compare on the GPU (tests/cpp/dpp_step_test.hip).
__host__ __device__ constexpr unsigned long long nb_keepmax_mask(int K, int J)
{
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) {
        const bool lower = (i & J) == 0, up = (K >= 64) ? true : ((i & K) == 0);
        if (lower == up) m |= 1ull << i;
    }
    return m;
}
template <int K, int J> __device__ __forceinline__ void nb_bitonic_step(uint32_t &hi, uint32_t &lo)
{
    constexpr unsigned long long KM = nb_keepmax_mask(K, J);
    uint32_t phi, plo;
    if constexpr (J == 1 || J == 2 || J == 8) {
        // the partner lane (lane ^ J) is inside the lane's own row of 16: v_mov_b32_dpp, no LDS crossbar.  (Partners 4, 16
        // and 32 lanes away could also stay on the VALU -- masked DPP pairs, v_permlane16/32_swap -- but the phase is bound
        // by VALU issue, DPP and 3-operand instructions cost ~4.4 SIMD-cycles each (tools/valu_rate.hip): measured slower.)
        constexpr int CTRL = (J == 1) ? 0xB1 /* quad_perm [1,0,3,2] */ : (J == 2) ? 0x4E /* quad_perm [2,3,0,1] */ : 0x128 /* row_ror:8 */;
        phi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
        plo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    } else if constexpr (J < 32) {
        phi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)hi, (J << 10) | 0x1f);
        plo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)lo, (J << 10) | 0x1f);
    } else {
        phi = (uint32_t)__shfl_xor((int)hi, 32, 64);
        plo = (uint32_t)__shfl_xor((int)lo, 32, 64);
    }
    // one 64-bit compare into a scalar mask (any SGPR pair: the sorts in flight do not queue on VCC), the step's
    // keep-max lane pattern XNORed in on the scalar unit, two selects
    const bool gt = (((unsigned long long)phi << 32) | plo) > (((unsigned long long)hi << 32) | lo);
    const bool take = __builtin_amdgcn_inverse_ballot_w64(~(__builtin_amdgcn_ballot_w64(gt) ^ KM));
    hi = take ? phi : hi;
    lo = take ? plo : lo;
}
// W independent sorts advance through the network together: each step waits ~an LDS round trip for its
// partner keys, so one sort alone is latency-bound; W of them fill that latency with each other's work.
template <int K, int J, int W> __device__ __forceinline__ void nb_bitonic_merge(uint32_t (&hi)[W], uint32_t (&lo)[W])
{
#pragma unroll
    for (int i = 0; i < W; i++) nb_bitonic_step<K, J>(hi[i], lo[i]);
    if constexpr (J > 1) nb_bitonic_merge<K, J / 2, W>(hi, lo);
}
template <int K, int W> __device__ __forceinline__ void nb_bitonic_sort(uint32_t (&hi)[W], uint32_t (&lo)[W])
{
    if constexpr (K > 2) nb_bitonic_sort<K / 2, W>(hi, lo);
    nb_bitonic_merge<K, K / 2, W>(hi, lo);
}

// The same network on ONE 32-bit key per lane: the order-preserving image of the value with its low 6 bits replaced by
// 63 - index.  Exact only when no two values of the vector differ in nothing but those 6 bits, so the caller VERIFIES the
// resulting permutation against the full (value, index) order and repeats the vector on the 64-bit network when a neighbouring
// pair is out of order (the keys are pairwise distinct, so the permutation whose neighbours are all in order is the one the
// stable sort produces).
// One compare-exchange = partner fetch + v_med3_u32(k, partner, B) with B = 0xffffffff in the lanes that keep the larger key
// (the median of {k, partner, max} is the larger one) and 0 in the others (the smaller one): no compare, no mask, no select.  B is
// the same for the four sorts in flight; the 21 keep-max lane patterns of the network are one bit each of a per-lane word
// (nb_keepmax_word), so a step costs one v_bfe_i32 and, per sort, a v_mov_b32_dpp (or ds_swizzle) and a v_med3_u32.
__host__ __device__ constexpr int nb_step_index(int K, int J)
{
    int n = 0; // steps in execution order: K = 2, 4, ..., 64; inside a merge J = K/2, K/4, ..., 1
    for (int k = 2; k < K; k *= 2)
        for (int j = k / 2; j >= 1; j /= 2) n++;
    for (int j = K / 2; j > J; j /= 2) n++;
    return n;
}
__device__ __forceinline__ uint32_t nb_keepmax_word(int lane)
{
    uint32_t w = 0;
    for (int K = 2; K <= 64; K *= 2)
        for (int J = K / 2; J >= 1; J /= 2) {
            const bool lower = (lane & J) == 0, up = (K >= 64) ? true : ((lane & K) == 0);
            if (lower == up) w |= 1u << nb_step_index(K, J);
        }
    return w;
}
template <int K, int J> __device__ __forceinline__ void nb_bitonic_step32x4(uint32_t (&k)[4], uint32_t kmw)
{
    const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int)kmw, nb_step_index(K, J), 1); // 0xffffffff where this lane keeps the larger key
    if constexpr (J == 1 || J == 2 || J == 8) {
        // the four sorts interleaved so that every hazard distance (a DPP source written by the VALU needs 2 wait states) is met
        uint32_t p0, p1, p2, p3;
#define NB_DPP_STEP(CTRL)                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                   \
                 "v_mov_b32_dpp %4, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                 "v_mov_b32_dpp %5, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                 "v_mov_b32_dpp %6, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                 "v_mov_b32_dpp %7, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                 "v_med3_u32 %0, %0, %4, %8\n\t"                                                                  \
                 "v_med3_u32 %1, %1, %5, %8\n\t"                                                                  \
                 "v_med3_u32 %2, %2, %6, %8\n\t"                                                                  \
                 "v_med3_u32 %3, %3, %7, %8"                                                                       \
                 : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(B))
        if constexpr (J == 1) { NB_DPP_STEP("quad_perm:[1,0,3,2]"); }
        else if constexpr (J == 2) { NB_DPP_STEP("quad_perm:[2,3,0,1]"); }
        else { NB_DPP_STEP("row_ror:8"); }
#undef NB_DPP_STEP
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t pk;
            if constexpr (J < 32) pk = (uint32_t)__builtin_amdgcn_ds_swizzle((int)k[i], (J << 10) | 0x1f); // partners 4 and 16 lanes away: LDS crossbar
            else pk = (uint32_t)__shfl_xor((int)k[i], 32, 64);
            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(k[i]) : "v"(k[i]), "v"(pk), "v"(B));
        }
    }
}
template <int K, int J> __device__ __forceinline__ void nb_bitonic_merge32(uint32_t (&k)[4], uint32_t kmw)
{
    nb_bitonic_step32x4<K, J>(k, kmw);
    if constexpr (J > 1) nb_bitonic_merge32<K, J / 2>(k, kmw);
}
// kmw_cold: the same word again for the merges up to KC -- a caller inside a loop passes an opaque copy, so that the lane masks of
// those steps are extracted where they are used (one v_bfe_i32 each) instead of being hoisted out of the loop into registers the
// kernel does not have (k_nb_ems2: all 21 hoisted, three of them spilled to scratch and reloaded inside every sort)
template <int K, int KC = 0> __device__ __forceinline__ void nb_bitonic_sort32(uint32_t (&k)[4], uint32_t kmw, uint32_t kmw_cold = 0)
{
    if constexpr (K > 2) nb_bitonic_sort32<K / 2, KC>(k, kmw, kmw_cold);
    nb_bitonic_merge32<K, K / 2>(k, K <= KC ? kmw_cold : kmw);
}

template <int NACT> struct NbCn {
    int pb[NACT];                   // float index of each active neighbour's sorted pairs
    float v0[NACT], v1[NACT];       // its two largest values   (sort_L_v2c[..][0..1])
    int m0[NACT], m1[NACT];         // and their premultiplied symbols, as byte offsets (4 * symbol) into the thread's max array
    float s;                        // sumNonLLR, carried by reference through the walk
    char *E;                        // this thread's max array (EMS_L_c2v of its (row, edge)): q + 1 floats, entry of symbol x at byte 4x
    const float *pairs;
};
__device__ __forceinline__ float &nb_e(char *E, int sym4) { return *reinterpret_cast<float *>(E + sym4); }
// E[sym] = max(E[sym], v) as ONE LDS instruction instead of load / compare / select / store.  Equal to the reference's
// `if (s > E[sym]) E[sym] = s` (LDPC_Decoder.cpp:322-325) here: a running sum that starts at +0 and only adds and subtracts is
// never -0 (the one case in which the LDS maximum and the strict `>` differ), and there are no NaNs.
__device__ __forceinline__ void nb_ds_max(char *E, int sym4, float v)
{
    typedef __attribute__((address_space(3))) char lds_byte;
    asm volatile("ds_max_f32 %0, %1" : : "v"((lds_byte *)(E + sym4)), "v"(v) : "memory");
}

// Sub-walk entered with diff == 1 (one deviation already spent) at depth D: exactly one leaf (all
// remaining positions at k = 0), then on the way back every position tries k = 1, exceeds Nc = 1
// and is undone (LDPC_Decoder.cpp:347-354).  Returns the leaf (sum, symbol); c.s keeps the drift.
template <int D, int NACT> __device__ __forceinline__ void nb_t1(NbCn<NACT> &c, int symbase, float &sLeaf, int &symLeaf)
{
    float s = c.s;
    int sym = symbase;
#pragma unroll
    for (int d = D; d < NACT; d++) {
        s = s + c.v0[d];
        sym ^= c.m0[d];
    }
    sLeaf = s;
    symLeaf = sym;
#pragma unroll
    for (int d = NACT - 1; d >= D; d--) {
        s = s - c.v0[d];
        s = s + c.v1[d];
        s = s - c.v1[d];
    }
    c.s = s;
}

// conf(q, 1) entered with diff == 0 at depth D (LDPC_Decoder.cpp:286 via :319-359).
// CH = leaves batched per LDS round trip (their symbols are pairwise distinct inside one k-loop,
// because the sorted symbol list is a permutation and multiplication by h != 0 is a bijection).
// The float chain through c.s is the critical path of the whole phase (one rounding step after another, as the
// reference's by-reference recursion dictates); everything else is kept off it: the next batch's pairs and this
// batch's E entries (their addresses are pure symbol arithmetic) are requested BEFORE the chain runs.
template <int D, int NACT, int Q, bool ZS> __device__ __forceinline__ void nb_t0(NbCn<NACT> &c, int symbase)
{
    if constexpr (D == NACT - 1) {
        // deepest position with an all-zero prefix.  With non-zero coefficients these q leaves touch every symbol exactly once, so
        // they initialise E (no -DBL_MAX fill, LDPC_Decoder.cpp:277-280, needed) with plain stores.  A code with a coefficient 0 --
        // the reference's exponent-format files, read as it reads them -- sends all q leaves to ONE symbol: for such codes (ZS,
        // NbArgs::zero_coeff set by the host) E is filled with -inf at the end of phase B and these leaves are maxima like the others.
        constexpr int CH = (Q % 8 == 0) ? 8 : 1;
        for (int k0 = 0; k0 < Q; k0 += CH) {
            float v[CH];
            int m[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const float2 pr = *reinterpret_cast<const float2 *>(c.pairs + c.pb[D] + 2 * (k0 + i));
                v[i] = pr.x;
                m[i] = __float_as_int(pr.y);
            }
#pragma unroll
            for (int i = 0; i < CH; i++) {
                c.s = c.s + v[i];
                if constexpr (ZS) nb_ds_max(c.E, symbase ^ m[i], c.s);
                else nb_e(c.E, symbase ^ m[i]) = c.s; // the XOR and the add to the base are one v_xad_u32
                c.s = c.s - v[i];
            }
        }
    } else {
        c.s = c.s + c.v0[D];
        nb_t0<D + 1, NACT, Q, ZS>(c, symbase ^ c.m0[D]);
        c.s = c.s - c.v0[D];
        int sfx = symbase; // symbol of a leaf that deviates here: every deeper position at k = 0
#pragma unroll
        for (int d = D + 1; d < NACT; d++) sfx ^= c.m0[d];
        constexpr int CH = ((Q - 1) % 7 == 0) ? 7 : ((Q - 1) % 5 == 0) ? 5 : ((Q - 1) % 3 == 0) ? 3 : 1; // 63 = 9 x 7, 255 = 51 x 5, 15 = 3 x 5
        // Two register sets for the prefetched pairs, used in turn (the loop body is written out twice): with one set, read by this
        // batch and overwritten by the prefetch of the next, the compiler copied all 2 CH registers every round (two v_mov_b32 per leaf
        // between the links of a chain that pays 4 cycles for every independent instruction).
        constexpr int NB = (Q - 1) / CH; // batches: 9 (q = 64), 51 (q = 256), 3 (q = 16)
        static_assert(NB * CH == Q - 1, "whole batches");
        float2 prA[CH], prB[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) prA[i] = *reinterpret_cast<const float2 *>(c.pairs + c.pb[D] + 2 * (1 + i));
        auto batch = [&](const float2 (&cur)[CH], float2 (&nxt)[CH], int kn) { // walk the CH leaves of `cur`; meanwhile fetch batch kn into `nxt`
            float v[CH], sl[CH];
            int sy[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                v[i] = cur[i].x;
                sy[i] = sfx ^ __float_as_int(cur[i].y); // byte offset of the leaf's symbol
            }
#pragma unroll
            for (int i = 0; i < CH; i++) nxt[i] = *reinterpret_cast<const float2 *>(c.pairs + c.pb[D] + 2 * (kn + i));
#pragma unroll
            for (int i = 0; i < CH; i++) {
                int unused;
                c.s = c.s + v[i];
                nb_t1<D + 1, NACT>(c, 0, sl[i], unused);
                c.s = c.s - v[i];
            }
#pragma unroll
            for (int i = 0; i < CH; i++) nb_ds_max(c.E, sy[i], sl[i]); // :322-325
        };
#pragma unroll 1
        for (int b = 0; b + 1 < NB; b += 2) {
            batch(prA, prB, 1 + (b + 1) * CH);
            batch(prB, prA, 1 + min(b + 2, NB - 1) * CH); // (the last pair of an even count re-reads a valid batch: no branch in the loop)
        }
        if constexpr (NB % 2 == 1) batch(prA, prB, 1); // the odd batch out (its prefetch is a dummy)
    }
}

// General conf(Nm, Nc) walk (LDPC_Decoder.cpp:319-359), the reference's recursion as is.
template <int D, int NACT> __device__ void nb_conf(NbCn<NACT> &c, int symbase, int diff, int Nm, int Nc)
{
    if constexpr (D == NACT) {
        nb_ds_max(c.E, symbase, c.s); // :322-325
    } else {
        for (int k = 0; k < Nm; k++) {
            float v;
            int m;
            if (k == 0) { v = c.v0[D]; m = c.m0[D]; }
            else if (k == 1) { v = c.v1[D]; m = c.m1[D]; }
            else {
                const float2 pr = *reinterpret_cast<const float2 *>(c.pairs + c.pb[D] + 2 * k);
                v = pr.x;
                m = __float_as_int(pr.y);
            }
            c.s = c.s + v;
            const int d2 = diff + (k != 0 ? 1 : 0);
            const bool go = d2 <= Nc;
            if (go) nb_conf<D + 1, NACT>(c, symbase ^ m, d2, Nm, Nc);
            c.s = c.s - v;
            if (!go) break;
        }
    }
}

template <int W, int Q>
__device__ void nb_cn_update(const NbArgs &a, const unsigned short *cn_src, const float *pairs, float *E, int QP, int row, int e, int thr)
{
    constexpr int NACT = W - 1;
    NbCn<NACT> c;
    c.E = reinterpret_cast<char *>(E + thr * QP);
    c.pairs = pairs;
    const int PST = nb_pair_stride(a.q);
#pragma unroll
    for (int i = 0; i < NACT; i++) {
        const int pos = i + (i >= e ? 1 : 0); // ascending positions, skipping `except` (:327-331)
        c.pb[i] = cn_src[row * a.dc + pos] * PST;
        const float2 p0 = *reinterpret_cast<const float2 *>(pairs + c.pb[i]);
        const float2 p1 = *reinterpret_cast<const float2 *>(pairs + c.pb[i] + 2);
        c.v0[i] = p0.x; c.m0[i] = __float_as_int(p0.y);
        c.v1[i] = p1.x; c.m1[i] = __float_as_int(p1.y);
    }
    c.s = 0.0f;
    if (a.zero_coeff) nb_t0<0, NACT, Q, true>(c, 0); // ConstructConf(GFQ, 1) :286 (workgroup-uniform branch)
    else nb_t0<0, NACT, Q, false>(c, 0);
    c.s = 0.0f;
    int Nc = a.Nc;
    if (a.Nc == a.dcmax_cfg - 1) Nc = W - 1; // :294-297
    nb_conf<0, NACT>(c, 0, 0, a.Nm, Nc);     // ConstructConf(EMS_Nm, EMS_Nc) :300
}

// Q: field size (= lanes used per vector); DVM: bound on the column weight (loops over a node's edges are unrolled to it)
template <int Q, int DVM, int NT> __global__ __launch_bounds__(NT) void k_nb_ems(NbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int frame = blockIdx.x; // (a persistent workgroup takes its frames from a.work instead, see below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = NT / 64;
    constexpr int q = Q; // == a.q (the host picks the instantiation)
    const int N = a.N, M = a.M, dv = a.dv, dc = a.dc;
    const int NE = N * dv, TC = M * dc, PST = nb_pair_stride(q);
    constexpr int QP = q + 1; // E is [M*dc][q + 1]: one max array per (row, edge) thread, odd stride: lanes that walk different arrays
                              // and lanes that read one array across symbols both spread over the banks
    float *pairs = lds;                    // [NE][PST]
    float *E = pairs + NE * PST;           // [q][TC]
    int *outs = reinterpret_cast<int *>(E + TC * QP);  // [N]
    int *flag = outs + N;                  // [4]
    unsigned char *mulb = reinterpret_cast<unsigned char *>(flag + 4); // [q][q]
    // graph tables as u16 in LDS: a global load whose value steers a branch or an address costs ~1 us each,
    // and the phases below would pay it per node / per edge / per iteration
    unsigned short *t_vn_w = reinterpret_cast<unsigned short *>(mulb + q * q); // [N]
    unsigned short *t_vn_thr = t_vn_w + N;      // [N][dv]
    unsigned short *t_vn_gf = t_vn_thr + NE;    // [N][dv]
    unsigned short *t_cn_w = t_vn_gf + NE;      // [M]
    unsigned short *t_cn_src = t_cn_w + M;      // [M][dc]
    unsigned short *t_cn_gf = t_cn_src + TC;    // [M][dc]
    unsigned short *t_cn_vn = t_cn_gf + TC;     // [M][dc]
    unsigned char *t_elive = reinterpret_cast<unsigned char *>(t_cn_vn + TC); // [NE] slot d of a column is an edge when d < its weight
    for (int i = tid; i < NE; i += NT) t_elive[i] = (i % dv) < a.vn_w[i / dv]; // (a coefficient may be 0: the reference's exponent-format files)
    for (int i = tid; i < N; i += NT) t_vn_w[i] = (unsigned short)a.vn_w[i];
    for (int i = tid; i < NE; i += NT) { t_vn_thr[i] = (unsigned short)a.vn_thr[i]; t_vn_gf[i] = (unsigned short)a.vn_gf[i]; }
    for (int i = tid; i < M; i += NT) t_cn_w[i] = (unsigned short)a.cn_w[i];
    for (int i = tid; i < TC; i += NT) {
        t_cn_src[i] = (unsigned short)a.cn_src[i]; t_cn_gf[i] = (unsigned short)a.cn_gf[i]; t_cn_vn[i] = (unsigned short)a.cn_vn[i];
    }

    for (int i = tid; i < q * q; i += NT) mulb[i] = a.mul[i];
    __syncthreads();

    const float *Lch = nullptr; // of the frame being decoded (set per frame below)
    float *LLRo = nullptr;
    const bool active = lane < q - 1;          // lanes 0..q-2 <-> field elements 1..q-1
    const int sym = active ? lane + 1 : 0;     // lane q-1 carries element 0 (value 0, :250)
    constexpr int CPW = 96 * 64 / NT; // columns per wave whose channel vector stays in registers (N <= 96)
    float lch[CPW];
    // The graph does not change between iterations: for narrow codes (dv <= 2) each lane keeps, per column and edge, the
    // LDS offset of ITS entry of the check thread's max array, E[mul(sym, h)][thr], so an iteration starts with that one
    // read instead of three dependent table look-ups.  E[0][thr] (the reference's EMS_L_c2v[0]) is the same read in the
    // lane that carries field element 0.
    constexpr bool kInvariantOffsets = DVM <= 2;
    int evoff[kInvariantOffsets ? CPW : 1][DVM];
    unsigned wmask = 0; // bit ci*DVM + d: edge d of column ci exists
    if constexpr (kInvariantOffsets) {
#pragma unroll
        for (int ci = 0; ci < CPW; ci++) {
            const int col = min(wave + ci * nwaves, N - 1);
            const int w = t_vn_w[col];
#pragma unroll
            for (int d = 0; d < DVM; d++) {
                const int dd = min(d, dv - 1);
                evoff[ci][d] = t_vn_thr[col * dv + dd] * QP + mulb[sym * q + t_vn_gf[col * dv + dd]];
                if (d < w) wmask |= 1u << (ci * DVM + d);
            }
        }
    }
    // one variable node, lane <-> field element (LDPC_Decoder.cpp:202-251)
    auto vn_column = [&](int col, float llr, bool store, int ci) {
        const int w = t_vn_w[col];
        float c2[DVM];
#pragma unroll
        for (int d = 0; d < DVM; d++) {
            float e0, ev;
            bool on;
            if (kInvariantOffsets && ci >= 0) {
                ev = E[evoff[kInvariantOffsets ? ci : 0][d]];
                e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), Q - 1));
                on = (wmask >> (ci * DVM + d)) & 1u;
            } else {
                const int dd = min(d, dv - 1);
                const int thr = t_vn_thr[col * dv + dd], h = t_vn_gf[col * dv + dd];
                e0 = E[thr * QP];
                ev = E[thr * QP + mulb[sym * q + h]];
                on = d < w;
            }
            const float c = nb_div12(ev - e0); // :309, double division
            c2[d] = on ? c : 0.0f;
            llr = on ? llr + c : llr;          // :208-213, ascending d
        }
        // DecideLLRVector (:71-91): running max from 0, strict >, first maximum wins
        const float v = active ? llr : -__builtin_inff();
        const float mx = nb_wave_max(v);
        const unsigned long long eq = __ballot(active && v == mx);
        const int dec = (mx > 0.0f) ? (int)__builtin_ctzll(eq) + 1 : 0;
        if (store) {
            if (lane == 0) outs[col] = dec;
            if (LLRo && active) LLRo[col * (q - 1) + lane] = llr;
#pragma unroll
            for (int d = 0; d < DVM; d++)
                if (d < w && lane < q) pairs[(col * dv + d) * PST + 2 * lane] = active ? llr - c2[d] : 0.0f; // :241-251
        }
    };
    int it = 0, ok = 0;
    // Frames differ in their iteration counts by a factor of 20 (a frame leaves when its syndrome is zero), and one frame fills a
    // CU: dispatched one workgroup per frame the CUs ended up 17 % apart (12.2 ms where the sum of the iterations says 10.1,
    // tools/nb_fixed_cost.py).  PERSISTENT workgroups -- the grid fills the chip once -- take frame after frame from a counter, and
    // load the graph tables and the GF table into LDS once instead of once per frame.
    for (;;) {
    if (a.work) {
        if (tid == 0) flag[1] = atomicAdd(a.work, 1);
        __syncthreads();
        frame = __builtin_amdgcn_readfirstlane(flag[1]); // uniform: the frame's pointers then live in scalar registers
    }
    if (frame >= a.B) break;
    Lch = a.Lch + (size_t)frame * N * (q - 1);
    LLRo = a.LLR ? a.LLR + (size_t)frame * N * (q - 1) : nullptr;
    {
        int lo = lane; // opaque per frame: left alone the compiler keeps the six load offsets of a lane across the whole frame loop
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int ci = 0; ci < CPW; ci++) {
            const int col = min(wave + ci * nwaves, N - 1);
            lch[ci] = active ? Lch[col * (q - 1) + lo] : 0.0f;
        }
    }
    for (int i = tid; i < TC * QP; i += NT) E[i] = 0.0f; // L_c2v = 0 (:185-193): (0-0)/1.2 == +0
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    it = 0;
    ok = 0;
#ifdef NB_STAMP
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define NB_T(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define NB_T(i)
#endif
    while (it < a.max_iter) {
        it++;
        NB_T(5)
        // ---- A: variable nodes ------------------------------------------------------------
        // The first CPW columns of a wave run as straight-line code on a clamped column index (only the stores are
        // conditional), so that their table look-ups, E gathers and double divisions overlap; L_ch stays in registers.
        int wv = wave; // opaque per iteration: otherwise every column's addresses are hoisted out of the loop and spilled
        asm volatile("" : "+s"(wv));
#pragma unroll
        for (int ci = 0; ci < CPW; ci++) {
            const int col = wv + ci * nwaves;
            vn_column(min(col, N - 1), lch[ci], col < N, ci);
            if (ci & 1) __builtin_amdgcn_sched_barrier(0); // two columns in flight: bounds the registers (1024 threads: 128 VGPRs)
        }
        for (int col = wave + CPW * nwaves; col < N; col += nwaves) vn_column(col, active ? Lch[col * (q - 1) + lane] : 0.0f, true, -1);
        NB_T(0)
        __syncthreads();
        NB_T(4)
        // ---- S: syndrome (:218-238) ----------------------------------------------------------
        if (tid < M) {
            int s = 0;
            for (int i = 0; i < t_cn_w[tid]; i++) s ^= mulb[outs[t_cn_vn[tid * dc + i]] * q + t_cn_gf[tid * dc + i]];
            if (s) flag[0] = 1;
        }
        NB_T(1)
        __syncthreads();
        NB_T(4)
        if (flag[0] == 0) {
            it--; // :236
            ok = 1;
            break;
        }
        // ---- B: stable descending sort of every v2c vector (:17-36, :253-269) -----------------
        constexpr int SW = 4; // sorts in flight per wave
        const uint32_t kmw = nb_keepmax_word(lane);
        for (int e0 = wave * SW; e0 < NE; e0 += nwaves * SW) {
            uint32_t khi[SW], k32[SW];
            bool live[SW];
            // premultiplied symbol of THIS lane's field element on the four edges, one byte each: a constant of the original lane, so
            // after the sort it comes over with the permutation (one ds_bpermute) instead of through t_vn_gf -> mulb behind the network
            static_assert(SW == 4, "one word of symbol bytes per group");
            uint32_t pmw = 0;
#pragma unroll
            for (int i = 0; i < SW; i++) pmw |= (uint32_t)mulb[(active ? lane + 1 : 0) * q + t_vn_gf[min(e0 + i, NE - 1)]] << (8 * i);
#pragma unroll
            for (int i = 0; i < SW; i++) {
                const int edge = e0 + i;
                live[i] = edge < NE && t_elive[min(edge, NE - 1)];
                const float val = (live[i] && lane < q) ? pairs[edge * PST + 2 * lane] : 0.0f;
                // order-preserving integer image of the float; +0.0f folds -0 onto +0 (they compare equal)
                const uint32_t b = __float_as_uint(val + 0.0f);
                khi[i] = (lane < q) ? (b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) : 0u;
                // stable descending order = descending order of the distinct keys (image, 63 - index); first on the 32-bit keys
                k32[i] = (khi[i] & 0xffffffc0u) | (63u - (unsigned)lane);
            }
            if (!(NB_ABLATE & 2)) nb_bitonic_sort32<64>(k32, kmw);
            // verify: position `lane` holds element idx; its successor must be smaller in (image, 63 - index)
            int idx[SW];
            bool redo = false;
#pragma unroll
            for (int i = 0; i < SW; i++) idx[i] = (int)(~k32[i] & 63u); // the key's low bits are 63 - (original position)
            // two neighbours of the sorted order agree in the 26 bits the short keys carry?  (key ^ next lane's key in one instruction
            // each; lane 63 has no neighbour.)  If none do, the short-key order IS the order of the full (value, index) keys.
            bool amb;
            {
                uint32_t x0, x1, x2, x3;
                asm("s_nop 1\n\t"
                    "v_xor_b32_dpp %0, %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_xor_b32_dpp %1, %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_xor_b32_dpp %2, %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_xor_b32_dpp %3, %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
                    : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(k32[0]), "v"(k32[1]), "v"(k32[2]), "v"(k32[3]));
                const unsigned long long am = __builtin_amdgcn_ballot_w64(x0 < 64u) | __builtin_amdgcn_ballot_w64(x1 < 64u) |
                                              __builtin_amdgcn_ballot_w64(x2 < 64u) | __builtin_amdgcn_ballot_w64(x3 < 64u);
                amb = (am & 0x7fffffffffffffffull) != 0ull;
            }
            if (amb) { // verify against the full keys
                uint32_t img[SW];
#pragma unroll
                for (int i = 0; i < SW; i++) img[i] = (uint32_t)__shfl((int)khi[i], idx[i], 64); // full image of the element now at this position
#pragma unroll
                for (int i = 0; i < SW; i++) {
                    // the successor's (image, index): DPP wave_shl:1 (lane i reads lane i + 1; lane 63 is not looked at)
                    const uint32_t nimg = (uint32_t)__builtin_amdgcn_update_dpp((int)img[i], (int)img[i], 0x130, 0xf, 0xf, false);
                    const int nidx = __builtin_amdgcn_update_dpp(idx[i], idx[i], 0x130, 0xf, 0xf, false);
                    const bool in_order = img[i] > nimg || (img[i] == nimg && idx[i] < nidx);
                    redo = redo || (__builtin_amdgcn_ballot_w64(!in_order && lane < 63) != 0ull);
                }
            }
            if (redo) { // two values that differ only in their low 6 bits: this group again, on the full keys (wave-uniform branch)
                uint32_t klo[SW];
#pragma unroll
                for (int i = 0; i < SW; i++) klo[i] = 63u - (unsigned)lane;
                nb_bitonic_sort<64, SW>(khi, klo);
#pragma unroll
                for (int i = 0; i < SW; i++) idx[i] = 63 - (int)klo[i];
            }
#pragma unroll
            for (int i = 0; i < SW; i++) {
                const int edge = e0 + i;
                // idx: original position of the element that belongs at position `lane`; its premultiplied symbol
                // (GFMultiply(sort_Entr_v2c, linkVNs_GF) of :334) is byte i of that lane's pmw
                const uint32_t pmo = (uint32_t)__builtin_amdgcn_ds_bpermute(idx[i] << 2, (int)pmw);
                if (live[i] && lane < q) {
                    float2 pr;
                    pr.x = pairs[edge * PST + 2 * idx[i]];
                    pr.y = __int_as_float((int)((pmo >> (8 * i)) & 0xffu) << 2); // as a byte offset into the thread's max array
                    *reinterpret_cast<float2 *>(pairs + edge * PST + 2 * lane) = pr;
                }
            }
        }
        if (a.zero_coeff) // EMS_L_c2v = -DBL_MAX (:277-280), see nb_t0; nobody reads E between phases A and C
            for (int i = tid; i < TC * QP; i += NT) E[i] = -__builtin_inff();
        NB_T(2)
        __syncthreads();
        NB_T(4)
        // ---- C: check nodes (:272-303) -----------------------------------------------------------
        if (tid < TC) {
            const int row = tid / dc, e = tid - row * dc, w = t_cn_w[row];
            if (e < w && !(NB_ABLATE & 1)) {
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
        NB_T(3)
        __syncthreads();
        NB_T(4)
        if (tid == 0) flag[0] = 0; // next write is two barriers away, last read was two barriers ago
    }
#ifdef NB_STAMP
    if (tid == 0 && a.c2v && frame == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.c2v);
        for (int i = 0; i < 6; i++) o[i] = tacc[i];
    }
    if (frame == 0) return;
#endif
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

// Demodulate, BPSK branch (LDPC_Decoder.cpp:139-157): one thread per (frame, symbol, element).
__global__ __launch_bounds__(256) void k_nb_demod_bpsk(const float *rx, float sigma, int B, int N, int q, int m, float *Lch)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * N * (q - 1);
    if (id >= total) return;
    const int k = (int)(id % (q - 1)) + 1;
    const size_t bs = id / (q - 1); // b*N + s
    const float *r = rx + bs * m;
    const float s2 = sigma * sigma;
    float acc = 0.0f;
    for (int b = 0; b < m; b++)
        if ((k & (1 << b)) != 0) acc += (float)(-2) * r[b] / s2;
    Lch[id] = acc;
}

// Demodulate, n_QAM != 2 branch (LDPC_Decoder.cpp:160-169): one received point per code symbol, float arithmetic in the
// reference's order.  rx [B][N][2] (Real, Image), con [q][2].
__global__ __launch_bounds__(256) void k_nb_demod_qam(const float *rx, const float *con, float sigma, int B, int N, int q, float *Lch)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * N * (q - 1);
    if (id >= total) return;
    const int k = (int)(id % (q - 1)) + 1;
    const size_t bs = id / (q - 1); // b*N + s
    const float yr = rx[2 * bs], yi = rx[2 * bs + 1];
    const float c0r = con[0], c0i = con[1], ckr = con[2 * k], cki = con[2 * k + 1];
    Lch[id] = ((2 * yr - c0r - ckr) * (ckr - c0r) + (2 * yi - c0i - cki) * (cki - c0i)) / (2 * sigma * sigma);
}

// Statistic (Simulation.cpp:256-279): one thread per frame.
__global__ __launch_bounds__(256) void k_nb_statistic(const int *out, const int *iters, const int *ok, const int *cw, int B, int N,
                                                      long long *counters)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    long long v[4] = {0, 0, 0, 0};
    if (f < B) {
        int err = 0;
        for (int i = 0; i < N; i++) err += (out[(size_t)f * N + i] != cw[i]) ? 1 : 0;
        v[0] = err != 0;
        v[1] = err;
        v[2] = iters[f];
        v[3] = ok[f];
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        long long x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd((unsigned long long *)&counters[c], (unsigned long long)x);
    }
}

} // namespace cldpc
