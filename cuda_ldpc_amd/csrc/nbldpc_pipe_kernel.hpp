// nbldpc_pipe_kernel.hpp -- k_nb_ems2: the fused GF(q) EMS decoder with TWO frames in flight per workgroup.
//
// Same arithmetic, same order, same bits as k_nb_ems (nbldpc_kernel.hpp; reference: myNBLDPC/src/LDPC_Decoder.cpp:172-359).  What
// changes is WHEN things run.  k_nb_ems runs the phases of one frame back to back: variable nodes (A), syndrome (S), the
// 192 sorts (B), then the check-node walk (C) -- 192 chains of ~1 260 DEPENDENT float additions (the reference's by-reference
// running sum) on 3 of the 16 waves while 13 wait: 16.5 k of an iteration's 40 k cycles at one instruction every ~8.5 cycles per
// SIMD (stamps in profiles/r03_nb_conflict_free_layout.txt; tools/dep_chain.hip: it is the addition latency, not the LDS).
// A second frame cannot simply be made resident: one frame's sorted pairs (100 KB) and max arrays (48 KB) fill the CU's LDS.
//
// Here the workgroup has fixed roles -- the first NCW waves (3 for the BDS code) only ever walk check rows, the others
// ("AB waves") do phases A, S, B -- and two frame slots that run half an iteration apart:
//
//     half-step h (slot s = h & 1 in its A/S/B half, slot c = 1 - s in its C half)
//       stage 1a  AB waves: wait for the count of finished walks; store slot c's sorted pairs, kept in REGISTERS since its phase
//                 B, into the one `pairs` array (slot s's pairs, which its walk read in the previous half-step, are dead); read the
//                 max-array entries E of this lane's edges of slot s (written by C(s) in the previous half-step)
//       -- barrier (the only one) --
//       stage 1b  C waves: C(c): pairs -> E, then count off   |  AB waves: A(s) on the values read: c2v, LLR, hard decision,
//       stage 2                                                |  v2c(s) -> registers; count off and poll until all AB waves are
//                                                              |  there; S(s) (every wave for itself); B(s): v2c(s) sorted in
//                                                              |  registers; request the next half-step's channel values
//
// so one `pairs` array and one `E` array serve both frames (E is free once A(s) has read it; pairs is free once C has read it)
// and the walk's latency chains run beside the other frame's phase A and sorts.  A frame whose syndrome is zero (or that has used
// max_iter iterations) leaves in its S; its slot takes the next frame from the counter.  The channel vectors are re-read from
// global memory every iteration (24 KB per frame, L2-resident) instead of living in registers: the registers hold v2c / the
// sorted pairs.  Not offered here (k_nb_ems takes those calls): codes with a zero coefficient, column weights above 2, the
// L_c2v state output.
#pragma once
#include "nbldpc_kernel.hpp"

#ifndef NB_PIPE_PRIO
#define NB_PIPE_PRIO 1 // experiments (0 = none): 1 = the sorting waves above the walking waves in stage 2, 2 = the walking waves above
#endif

namespace cldpc {

constexpr int kNbPipeCpw = 8; // column slots per AB wave (13 AB waves x 8 >= 96 columns)
#ifndef NB_PIPE_COLD_K
#define NB_PIPE_COLD_K 4 // the lane masks of the merges up to this size are extracted inside the loop (see nb_bitonic_sort32)
#endif
#ifndef NB_PIPE_ONE_S
#define NB_PIPE_ONE_S 0 // experiment: 1 = the last sorting wave to finish phase A computes the syndrome for all, 0 = every wave for itself
#endif
#ifndef NB_PIPE_LATE_LCH
#define NB_PIPE_LATE_LCH 1 // experiment: 1 = the next phase A's channel values are requested after the sorts, 0 = before
#endif
#ifndef NB_PIPE_WALK_COLS
#define NB_PIPE_WALK_COLS 8 // what a walking wave costs its SIMD, in columns of A/S/B work (the column map below balances the SIMDs with it)
#endif

// extra LDS behind k_nb_ems's layout: the second slot's hard symbols, the slot words, the SIMD of every wave and the column map
// and the syndrome's row table (two slots x M rows x 4 or 8 words)
__host__ __device__ inline int nb_pipe_row_words(int dc) { return dc <= 4 ? 4 : 8; }
__host__ __device__ inline size_t nb_pipe_extra_lds(int N, int M, int dc)
{
    return (size_t)(((N + 3) & ~3) + 16 + 16 + 16 * kNbPipeCpw / 4 + 2 * M * nb_pipe_row_words(dc)) * sizeof(int);
}

template <int Q, int NT> __global__ __launch_bounds__(NT) void k_nb_ems2(NbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int q = Q, DVM = 2, CPW = kNbPipeCpw, QP = q + 1, SW = 4, NEW = CPW * DVM; // NEW: edges per AB wave
    static_assert(NEW % SW == 0, "whole groups of sorts");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = NT / 64;
    const int N = a.N, M = a.M, dv = a.dv, dc = a.dc;
    const int NE = N * dv, TC = M * dc, PST = nb_pair_stride(q);
    const int NCW = (TC + 63) >> 6, NAB = nwaves - NCW; // walking waves, A/S/B waves
    const bool is_c = wave < NCW;
    const int abw = wave - NCW;
    float *pairs = lds;                    // [NE][PST]
    float *E = pairs + NE * PST;           // [TC][q + 1]
    int *outs0 = reinterpret_cast<int *>(E + TC * QP);  // [N] slot 0
    int *flag = outs0 + N;                 // [4] (layout of k_nb_ems up to here)
    unsigned char *mulb = reinterpret_cast<unsigned char *>(flag + 4); // [q][q]
    unsigned short *t_vn_w = reinterpret_cast<unsigned short *>(mulb + q * q); // [N]
    unsigned short *t_vn_thr = t_vn_w + N;      // [N][dv]
    unsigned short *t_vn_gf = t_vn_thr + NE;    // [N][dv]
    unsigned short *t_cn_w = t_vn_gf + NE;      // [M]
    unsigned short *t_cn_src = t_cn_w + M;      // [M][dc]
    unsigned short *t_cn_gf = t_cn_src + TC;    // [M][dc]
    unsigned short *t_cn_vn = t_cn_gf + TC;     // [M][dc]
    unsigned char *t_elive = reinterpret_cast<unsigned char *>(t_cn_vn + TC); // [NE] (unused here; keeps the layout of k_nb_ems)
    // (offsets, not integer-cast pointers: the stores below must stay LDS stores)
    const int o1 = (int)((((reinterpret_cast<char *>(t_elive + NE) - reinterpret_cast<char *>(lds)) + 15) & ~15) / 4);
    int *outs1 = reinterpret_cast<int *>(lds) + o1; // [N] slot 1
    int *st = outs1 + ((N + 3) & ~3); // (16-byte aligned) slot words: [0],[1] next frame of slot 0 / 1; [2],[3] half-step (+ 1) in which the slot last retired; [4] pairs hold a sorted frame; [5] walks finished (NCW per half-step); [6] sorting waves past phase A (NAB per half-step); [7] syndrome of this half-step's frame is non-zero; [8] half-step (+ 1) whose [7] is valid
    for (int i = tid; i < N; i += NT) t_vn_w[i] = (unsigned short)a.vn_w[i];
    for (int i = tid; i < NE; i += NT) { t_vn_thr[i] = (unsigned short)a.vn_thr[i]; t_vn_gf[i] = (unsigned short)a.vn_gf[i]; }
    for (int i = tid; i < M; i += NT) t_cn_w[i] = (unsigned short)a.cn_w[i];
    for (int i = tid; i < TC; i += NT) {
        t_cn_src[i] = (unsigned short)a.cn_src[i]; t_cn_gf[i] = (unsigned short)a.cn_gf[i]; t_cn_vn[i] = (unsigned short)a.cn_vn[i];
    }
    for (int i = tid; i < q * q; i += NT) mulb[i] = a.mul[i];
    int *wsimd = st + 16; // [16] the SIMD each wave runs on
    unsigned char *colmap = reinterpret_cast<unsigned char *>(st + 32); // [nwaves][CPW] the columns of each wave (0xff: none)
    if (tid == 0) {
        const int f = atomicAdd(a.work, 2); // the first two frames of this workgroup
        st[0] = f; st[1] = f + 1; st[2] = 0; st[3] = 0; st[4] = 0; st[5] = 0; st[6] = 0; st[7] = 0; st[8] = 0;
    }
    // the syndrome's terms, one word each: byte address of the hard symbol outs_slot[variable] | coefficient << 18; a row's terms
    // side by side (one 16-byte read), padded with (symbol 0 of the slot, coefficient 0): a product with 0 adds nothing
    const int RW = nb_pipe_row_words(dc);
    unsigned *rowtab = reinterpret_cast<unsigned *>(st + 64); // [2][M][RW]
    if (lane == 0) wsimd[wave] = (int)((__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) >> 4) & 3u); // HW_REG_HW_ID.SIMD_ID
    for (int i = tid; i < nwaves * CPW; i += NT) colmap[i] = 0xff;
    __syncthreads();
    for (int i = tid; i < 2 * M * RW; i += NT) {
        const int slot = i / (M * RW), r = (i / RW) % M, k = i % RW;
        const int *o = slot ? outs1 : outs0;
        const bool on = k < t_cn_w[r];
        const int vn = on ? t_cn_vn[r * dc + k] : 0, gf = on ? t_cn_gf[r * dc + k] : 0;
        rowtab[i] = (unsigned)(reinterpret_cast<const char *>(o + vn) - reinterpret_cast<const char *>(lds)) | ((unsigned)gf << 18);
    }
    // Column map.  The workgroup's 16 waves sit on 4 SIMDs, 4 each; a SIMD that hosts a walking wave has 3 sorting waves, one that
    // hosts none has 4, and the walk costs its SIMD about NB_PIPE_WALK_COLS columns' worth of issue slots per half-step.  Columns are
    // dealt out in pairs (one group of DVM * 2 = 4 sorts): each pair goes to the SIMD with the least work so far, there to the wave
    // with the fewest pairs (the youngest wave of a SIMD, which the issue arbiter serves last, ends up with the short hand).
    // (one wave, lane <-> wave of the workgroup; 48 rounds of a 6-step minimum)
    if (wave == 0) {
        const int w = lane;
        const bool isab = w >= NCW && w < nwaves;
        const int sd = wsimd[min(w, nwaves - 1)];
        int ld = 0, ng = 0; // work of this lane's SIMD (in columns), pairs of this lane's wave
        for (int c = 0; c < NCW; c++) ld += (sd == __builtin_amdgcn_readfirstlane(wsimd[c])) ? NB_PIPE_WALK_COLS : 0;
        for (int g = 0; 2 * g < N; g++) {
            unsigned m = (isab && ng < CPW / 2) ? (((unsigned)ld << 16) | ((unsigned)ng << 8) | (unsigned)w) : 0xffffffffu;
#pragma unroll
            for (int off = 32; off; off >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, off, 64));
            const int bw = (int)(m & 0xffu), bsd = __shfl(sd, bw, 64);
            if (w == bw) {
                colmap[w * CPW + 2 * ng] = (unsigned char)(2 * g);
                if (2 * g + 1 < N) colmap[w * CPW + 2 * ng + 1] = (unsigned char)(2 * g + 1);
                ng++;
            }
            ld += (sd == bsd) ? 2 : 0;
        }
    }
    __syncthreads();

    // The two kinds of wave run two separate loops with the same sequence of barriers (and the same slot bookkeeping, derived from
    // the same LDS words), so that the register allocation of the walk does not carry the A/S/B waves' long-lived arrays, and
    // vice versa.
    int frame0 = __builtin_amdgcn_readfirstlane(st[0]), frame1 = __builtin_amdgcn_readfirstlane(st[1]);
#ifdef NB_STAMP
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    long long nhalf = 0;
#define NB_PT(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define NB_PT(i)
#endif
    // ONE workgroup barrier per half-step, at the end of stage 1 (everyone needs everyone there: the walk needs every wave's stores
    // into `pairs`, the syndrome every wave's hard symbols).  The end of stage 2 is NOT a barrier: a sorting wave only has to know
    // that the WALK of this half-step is over before it overwrites `pairs` and reads E in the next stage 1, so the walking waves
    // count themselves off in st[5] (after their LDS operations have landed) and the sorting waves poll it.  The sorting waves of
    // the SIMD that hosts no walking wave finish their sorts ~5 k cycles before the others and start the next stage 1 -- where
    // that SIMD, with four of them, is the slow one -- that much earlier.
    if (is_c) {
        for (int h = 0;; h++) {
            const int s = h & 1, c = s ^ 1;
            NB_PT(0)
            __syncthreads(); // ---- end of stage 1: slot c's sorted pairs are in `pairs` (st[4]), E has been read by A(s)
            NB_PT(1)
            if (h > 0 && __builtin_amdgcn_readfirstlane(st[2 + c]) == h) { // slot c retired in the previous half-step: its next frame
                const int f = __builtin_amdgcn_readfirstlane(st[c]);
                if (c) frame1 = f; else frame0 = f;
            }
            if (frame0 >= a.B && frame1 >= a.B) break; // the counter only grows: both slots are past the batch
            if (NB_PIPE_PRIO == 2) __builtin_amdgcn_s_setprio(3);
            if (__builtin_amdgcn_readfirstlane(st[4]) && tid < TC) { // C(c): check nodes (:272-303)
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
            if (NB_PIPE_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            NB_PT(2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's reads of `pairs` and stores into E have landed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(&st[5], 1);
            NB_PT(3)
#ifdef NB_STAMP
            nhalf++;
#endif
        }
#ifdef NB_STAMP
        if (lane == 0 && blockIdx.x == 0 && a.LLR) { // every wave of workgroup 0: its intervals, half-steps, HW_ID (SIMD in bits 5:4)
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.LLR) + wave * 8;
            for (int i = 0; i < 6; i++) o[i] = tacc[i];
            o[6] = (unsigned long long)nhalf;
            o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
        }
#endif
        return;
    }

    const bool active = lane < q - 1;          // lanes 0..q-2 <-> field elements 1..q-1
    const int sym = active ? lane + 1 : 0;     // lane q-1 carries element 0 (value 0, :250)
    // per AB wave: its columns (column map above, 4 to a word), per column and edge the LDS offset of THIS lane's entry of the check thread's max
    // array, E[thr][mul(sym, h)] (the graph does not change: one read per iteration instead of three dependent look-ups)
    static_assert(CPW == 8, "two words of four column bytes");
    unsigned colw0 = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned *>(colmap + wave * CPW));
    unsigned colw1 = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned *>(colmap + wave * CPW + 4));
    auto colof = [](unsigned c0, unsigned c1, int ci) { return (int)(((ci < 4 ? c0 : c1) >> (8 * (ci & 3))) & 0xffu); };
    int evoff[CPW][DVM];
    unsigned wmask = 0, cmask = 0; // bit ci*DVM + d: edge d of column ci exists; bit ci: the column exists
#pragma unroll
    for (int ci = 0; ci < CPW; ci++) {
        const int colr = colof(colw0, colw1, ci), col = min(colr, N - 1);
        const int w = t_vn_w[col];
        if (colr < N) cmask |= 1u << ci;
#pragma unroll
        for (int d = 0; d < DVM; d++) {
            const int dd = min(d, dv - 1);
            evoff[ci][d] = t_vn_thr[col * dv + dd] * QP + mulb[sym * q + t_vn_gf[col * dv + dd]];
            if (d < w && colr < N) wmask |= 1u << (ci * DVM + d);
        }
    }
    // premultiplied symbols of THIS lane's field element on the wave's edges, one byte each (4 edges = one group of sorts per word)
    static_assert(SW == 4, "one word of symbol bytes per group");
    uint32_t pmw[NEW / SW];
#pragma unroll
    for (int g = 0; g < NEW / SW; g++) {
        pmw[g] = 0;
#pragma unroll
        for (int i = 0; i < SW; i++) {
            const int j = g * SW + i, ci = j / DVM, d = j % DVM;
            const int edge = min(colof(colw0, colw1, ci), N - 1) * dv + min(d, dv - 1);
            pmw[g] |= (uint32_t)mulb[sym * q + t_vn_gf[edge]] << (8 * i);
        }
    }
    wmask = __builtin_amdgcn_readfirstlane(wmask); // wave-uniform by construction; the compiler cannot know (they come out of LDS):
    cmask = __builtin_amdgcn_readfirstlane(cmask); // as scalars, the tests below are scalar branches instead of EXEC masking
    int it0 = 0, it1 = 0;
    bool sorted0 = false, sorted1 = false; // the slot's sorted pairs sit in sval / ssym
    float sval[NEW];                       // sorted values of this wave's edges (lane <-> sorted position)
    unsigned ssym[NEW / 4];                // their premultiplied symbols mul(symbol, h), one byte each
    const uint32_t kmw = nb_keepmax_word(lane);
    // channel vectors of the columns of this wave for the slot whose phase A comes next: loaded one half-step ahead (global
    // memory, ~1 us away), so that the loads fly during the sorts of the other slot
    float lch[CPW];
    // (lane q-1 loads its neighbour's entry: every use of its lch / llr below is masked by `active`.  Scalar base per column + one
    // constant lane offset: no vector address arithmetic; the column words come in as parameters so that the column bases are recomputed on
    // the scalar unit every half-step instead of being hoisted out of the loop and spilled)
    unsigned lch_lo = (unsigned)min(lane, q - 2);
    auto load_lch = [&](int frame, unsigned c0, unsigned c1) {
        const float *Lch = a.Lch + (size_t)min(frame, a.B - 1) * N * (q - 1);
        unsigned lo = lch_lo;
        asm volatile("" : "+v"(lo)); // opaque: the loads are issued here
#pragma unroll
        for (int ci = 0; ci < CPW; ci++) {
            const float *p = Lch + min(colof(c0, c1, ci), N - 1) * (q - 1);
            lch[ci] = p[lo];
        }
    };
    load_lch(frame0, colw0, colw1);

    const int abw_fixed = abw;
    const unsigned wmask_fixed = wmask, cmask_fixed = cmask, colw0_fixed = colw0, colw1_fixed = colw1;
    for (int h = 0;; h++) {
        const int s = h & 1;
        const int frame_s = s ? frame1 : frame0;
        const bool act_s = frame_s < a.B;
        int abw = abw_fixed; // opaque per half-step: left alone the compiler hoists every column's addresses out of the loop and spills ~150 scalars
        asm volatile("" : "+s"(abw));
        unsigned wmask = wmask_fixed, cmask = cmask_fixed; // likewise: one scalar bit test per use instead of 24 lane masks held (and spilled) across the loop
        asm volatile("" : "+s"(wmask), "+s"(cmask));
        unsigned colw0 = colw0_fixed, colw1 = colw1_fixed;
        asm volatile("" : "+s"(colw0), "+s"(colw1));
        int *outs = outs0 + (s ? (int)(outs1 - outs0) : 0);
        float v2c[NEW];
        // the walk of the previous half-step must be over before this wave touches `pairs` / E again (see the walking waves' loop)
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < NCW * h) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        NB_PT(0)
        // ---- stage 1a: everything that touches `pairs` and E -----------------------------------------------------------------
        float evs[NEW]; // the max-array entries of this wave's columns (read before the barrier: the walk behind it overwrites E)
        bool fresh = false;
        {
            const bool have = s ? sorted0 : sorted1; // slot c was sorted in the previous half-step
            if (have) {
#pragma unroll
                for (int j = 0; j < NEW; j++) {
                    const int ci = j / DVM, d = j % DVM;
                    if ((wmask >> j) & 1u) {
                        float2 pr;
                        pr.x = sval[j];
                        pr.y = __int_as_float((int)((ssym[j >> 2] >> (8 * (j & 3))) & 0xffu) << 2); // byte offset into the thread's max array
                        *reinterpret_cast<float2 *>(pairs + (colof(colw0, colw1, ci) * dv + d) * PST + 2 * lane) = pr;
                    }
                }
            }
            if (abw == 0 && lane == 0) st[4] = have ? 1 : 0;
            if (s) sorted0 = false; else sorted1 = false;
            if (act_s) {
                const int it = (s ? it1 : it0) + 1;
                if (s) it1 = it; else it0 = it;
                fresh = it == 1; // L_c2v = 0 (:185-193): (0-0)/1.2 == +0
                // all the entries in one batch of reads (a frame's first iteration: L_c2v = 0, made by reading nothing --
                // (0 - 0) / 1.2 == +0 goes through the same arithmetic, no branch)
#pragma unroll
                for (int j = 0; j < NEW; j++) evs[j] = E[evoff[j / DVM][j % DVM]];
#pragma unroll
                for (int j = 0; j < NEW; j++) evs[j] = fresh ? 0.0f : evs[j];
            }
        }
        NB_PT(1)
        __syncthreads(); // ---- slot c's sorted pairs are in `pairs` (st[4]), E(s) has been read: the walking waves start C(c) here
        NB_PT(2)
        if (h > 0 && __builtin_amdgcn_readfirstlane(st[2 + (s ^ 1)]) == h) { // slot c retired in the previous half-step: its next frame
            const int f = __builtin_amdgcn_readfirstlane(st[s ^ 1]);
            if (s) frame0 = f; else frame1 = f;
        }
        const int frame_c = s ? frame0 : frame1;
        if (frame_s >= a.B && frame_c >= a.B) break; // the counter only grows: both slots are past the batch
        if (NB_PIPE_PRIO == 1) __builtin_amdgcn_s_setprio(3);
        // ---- stage 1b: A(s) on the values read above, beside the walk -----------------------------------------------------------
        if (act_s) {
            float *LLRo = a.LLR ? a.LLR + (size_t)frame_s * N * (q - 1) : nullptr;
#pragma unroll
            for (int ci = 0; ci < CPW; ci++) {
                if (!((cmask >> ci) & 1u)) { // no such column (scalar branch): its sorts are skipped as well
#pragma unroll
                    for (int d = 0; d < DVM; d++) v2c[ci * DVM + d] = 0.0f;
                    continue;
                }
                const int col = colof(colw0, colw1, ci);
                float llr = lch[ci];
                float c2[DVM];
#pragma unroll
                for (int d = 0; d < DVM; d++) {
                    const float ev = evs[ci * DVM + d];
                    const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), Q - 1));
                    c2[d] = nb_div12(ev - e0); // :309, double division
                }
                constexpr unsigned FULL = (1u << DVM) - 1u;
                if (((wmask >> (ci * DVM)) & FULL) == FULL) { // every edge of the column exists (scalar branch): no selects
#pragma unroll
                    for (int d = 0; d < DVM; d++) llr = llr + c2[d]; // :208-213, ascending d
                } else {
#pragma unroll
                    for (int d = 0; d < DVM; d++) {
                        const bool on = (wmask >> (ci * DVM + d)) & 1u;
                        c2[d] = on ? c2[d] : 0.0f;
                        llr = on ? llr + c2[d] : llr;
                    }
                }
                // DecideLLRVector (:71-91): running max from 0, strict >, first maximum wins
                const float v = active ? llr : -__builtin_inff();
                const float mx = nb_wave_max(v);
                const unsigned long long eq = __builtin_amdgcn_ballot_w64(v == mx) & ((1ull << (Q - 1)) - 1ull); // (the active lanes, as a constant)
                const int dec = (mx > 0.0f) ? (int)__builtin_ctzll(eq) + 1 : 0;
                if (lane == 0) outs[col] = dec;
                if (LLRo && active) LLRo[col * (q - 1) + lane] = llr;
#pragma unroll
                for (int d = 0; d < DVM; d++) v2c[ci * DVM + d] = active ? llr - c2[d] : 0.0f; // :241-251 (element 0: value 0)
            }
        }
        // The syndrome (:218-238) needs every sorting wave's hard symbols: the waves count themselves off in st[6] and poll (the
        // walking waves are busy and must not be part of this: no workgroup barrier); then every wave computes it for itself, lane <->
        // check row, 64 rows per round.  (NB_PIPE_ONE_S: only the LAST wave to arrive computes it, publishes it in st[7] and opens
        // st[8] -- 12 syndromes fewer per half-step, but behind one another on the critical path: 6 % slower.)
        // The first rows' terms are constants of the graph: requested before the wait.
        const unsigned *rt = rowtab + s * M * RW;
        const uint4 w4first = *reinterpret_cast<const uint4 *>(rt + min(lane, M - 1) * RW);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int arrived = 0;
        bool bad_own = false;
        if (lane == 0) arrived = atomicAdd(&st[6], 1);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        NB_PT(3)
        if (NB_PIPE_ONE_S == 0 || arrived == NAB * (h + 1) - 1) {
            if (NB_PIPE_ONE_S == 0) while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < NAB * (h + 1)) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool bad = false;
            if (act_s) {
                for (int r0 = 0; r0 < M; r0 += 64) {
                    const int r = r0 + lane;
                    int sy = 0;
                    for (int k0 = 0; k0 < RW; k0 += 4) { // (one round for rows of weight <= 4; two LDS round trips per round)
                        const uint4 w4 = (r0 == 0 && k0 == 0) ? w4first : *reinterpret_cast<const uint4 *>(rt + min(r, M - 1) * RW + k0);
                        const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
                        int hs[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) hs[k] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(lds) + (w[k] & 0x3ffffu));
#pragma unroll
                        for (int k = 0; k < 4; k++) sy ^= mulb[hs[k] * q + (int)(w[k] >> 18)];
                    }
                    bad = bad || (__builtin_amdgcn_ballot_w64(sy != 0 && r < M) != 0ull);
                }
            }
            if (NB_PIPE_ONE_S == 0) bad_own = bad;
            else if (lane == 0) {
                st[7] = bad ? 1 : 0;
                __hip_atomic_store(&st[8], h + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (NB_PIPE_ONE_S) {
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < h + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const bool bad = NB_PIPE_ONE_S ? (__builtin_amdgcn_readfirstlane(st[7]) != 0) : bad_own;
        NB_PT(4)
        // ---- stage 2 ----------------------------------------------------------------------------------------------------
        if (!NB_PIPE_LATE_LCH) load_lch(frame_c, colw0, colw1); // phase A of the next half-step is slot c's
        if (act_s) {
            const int it = s ? it1 : it0;
            if (!bad || it == a.max_iter) { // the frame leaves: zero syndrome (:232-238, iter_number-- first) or maxIT iterations used
                for (int i = abw * 64 + lane; i < N; i += NAB * 64) a.out[(size_t)frame_s * N + i] = outs[i];
                if (abw == 0 && lane == 0) {
                    a.iters[frame_s] = bad ? it : it - 1;
                    a.ok[frame_s] = bad ? 0 : 1;
                    st[s] = atomicAdd(a.work, 1);
                    st[2 + s] = h + 1; // read by every wave after the next barrier; the next write is two barriers away
                }
                if (s) it1 = 0; else it0 = 0;
            } else {
                // B: stable descending sort of every v2c vector (:17-36, :253-269), as in k_nb_ems but from and to registers
#pragma unroll
                for (int g = 0; g < NEW / SW; g++) {
                    if (((wmask >> (g * SW)) & ((1u << SW) - 1u)) == 0u) continue; // none of these edges exists (scalar branch)
                    uint32_t khi[SW], k32[SW];
#pragma unroll
                    for (int i = 0; i < SW; i++) {
                        const uint32_t b = __float_as_uint(v2c[g * SW + i] + 0.0f); // +0.0f folds -0 onto +0 (they compare equal)
                        const uint32_t sg = (uint32_t)((int)b >> 31);               // order-preserving image: ~b below zero, b | sign bit above
                        khi[i] = (b ^ sg) | (~sg & 0x80000000u);
                        k32[i] = (khi[i] & 0xffffffc0u) | (63u - (unsigned)lane);
                    }
                    uint32_t kmw_cold = kmw;
                    asm volatile("" : "+v"(kmw_cold));
                    nb_bitonic_sort32<64, NB_PIPE_COLD_K>(k32, kmw, kmw_cold);
                    int idx[SW];
                    // two neighbours of the sorted order agree in the 26 bits the short keys carry?  key ^ next lane's key in one
                    // instruction each (DPP operand; lane 63 has no neighbour and is masked out of the ballots)
                    bool amb;
                    {
                        uint32_t x0, x1, x2, x3;
                        asm("s_nop 1\n\t"
                            "v_xor_b32_dpp %0, %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_xor_b32_dpp %1, %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_xor_b32_dpp %2, %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_xor_b32_dpp %3, %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
                            : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(k32[0]), "v"(k32[1]), "v"(k32[2]), "v"(k32[3]));
                        static_assert(SW == 4, "four sorts in flight");
                        const unsigned long long am = __builtin_amdgcn_ballot_w64(x0 < 64u) | __builtin_amdgcn_ballot_w64(x1 < 64u) |
                                                      __builtin_amdgcn_ballot_w64(x2 < 64u) | __builtin_amdgcn_ballot_w64(x3 < 64u);
                        amb = (am & 0x7fffffffffffffffull) != 0ull;
                    }
#pragma unroll
                    for (int i = 0; i < SW; i++) idx[i] = (int)(~k32[i] & 63u); // the key's low bits are 63 - (original position)
                    // No such pair: all 64 short keys differ above their index bits, so their order IS the order of the full
                    // (value, index) keys.  Otherwise verify the permutation against the full keys, as k_nb_ems does.
                    if (amb) {
                        uint32_t img[SW];
                        bool redo = false;
#pragma unroll
                        for (int i = 0; i < SW; i++) img[i] = (uint32_t)__shfl((int)khi[i], idx[i], 64);
#pragma unroll
                        for (int i = 0; i < SW; i++) {
                            const uint32_t nimg = (uint32_t)__builtin_amdgcn_update_dpp((int)img[i], (int)img[i], 0x130, 0xf, 0xf, false);
                            const int nidx = __builtin_amdgcn_update_dpp(idx[i], idx[i], 0x130, 0xf, 0xf, false);
                            const bool in_order = img[i] > nimg || (img[i] == nimg && idx[i] < nidx);
                            redo = redo || (__builtin_amdgcn_ballot_w64(!in_order && lane < 63) != 0ull);
                        }
                        if (redo) { // two values that differ only in their low 6 bits: this group again, on the full keys (wave-uniform branch)
                            uint32_t klo[SW];
#pragma unroll
                            for (int i = 0; i < SW; i++) klo[i] = 63u - (unsigned)lane;
                            nb_bitonic_sort<64, SW>(khi, klo);
#pragma unroll
                            for (int i = 0; i < SW; i++) idx[i] = 63 - (int)klo[i];
                        }
                    }
                    // idx: original position of the element that belongs at position `lane`.  Its value and its premultiplied symbol
                    // (GFMultiply(sort_Entr_v2c, linkVNs_GF) of :334; a constant of the original lane, pmw) come over with the same
                    // permutation; byte i of the i-th permuted word is the i-th edge's
                    uint32_t r[SW];
#pragma unroll
                    for (int i = 0; i < SW; i++) {
                        const int a4 = idx[i] << 2;
                        sval[g * SW + i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a4, __builtin_bit_cast(int, v2c[g * SW + i])));
                        r[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)pmw[g]);
                    }
                    const uint32_t t1 = (r[0] & 0xffu) | (r[1] & ~0xffu), t2 = (t1 & 0xffffu) | (r[2] & ~0xffffu);
                    ssym[g] = (t2 & 0xffffffu) | (r[3] & ~0xffffffu);
                }
                if (s) sorted1 = true; else sorted0 = true;
            }
        }
        // phase A of the next half-step is slot c's: its channel values are requested here, AFTER the sorts (8 registers the sorts
        // need), and fly while this wave waits for the walk, stores its pairs and stands at the barrier
        if (NB_PIPE_LATE_LCH) load_lch(frame_c, colw0, colw1);
        if (NB_PIPE_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        NB_PT(5)
#ifdef NB_STAMP
        nhalf++;
#endif
    }
#ifdef NB_STAMP
    if (lane == 0 && blockIdx.x == 0 && a.LLR) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.LLR) + wave * 8;
        for (int i = 0; i < 6; i++) o[i] = tacc[i];
        o[6] = (unsigned long long)nhalf;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
    }
#endif
}

} // namespace cldpc
