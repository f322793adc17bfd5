// bldpc_qcr2_kernel.hpp -- k_qcr2: the register-state kernel of bldpc_qcr_kernel.hpp (same schedule, same arithmetic, same
// bits) without the address arithmetic.  k_qcr spends four vector instructions per edge and phase on `(t + shift) mod Z` and
// the column base, twice per edge and iteration (keeping a thread's 230 addresses in registers spills), a quarter of the
// instructions of a kernel that the vector unit bounds.
//
// Lanes run along the circulant; a wave covers 64 consecutive positions (a tile), so through a cyclic shift it meets
// positions rb .. rb + 63 with rb = (64 tile + shift) mod Z: consecutive, except where they wrap past Z.  Every column of S
// therefore carries a HALO: words Z .. Z + 63 repeat words 0 .. 63 (4 (Z + 64) bytes per column; J15_L30_Z1280: 157.5 KB of
// the CU's 160 KB), a read at rb + lane is always right, and the address of an edge is ONE addition: 4 * lane + the
// wave-uniform byte offset of (column, rb), which the host tabulates per (block row, tile, slot) and the wave fetches with
// scalar loads.  Phase 2 uses the same address for its read and its write, and it works on the home words only.  Exactly one
// tile of every block wraps (its 64 positions straddle Z: 5 % of the (block, tile) pairs); the order of a row's slots is free
// (min1 / min2 / sign product are symmetric, a variable is met once per block row), so the host orders the slots PER
// (block row, tile) and puts the wrapped block, if any, into one of the last two slots, whose addresses take three more
// instructions (lanes from k on lie 4 Z bytes lower, k = 64 when nothing wraps).  J15_L30_Z1280 has at most two wrapped
// blocks per (row, tile); random shifts of the same shape exceed two in two matrices out of three and three in one out of
// sixteen: NG = 2 and NG = 3 are both instantiated, the host picks the first that fits and k_qcr beyond.  The halos are refreshed once per
// iteration, by the closing pass that adds the channel values, before phase 1 reads through them.  The code stays
// straight-line and short: the loop body is 75 KB of instructions streamed through a 64 KB cache every iteration; a first
// version with a wave-uniform branch per wrapped block into out-of-line code lost 20 % to it, and even the never-executed
// out-of-line loop for a third wrapped block cost 15 % by its size alone.
// Two more things keep it free of special cases:
//   * a column's first edge is `0 + R_0` (LDPC_Decoder.cu:188-204): S is zeroed between the phases (16-byte stores, one
//     more barrier), so that every edge is the same `S + R`;
//   * the slot of a light row (padding) addresses 64 words holding +inf: phase 1 sees Q = inf - R = +inf (neutral for
//     min1 / min2, never the first minimum, sign 0), phase 2 stores inf + R = +inf back.
// Tried first and dropped (tools/lds_atomic_probe.hip, profiles/r02_lds_atomic_probe.txt): ds_read_addtid_b32 /
// ds_write_addtid_b32, whose address is M0 + 4 * lane with no vector instruction at all -- bit-identical, and slower
// (123 k codewords/s against 175 k): three scalar instructions per access (s_mov m0, the wait state, the access) and the
// hand-counted waits cost a wave more issue slots than the additions they replace; ds_add_f32, which would fold read + add
// + write into one instruction with the same bits as v_add_f32, retires one lane every 3 cycles (192 cycles per wave).
// (no includes, no namespace: this file is spliced into namespace cldpc by bldpc_qc_kernel.hpp, after bldpc_qcr_kernel.hpp)

#ifndef QCR2_ABLATE
#define QCR2_ABLATE 0 // experiments only (timing, wrong results): 1 no zeroing pass, 2 nor its barrier, 4 no wrapped blocks, 8 no phase 2, 16 no phase 1
#endif

template <int J_, int L_, int Z_, int TPB_, int WCS_, int YB_, int NG_> struct Qcr2Geom {
    static constexpr int J = J_, L = L_, Z = Z_, TPB = TPB_, WCS = WCS_, ZR = (Z + TPB - 1) / TPB, YB = YB_;
    static constexpr int NG = NG_;           // slots with per-lane addresses in phase 2 (the last NG of a row): wrapped blocks per (row, tile) it takes
    static constexpr int NT = Z / 64;        // tiles of the circulant
    static constexpr int ZH = Z + 64;        // words per column of S, halo included
    static constexpr int S_BYTES = L * ZH * 4;
    static constexpr int INF_BASE = S_BYTES, FLAG_BASE = S_BYTES + 256, lds_bytes = FLAG_BASE + 16;
    static constexpr bool RAGGED = (Z % TPB) != 0;
    static_assert((L * Z) % (TPB * YB) == 0, "the closing pass runs in whole batches");
    static_assert(Z % 64 == 0 && TPB % 64 == 0 && TPB <= 1024, "threads must tile the circulant in whole waves");
    static_assert(NG >= 1 && WCS > NG && WCS <= 27 && S_BYTES < (1 << 18), "sign bits and the 5-bit index share one word; slot 0 is the register-resident column");
    static_assert(lds_bytes <= 160 * 1024, "S with its halos must fit one CU's LDS");
};


template <typename GM, bool HIST, int NZ>
__device__ __forceinline__ void qcr2_iterations(const QcArgs &a, char *lds, int *lds_flag, const __amdgpu_buffer_rsrc_t yrs, float (&S0)[GM::ZR],
                                                unsigned long long &hist, int &stop)
{
    constexpr int J = GM::J, Z = GM::Z, TPB = GM::TPB, WCS = GM::WCS, N = GM::L * Z, NS = N / TPB, YB = GM::YB, NT = GM::NT, ZH = GM::ZH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // [J][NT][WCS]: byte offset of lane 0's word for every (block row, tile, slot), ONE table for both phases (9.6 KB for J15_L30_Z1280: it
    // has to stay in the 16 KB scalar cache -- two tables of this size did not, and every block row then waited for an L2 round trip:
    // 5.07 -> 4.5x ms); the last NG slots carry their first wrapped lane in bits 18..24, which phase 1 (reads through the halo) masks off
    const qcr_const_u32 *tx = (const qcr_const_u32 *)a.vn_meta;
    const int lcbase = a.lc * Z;
    auto y_at = [&](int stride_idx) -> float { // y[tid + stride_idx * TPB]
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, tid * 4, stride_idx * TPB * 4, 0));
    };
    auto y_lc = [&](int z) -> float { // y[LC*Z + tid + z * TPB]
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, tid * 4, (lcbase + z * TPB) * 4, 0));
    };
    auto recon = [&](float a1, float a2, unsigned ww, int p) -> float {
        const float mag = ((int)(ww >> 27) == p) ? a2 : a1;
#ifdef QCR2_AND_OR /* experiment: the sign insert as the compiler's v_lshlrev + v_and_or_b32 (half rate) */
        return u2f(f2u(mag) | ((ww >> (WCS - 1 - p)) << 31));
#else
        return u2f(__builtin_amdgcn_bitop3_b32(ww << (31 - (WCS - 1 - p)), 0x80000000u, f2u(mag), 0xEA)); // (sh & sign) | mag: v_bitop3_b32 issues at the full rate
#endif
    };
    auto flags_collect = [&](int it) { // two flag words, used by odd and even iterations in turn (see k_qc)
        if (tid == 0) {
            const int flag = lds_flag[it & 1] ? 0 : 1;
            lds_flag[(it + 1) & 1] = 0;
            if (flag && it <= 64) hist |= (1ull << (it - 1));
        }
    };

    float m1[J][NZ], m2[J][NZ]; // check states: R_p = +-m1, +-m2 on edge idx
    unsigned w2[J][NZ];         // idx << 27 | output sign bits (edge p at bit WCS-1-p)
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int z = 0; z < NZ; z++) {
            m1[j][z] = 0.0f; m2[j][z] = 0.0f; w2[j][z] = 0u; // every R starts as +0 (LDPC_Decoder.cu:82)
        }

    int la = lane * 4;
    for (int it = 1; it < a.max_iter; it++) {
        // fresh (opaque) table pointers per phase, see k_qcr; the lane offset too: left alone the compiler forms `la + constant` for
        // every column and stride of the closing pass ahead of the loop and keeps 30 registers of them
        asm volatile("" : "+s"(tx), "+v"(la));
        // ---- phase 1: check nodes on S of iteration `it` (LDPC_Decoder.cu:279-314).  The offsets of block row j + 1 are
        // requested while row j is computed: the tables (29 KB) do not stay in the scalar cache, a load costs an L2 round trip.
        unsigned tn[NZ][WCS];
#pragma unroll
        for (int z = 0; z < NZ; z++)
#pragma unroll
            for (int p = 1; p < WCS; p++) tn[z][p] = tx[(0 * NT + wv + z * (TPB / 64)) * WCS + p];
#pragma unroll
        for (int j = 0; j < ((QCR2_ABLATE & 16) ? 0 : J); j++) {
            float Sv[NZ][WCS];
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                Sv[z][0] = S0[z];
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    float sv[1];
                    lds_ld<1>(sv, lds, (int)(p >= WCS - GM::NG ? tn[z][p] & 0x3ffffu : tn[z][p]) + la); // (a per-lane slot's word carries its wrap lane: phase 2)
                    Sv[z][p] = sv[0];
                }
            }
            if (j + 1 < J) {
#pragma unroll
                for (int z = 0; z < NZ; z++)
#pragma unroll
                    for (int p = 1; p < WCS; p++) tn[z][p] = tx[((j + 1) * NT + wv + z * (TPB / 64)) * WCS + p];
            }
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                float n1 = __builtin_inff(), n2 = __builtin_inff();
                unsigned signs = 0, lower = 0;
#pragma unroll
                for (int p = 0; p < WCS; p++) {
                    const float q = Sv[z][p] - recon(m1[j][z], m2[j][z], w2[j][z], p); // Q = S - R (:206-209); a pad slot: +inf
                    const float aq = __builtin_fabsf(q);
                    // "this edge lowered the running minimum" as a bit per edge: the sign of |q| - n1 (equal values give +0, inf - inf the
                    // positive default NaN), two full-rate instructions where compare + v_cndmask_b32 into an index are two of the half-rate class
                    lower = __builtin_amdgcn_alignbit(lower, f2u(aq - n1), 31);
                    n2 = __builtin_amdgcn_fmed3f(n1, n2, aq);
                    n1 = __builtin_fminf(n1, aq);
                    signs = __builtin_amdgcn_alignbit(signs, f2u(q), 31); // (signs << 1) | sign(q)
                }
                // first edge holding the minimum (:298-305) = the LAST edge that lowered it = the lowest set bit (edge p at bit WCS-1-p;
                // no edge below +inf: edge 0, as the index form starts from)
                const int idx = WCS - 1 - (int)__builtin_ctz(lower | (1u << (WCS - 1)));
                // R_p = Sign[25]*Sign[p] * magnitude: output sign bit p = parity of all signs XOR sign p
                if (__builtin_popcount(signs) & 1) signs ^= (1u << WCS) - 1u;
                m1[j][z] = n1; m2[j][z] = n2;
                w2[j][z] = signs | ((unsigned)idx << 27);
                asm volatile("" : "+v"(m1[j][z]), "+v"(m2[j][z]), "+v"(w2[j][z])); // the state is complete here (see k_qcr)
            }
            __builtin_amdgcn_sched_barrier(0); // one block row's reads in flight at a time: bounds the VGPRs
        }
        asm volatile("" : "+s"(tx));
#pragma unroll
        for (int z = 0; z < NZ; z++) // phase 2's first block row: in flight across the zeroing pass
#pragma unroll
            for (int p = 0; p < WCS; p++) tn[z][p] = tx[(0 * NT + wv + z * (TPB / 64)) * WCS + p];
        __syncthreads();

        // ---- between the phases: S = 0, halos included (phase 1 is done with it), so that a column's first edge is an addition like the others
        {
            typedef float qcr2_v4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) qcr2_v4 lds_v4;
            const qcr2_v4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!(QCR2_ABLATE & 1))
                for (int i = tid; i < GM::S_BYTES / 16; i += TPB) *reinterpret_cast<lds_v4 *>(static_cast<unsigned>(i * 16)) = z4;
        }
        if (!(QCR2_ABLATE & 2)) __syncthreads();
        // ---- phase 2: S of iteration it+1 = ((0 + R_0) + R_1 + ...), block rows in ascending order (:188-204) ----
        float acc0[NZ], yv[YB];
#pragma unroll
        for (int z = 0; z < NZ; z++) acc0[z] = 0.0f;
#pragma unroll
        for (int j = 0; j < ((QCR2_ABLATE & 8) ? 0 : J); j++) {
            if (j + 1 == J) { // first batch of the closing pass
#pragma unroll
                for (int i = 0; i < YB; i++) yv[i] = y_at(i);
            }
            unsigned tc[NZ][WCS];
#pragma unroll
            for (int z = 0; z < NZ; z++)
#pragma unroll
                for (int p = 0; p < WCS; p++) tc[z][p] = tn[z][p];
            if (j + 1 < J) { // the next block row's offsets: in flight across this one's barrier
#pragma unroll
                for (int z = 0; z < NZ; z++)
#pragma unroll
                    for (int p = 0; p < WCS; p++) tn[z][p] = tx[((j + 1) * NT + wv + z * (TPB / 64)) * WCS + p];
            }
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                int va[WCS];
                float prev[WCS];
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    if (p >= WCS - GM::NG) { // the row's wrapped blocks for this tile, if any, sit in the last NG slots: lanes from k on
                        const unsigned w = (QCR2_ABLATE & 4) ? (tc[z][p] & 0x3ffffu) | (64u << 18) : tc[z][p]; // lie 4 Z lower (k in bits 18..24, 64: none)
                        va[p] = (int)(w & 0x3ffffu) + la - ((lane >= (int)(w >> 18)) ? 4 * Z : 0);
                    } else {
                        va[p] = (int)tc[z][p] + la;
                    }
                    float sv[1];
                    lds_ld<1>(sv, lds, va[p]);
                    prev[p] = sv[0];
                }
                acc0[z] += recon(m1[j][z], m2[j][z], w2[j][z], 0); // slot 0 = column LC: this thread's own variable
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    const float sv[1] = {prev[p] + recon(m1[j][z], m2[j][z], w2[j][z], p)};
                    lds_st<1>(lds, va[p], sv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        // ---- ... + y closes every sum (:205): one aligned pass over the tiles, channel values re-read (L2-resident).  The pass
        // also sweeps column LC's unused words rather than special-casing them.  Flag bookkeeping as in k_qcr.
        unsigned badbits = 0;
        int lim = a.length - tid, lcrel = lcbase - tid;
        if (HIST) asm volatile("" : "+v"(lim), "+v"(lcrel));
        int wvo = wv; // opaque per iteration: left alone the compiler computes the 50 addresses of this pass once, ahead of the loop, and spills
        asm volatile("" : "+s"(wvo), "+v"(la));
#pragma unroll
        for (int z = 0; z < NZ; z++) {
            S0[z] = acc0[z] + y_lc(z);
            if (HIST) badbits |= f2u(S0[z]) & (unsigned)((lcbase + z * TPB - lim) >> 31);
        }
#pragma unroll
        for (int i0 = 0; i0 < NS; i0 += YB) {
            float yn[YB], sv[YB][1];
            if (i0 + YB < NS) {
#pragma unroll
                for (int i = 0; i < YB; i++) yn[i] = y_at(i0 + YB + i); // next batch
            }
            int mo[YB];
            bool halo[YB];
#pragma unroll
            for (int i = 0; i < YB; i++) { // this wave's tile of stride i0 + i is tile g = (i0 + i) * (TPB / 64) + wave of the frame: column g / NT
                constexpr int W = TPB / 64;
                static_assert(W <= NT, "a stride of tiles crosses at most one column boundary");
                const int q0 = ((i0 + i) * W) / NT, r0 = ((i0 + i) * W) % NT; // compile-time: column and tile of wave 0's tile
                const int t0 = wvo + r0;                                      // < 2 NT
                const bool next = t0 >= NT;                                   // this wave's tile lies in the next column: skip the halo
                mo[i] = ((q0 * ZH + r0 * 64) * 4 + (next ? 256 : 0)) + wvo * 256 + la;
                halo[i] = r0 == 0 ? (wvo == 0) : (t0 == NT);                  // tile 0 of its column
                lds_ld<1>(sv[i], lds, mo[i]);
            }
#pragma unroll
            for (int i = 0; i < YB; i++) {
                sv[i][0] += yv[i];
                lds_st<1>(lds, mo[i], sv[i]);
                if (halo[i]) lds_st<1>(lds, mo[i] + 4 * Z, sv[i]); // positions 0 .. 63 of a column: the halo repeats them
                if (HIST) { // v < length and v outside [lcbase, lcbase + Z), with v = tid + (i0 + i) * TPB
                    const int k = (i0 + i) * TPB;
                    const unsigned in_len = (unsigned)((k - lim) >> 31), in_lc = ((unsigned)(k - lcrel) < (unsigned)Z) ? ~0u : 0u;
                    badbits |= f2u(sv[i][0]) & in_len & ~in_lc;
                }
            }
            if (i0 + YB < NS) {
#pragma unroll
                for (int i = 0; i < YB; i++) yv[i] = yn[i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (HIST && (badbits >> 31)) lds_flag[(it + 1) & 1] = 1;
        __syncthreads();
        if (HIST && it + 1 < a.max_iter) {
            flags_collect(it + 1);
            if (a.per_frame && !lds_flag[(it + 1) & 1]) { // per-frame exit: the frame stops with S of iteration it+1
                stop = it + 1;
                break;
            }
        }
    }
}

// LDS: S float[L][Z + 64] | 64 x +inf | flags.
// PERSIST (per-frame exit only): the grid fills the chip once and a workgroup takes frame after frame of its XCD from the counter
// a.work[xcd] -- frames leave after 1 ... max_iter iterations, and one workgroup per frame leaves the CUs far apart (see k_qc2p)
template <typename GM, bool HIST, bool PERSIST = false> __global__ __launch_bounds__(GM::TPB) void k_qcr2(QcArgs a)
{
    constexpr int Z = GM::Z, TPB = GM::TPB, ZR = GM::ZR, N = GM::L * Z;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int chunk = (a.nWG + 7) >> 3; // XCD-aware workgroup id, see k_qc
    int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (!PERSIST && wg >= a.nWG) return;
    const int F = a.F;
    const int tid = threadIdx.x;
    int *lds_flag = reinterpret_cast<int *>(lds + GM::FLAG_BASE);
    const bool zlast = !GM::RAGGED || (tid + (ZR - 1) * TPB < Z); // wave-uniform: this wave covers the last tile too
    auto s_byte = [&](int v) -> int { return (v + (v / Z) * 64) * 4; }; // (column * (Z + 64) + position) * 4
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
    const float *yf = a.y + (size_t)f * N;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(yf), 0, N * 4, 0x00020000);

    int bad = 0;
    for (int v = tid; v < N; v += TPB) { // iteration 1: S = (0 + 0 + ...) + y
        const float sv[1] = {0.0f + yf[v]};
        lds_st<1>(lds, s_byte(v), sv);
        if (v % Z < 64) lds_st<1>(lds, s_byte(v) + 4 * Z, sv);
        if (HIST) bad |= (int)(v < a.length) & (int)(sv[0] < 0);
    }
    if (tid < 64) { // what a light row's padding slot reads and writes back: +inf
        const float cv[1] = {__builtin_inff()};
        lds_st<1>(lds, GM::INF_BASE + tid * 4, cv);
    }
    float S0[ZR];
#pragma unroll
    for (int z = 0; z < ZR; z++) S0[z] = (z < ZR - 1 || zlast) ? 0.0f + yf[a.lc * Z + tid + z * TPB] : 0.0f;
    if (tid == 0) lds_flag[0] = lds_flag[1] = 0;
    unsigned long long hist = 0;
    int stop = 0; // per-frame exit: the iteration at which the frame's flag came up (workgroup-uniform)
    __syncthreads();
    if (HIST) {
        if (bad) lds_flag[1] = 1; // iteration 1
        __syncthreads();
        if (a.max_iter > 1) {
            if (tid == 0 && !lds_flag[1]) hist |= 1ull;
            if (a.per_frame && !lds_flag[1]) stop = 1;
        }
    }

    if (stop) {
    } else if (GM::RAGGED && !zlast) qcr2_iterations<GM, HIST, (GM::RAGGED ? ZR - 1 : ZR)>(a, lds, lds_flag, yrs, S0, hist, stop);
    else qcr2_iterations<GM, HIST, ZR>(a, lds, lds_flag, yrs, S0, hist, stop);
    const int last = stop ? stop : a.max_iter;

    // ---- outputs from S of the last iteration ----
#pragma unroll
    for (int z = 0; z < ZR; z++)
        if (z < ZR - 1 || zlast) { // column LC comes back from the registers
            const float sv[1] = {S0[z]};
            lds_st<1>(lds, s_byte(a.lc * Z + tid + z * TPB), sv);
        }
    __syncthreads();
    bad = 0;
    for (int n = tid; n < N; n += TPB) {
        float sv[1];
        lds_ld<1>(sv, lds, s_byte(n));
        const bool neg = sv[0] < 0;
        bad |= (int)(n < a.length) & (int)neg;
        const unsigned long long m = __ballot(neg); // one 32-bit word per half-wave (TPB % 64 == 0)
        if ((tid & 31) == 0) a.bits[(size_t)f * (N / 32) + (n >> 5)] = (unsigned)(m >> (tid & 32));
        if (a.app) a.app[(size_t)n * F + f] = sv[0];
    }
    if (bad) lds_flag[last & 1] = 1; // (HIST: the last round has already published the same verdict)
    __syncthreads();
    if (tid == 0) {
        const int flag = lds_flag[last & 1] ? 0 : 1;
        if (flag && last <= 64) hist |= (1ull << (last - 1));
        a.D[(size_t)N * F + f] = flag;
        if (HIST && a.hist) a.hist[f] = hist;
        if (HIST && a.per_frame) a.iters[f] = last;
    }
    if (!PERSIST) break;
    __syncthreads(); // S, the flags and the frame word are reused by the next frame
    } // next frame
}
