// bldpc_qcr_kernel.hpp -- fused QC min-sum kernel for LONG blocks (BLDPC_KERNEL_QC_LDS, third tier):
// codes whose compressed check states (12 M bytes) do not fit LDS next to the a-posteriori values either,
// i.e. BASELINE config 4, J15_L30_Z1280 (N = 38 400: 4 N = 150 KB of the CU's 160 KB).
// Included from the middle of bldpc_qc_kernel.hpp (needs QcArgs, lds_ld / lds_st).
//
// Where the state lives.  LDS holds ONE array, the a-posteriori value S of every variable (4 N bytes).  The
// check states -- (min1, min2) and a word with the index of the minimum edge and the sign bit of every output,
// from which R_p = +-min1 / +-min2 is rebuilt exactly (bldpc_qcc_kernel.hpp, bldpc_实习/LDPC_Decoder.cu:298-312)
// -- live in REGISTERS of the thread that owns the check: thread tid owns the checks at circulant positions
// tid + z*TPB of every block row, 3 VGPRs per state.  Nothing but the channel values (re-read once per
// iteration, L2-resident) and the packed hard bits touches HBM: 4 N bytes per frame and iteration instead of
// the 16 E + 8 N of the reference schedule (2.66 MB -> 154 KB at config 4).
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
// One block column stays out of LDS altogether: a column LC that meets every block row (the host picks one; codes
// without one do not use this kernel).  The order of the check rows INSIDE a block row is immaterial (checks have no
// outputs), so the host rotates every block row until its LC block has shift 0 (all other shifts of the row move
// with it); then variable (LC, t) meets check (j, t) in every block row j, i.e. only checks owned by the thread that
// also owns position t: its S and its running sum are registers.  The host also lists that block first in every
// row: the order of a row's slots does not matter either (min1, min2 and the sign product are symmetric, and when the
// minimum is duplicated min1 == min2, so the "first index" of LDPC_Decoder.cu:298-305 selects among equal values).
// The per-edge tables (column, shift, first-edge mark) are wave-uniform scalar loads, fetched one block row ahead.
// When TPB does not divide Z the last tile of the circulant is covered by the first waves only; those waves run
// an instantiation of the iteration loop with one more tile than the others (same number of barriers in both).
// (no includes, no namespace: this file is spliced into namespace cldpc by bldpc_qc_kernel.hpp)

template <int J_, int L_, int Z_, int TPB_, int WCS_, int MINW_, int YB_> struct QcrGeom {
    static constexpr int J = J_, L = L_, Z = Z_, TPB = TPB_, WCS = WCS_, ZR = (Z + TPB - 1) / TPB, MINW = MINW_; // every row has >= MINW edges
    static constexpr int YB = YB_; // channel values per batch of the closing pass
    static constexpr bool RAGGED = (Z % TPB) != 0;
    static_assert((L * Z) % (TPB * YB) == 0, "the closing pass runs in whole batches");
    static_assert(Z % 64 == 0 && TPB % 64 == 0 && TPB <= 1024, "threads must tile the circulant in whole waves");
    static_assert(WCS <= 27 && MINW >= 2, "sign bits and the 5-bit index share one word; slot 0 is the register-resident column");
};

// CN slot word: column | shift << 8 | first edge of its column << 19 | padding slot << 21
__host__ __device__ inline unsigned qcr_cn_meta(int col, int shift, int first, int pad)
{
    return (unsigned)col | ((unsigned)shift << 8) | ((unsigned)first << 19) | ((unsigned)pad << 21);
}

typedef __attribute__((address_space(4))) const unsigned qcr_const_u32;

// Iterations 1 .. max_iter-1 for a wave that covers NZ tiles of the circulant.  S0[z]: a-posteriori value of
// variable (LC, tid + z*TPB), kept in registers across the whole decode.
template <typename GM, bool HIST, int NZ>
__device__ __forceinline__ void qcr_iterations(const QcArgs &a, char *lds, int *lds_flag, const __amdgpu_buffer_rsrc_t yrs,
                                               float (&S0)[GM::ZR], unsigned long long &hist, int &stop)
{
    constexpr int J = GM::J, Z = GM::Z, TPB = GM::TPB, WCS = GM::WCS, MINW = GM::MINW, N = GM::L * Z, NS = N / TPB, YB = GM::YB;
    const int tid = threadIdx.x;
    // row slots, wave-uniform; read through the constant address space so that they are scalar loads whatever the
    // compiler can or can not prove about the kernel's own stores
    const qcr_const_u32 *cm = (const qcr_const_u32 *)a.cn_meta;
    const int lcbase = a.lc * Z; // first variable of the register-resident column
    const int dummy = (lcbase + tid) * 4; // that column's (unused) LDS words take the writes of a light row's padding slots
    auto y_at = [&](int stride_idx) -> float { // y[tid + stride_idx * TPB]
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, tid * 4, stride_idx * TPB * 4, 0));
    };
    auto y_lc = [&](int z) -> float { // y[LC*Z + tid + z * TPB]
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, tid * 4, (lcbase + z * TPB) * 4, 0));
    };
    // R_p of a check from its compressed state: magnitude m2 on the edge that held the minimum, m1 elsewhere, and the edge's sign bit.
    // `oh` = 1 << (index of that edge), formed once per state and phase: the select is then a sign-extended bit (v_bfe_i32) and a
    // bit-field insert (v_bfi_b32), two full-rate instructions, where a compare + v_cndmask_b32 are two of the half-rate class
    // (profiles/r02_micro_rates.txt).  Same bits.  PON J12_L69_Z256: 357 -> 373 k codewords/s.  (k_qcr2 keeps the compare form: there the
    // same change costs registers -- 155 -> 168 VGPRs with spills -- and J15_L30_Z1280 fell from 212 to 178 k codewords/s.)
    auto recon = [&](float a1, float a2, unsigned ww, unsigned oh, int p) -> float {
        // (inline asm: left to itself the compiler rebuilds a compare and a v_cndmask_b32 out of the C form)
        unsigned sel, mag, r;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(oh), "n"(p));               // all ones on the edge of the minimum
#ifdef QCR_BFI /* experiment: the bit-field inserts as v_bfi_b32 (half rate) */
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(mag) : "v"(sel), "v"(a2), "v"(a1));      // (m2 & sel) | (m1 & ~sel)
#else
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(mag) : "v"(sel), "v"(a2), "v"(a1)); // (m2 & sel) | (m1 & ~sel), full rate
#endif
        const unsigned sh = ww << (31 - (WCS - 1 - p));                                 // the edge's sign bit at bit 31
#ifdef QCR_BFI
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(0x80000000u), "v"(sh), "v"(mag)); // (sh & sign) | (mag & ~sign)
#else
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(0x80000000u), "v"(sh), "v"(mag)); // (sh & sign) | (mag & ~sign)
#endif
        return u2f(r);
    };
    // byte address of the S value behind row slot m for circulant position t: column base + (t + shift) mod Z
    auto s_addr = [&](unsigned m, int t) -> int {
        const unsigned c1 = (unsigned)t + ((m >> 8) & 2047u);
        const unsigned c = min(c1, c1 - (unsigned)Z); // c1 - Z wraps to a huge value unless c1 >= Z
        return (int)(((m & 255u) * (unsigned)Z + c) * 4u);
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

    for (int it = 1; it < a.max_iter; it++) {
        // The tables are loop-invariant and the compiler knows it: left alone it hoists every rotated address of every
        // edge out of the iteration loop and spills them.  A fresh (opaque) copy of the pointer per phase keeps the
        // loads, and the addresses computed from them, where they are used.
        asm volatile("" : "+s"(cm));
        // ---- phase 1: check nodes on S of iteration `it` (LDPC_Decoder.cu:279-314).  Straight-line code: a light
        // row's padding slots (p >= MINW, meta bit 21) read some valid address and are replaced by +inf, which is
        // neutral for min1/min2, never the first minimum, and has sign 0.
#pragma unroll
        for (int j = 0; j < J; j++) {
            unsigned m[WCS];
#pragma unroll
            for (int p = 1; p < WCS; p++) m[p] = cm[j * WCS + p];
            float Sv[NZ][WCS];
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                Sv[z][0] = S0[z];
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    float sv[1];
                    lds_ld<1>(sv, lds, s_addr(m[p], tid + z * TPB));
                    Sv[z][p] = sv[0];
                }
            }
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                float n1 = __builtin_inff(), n2 = __builtin_inff();
                unsigned signs = 0, lower = 0;
                const unsigned oh = 1u << (w2[j][z] >> 27); // the edge that held the previous minimum, one-hot (recon)
#pragma unroll
                for (int p = 0; p < WCS; p++) {
                    float q = Sv[z][p] - recon(m1[j][z], m2[j][z], w2[j][z], oh, p); // Q = S - R (:206-209)
                    if (p >= MINW) q = ((m[p] >> 21) & 1u) ? __builtin_inff() : q;
                    const float aq = __builtin_fabsf(q);
                    // "this edge lowered the running minimum" as a bit per edge: the sign of |q| - n1 (equal values give +0, inf - inf the
                    // positive default NaN), see k_qcr2
                    lower = __builtin_amdgcn_alignbit(lower, f2u(aq - n1), 31);
                    n2 = __builtin_amdgcn_fmed3f(n1, n2, aq);
                    n1 = __builtin_fminf(n1, aq);
                    signs = __builtin_amdgcn_alignbit(signs, f2u(q), 31); // (signs << 1) | sign(q)
                }
                // first edge holding the minimum (:298-305) = the last edge that lowered it = the lowest set bit (edge p at bit WCS-1-p)
                const int idx = WCS - 1 - (int)__builtin_ctz(lower | (1u << (WCS - 1)));
                // R_p = Sign[25]*Sign[p] * magnitude: output sign bit p = parity of all signs XOR sign p
                if (__builtin_popcount(signs) & 1) signs ^= (1u << WCS) - 1u;
                m1[j][z] = n1; m2[j][z] = n2;
                w2[j][z] = signs | ((unsigned)idx << 27);
                // the state is complete HERE: three registers, not the chain of values it was computed from (left alone
                // the compiler sinks the min2 chain into phase 2, where its inputs have to be spilled to survive)
                asm volatile("" : "+v"(m1[j][z]), "+v"(m2[j][z]), "+v"(w2[j][z]));
            }
            __builtin_amdgcn_sched_barrier(0); // one block row's reads in flight at a time: bounds the VGPRs
        }
        __syncthreads();

        // ---- phase 2: S of iteration it+1 = ((0 + R_0) + R_1 + ...), block rows in ascending order (:188-204) ----
        asm volatile("" : "+s"(cm));
        float acc0[NZ], yv[YB];
#pragma unroll
        for (int z = 0; z < NZ; z++) acc0[z] = 0.0f;
        unsigned mn[WCS];
#pragma unroll
        for (int p = 1; p < WCS; p++) mn[p] = cm[p];
#pragma unroll
        for (int j = 0; j < J; j++) {
            unsigned m[WCS];
#pragma unroll
            for (int p = 1; p < WCS; p++) m[p] = mn[p];
            if (j + 1 < J) { // the next block row's slots: in flight across this one's barrier
#pragma unroll
                for (int p = 1; p < WCS; p++) mn[p] = cm[(j + 1) * WCS + p];
            } else { // first batch of the closing pass
#pragma unroll
                for (int i = 0; i < YB; i++) yv[i] = y_at(i);
            }
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                const int t = tid + z * TPB;
                float acc[WCS];
                int va[WCS];
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    va[p] = s_addr(m[p], t);
                    if (p >= MINW) va[p] = ((m[p] >> 21) & 1u) ? dummy : va[p];
                    float sv[1];
                    lds_ld<1>(sv, lds, va[p]);
                    acc[p] = sv[0];
                }
                const unsigned oh = 1u << (w2[j][z] >> 27);
                acc0[z] += recon(m1[j][z], m2[j][z], w2[j][z], oh, 0); // slot 0 = column LC: this thread's own variable
#pragma unroll
                for (int p = 1; p < WCS; p++) {
                    const float prev = ((m[p] >> 19) & 1u) ? 0.0f : acc[p]; // a column's first edge starts from 0
                    const float sv[1] = {prev + recon(m1[j][z], m2[j][z], w2[j][z], oh, p)};
                    lds_st<1>(lds, va[p], sv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        // ---- ... + y closes every sum (:205): one aligned pass, channel values re-read (L2-resident).  The pass also
        // sweeps column LC's unused words (1/L of it) rather than special-casing them.
        // flag bookkeeping (HIST): OR of the sign bits of the values that lie inside the first `length` variables and
        // outside column LC's unused words.  `lim` is made opaque per iteration: the range tests are loop-invariant and
        // would otherwise be hoisted, one register per stride, and spilled.
        unsigned badbits = 0;
        int lim = a.length - tid, lcrel = lcbase - tid;
        if (HIST) asm volatile("" : "+v"(lim), "+v"(lcrel));
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
#pragma unroll
            for (int i = 0; i < YB; i++) lds_ld<1>(sv[i], lds, (tid + (i0 + i) * TPB) * 4);
#pragma unroll
            for (int i = 0; i < YB; i++) {
                const int v = tid + (i0 + i) * TPB;
                sv[i][0] += yv[i];
                lds_st<1>(lds, v * 4, sv[i]);
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

// LDS: S float[N] | flag.
// PERSIST (per-frame exit only): frames of this XCD from the counter a.work[xcd], see k_qcr2 / k_qc2p
template <typename GM, bool HIST, bool PERSIST = false> __global__ __launch_bounds__(GM::TPB) void k_qcr(QcArgs a)
{
    constexpr int Z = GM::Z, TPB = GM::TPB, ZR = GM::ZR, N = GM::L * Z;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int chunk = (a.nWG + 7) >> 3; // XCD-aware workgroup id, see k_qc
    int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (!PERSIST && wg >= a.nWG) return;
    const int F = a.F;
    const int tid = threadIdx.x;
    int *lds_flag = reinterpret_cast<int *>(lds + N * 4);
    const bool zlast = !GM::RAGGED || (tid + (ZR - 1) * TPB < Z); // wave-uniform: this wave covers the last tile too
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
    // the frame's channel values as a buffer resource: loads take one VGPR offset (4 tid) plus a scalar offset, instead
    // of a 64-bit address pair per load
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(yf), 0, N * 4, 0x00020000);

    int bad = 0;
    for (int v = tid; v < N; v += TPB) { // iteration 1: S = (0 + 0 + ...) + y
        const float sv[1] = {0.0f + yf[v]};
        lds_st<1>(lds, v * 4, sv);
        if (HIST) bad |= (int)(v < a.length) & (int)(sv[0] < 0);
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
    } else if (GM::RAGGED && !zlast) qcr_iterations<GM, HIST, (GM::RAGGED ? ZR - 1 : ZR)>(a, lds, lds_flag, yrs, S0, hist, stop);
    else qcr_iterations<GM, HIST, ZR>(a, lds, lds_flag, yrs, S0, hist, stop);
    const int last = stop ? stop : a.max_iter;

    // ---- outputs from S of the last iteration ----
#pragma unroll
    for (int z = 0; z < ZR; z++)
        if (z < ZR - 1 || zlast) { // column LC comes back from the registers
            const float sv[1] = {S0[z]};
            lds_st<1>(lds, (a.lc * Z + tid + z * TPB) * 4, sv);
        }
    __syncthreads();
    bad = 0;
    for (int n = tid; n < N; n += TPB) {
        float sv[1];
        lds_ld<1>(sv, lds, n * 4);
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

