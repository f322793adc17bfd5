// bldpc_table_kernels.hpp -- generic address-table kernels (BLDPC_KERNEL_TABLE).
//
// Messages live in HBM as rq[slot][F] (slot = m*Wc + p, frame-fastest), the
// layout the reference's Address_Variablenode table is defined against
// (bldpc_实习/Simulation.cu:363-387, LDPC_Decoder.cu:186-196).  They accept ANY
// table, including the as-written one, and any code size; the QC kernel in
// bldpc_qc_kernel.hpp is the fast path for codes whose state fits LDS.
//
// Differences from the reference's kernel pair: one workgroup row per node so
// the table entries are wave-uniform scalar loads, VEC frames per lane with
// 16-byte accesses, hard bits written only when someone will read them, the
// per-frame termination flag folded into the VN pass (the reference copies all
// of D to the host every iteration, LDPC_Decoder.cu:135-149).
#pragma once
#include "bldpc_math.hpp"

namespace cldpc {

constexpr int kMaxWv = 16; // reference bound: R[15] (LDPC_Decoder.cu:175)
constexpr int kMaxWc = 26; // reference bound: Q[25] (LDPC_Decoder.cu:267)

template <int VEC> struct VecT;
template <> struct VecT<1> { using f = float; using i = int; };
template <> struct VecT<2> { using f = float2; using i = int2; };
template <> struct VecT<4> { using f = float4; using i = int4; };

template <int VEC> __device__ __forceinline__ void vload(float (&d)[VEC], const float *p)
{
    typename VecT<VEC>::f v = *reinterpret_cast<const typename VecT<VEC>::f *>(p);
    __builtin_memcpy(d, &v, sizeof(v));
}
template <int VEC> __device__ __forceinline__ void vstore(float *p, const float (&s)[VEC])
{
    typename VecT<VEC>::f v;
    __builtin_memcpy(&v, s, sizeof(v));
    *reinterpret_cast<typename VecT<VEC>::f *>(p) = v;
}
template <int VEC> __device__ __forceinline__ void vstore_i(int *p, const int (&s)[VEC])
{
    typename VecT<VEC>::i v;
    __builtin_memcpy(&v, s, sizeof(v));
    *reinterpret_cast<typename VecT<VEC>::i *>(p) = v;
}

struct TableArgs {
    float *rq;              // [M*Wc][F]
    const float *y;         // [N][F]
    const int *addr;        // [N][Wv]
    const int *node_list;   // VN order by level (nullptr = identity)
    const unsigned char *wv_blk; // [L]
    const unsigned char *wc_blk; // [J]
    int *D;                 // [N+1][F] or nullptr (skip hard-bit store this pass)
    float *app;             // [N][F] or nullptr
    int *bad;               // [F] set to 1 when a frame has a 1 among its first `length` bits (nullptr = skip)
    const int *iters;       // per-frame exit: [F] iteration at which the frame stopped, 0 = still running (nullptr = off);
                            // a stopped frame keeps the D / app columns of that iteration
    int F, Z, Wv, Wc, length;
};

// Variable-node pass over nodes node_list[n0 .. n0+count) (LDPC_Decoder.cu:172-211).
// grid = (ceil(F / (VEC*256)), min(count, 65535)); block = 256.
template <int VEC> __global__ __launch_bounds__(256) void k_table_vn(TableArgs a, int n0, int count)
{
    const int f = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (f >= a.F) return;
    for (int k = blockIdx.y; k < count; k += gridDim.y) {
        const int n = a.node_list ? a.node_list[n0 + k] : n0 + k; // wave-uniform
        const int w = a.wv_blk[n / a.Z];
        const int *ad = a.addr + (size_t)n * a.Wv;
        float R[kMaxWv][VEC];
        float S[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) S[v] = 0.0f; // Add_result defined as 0
#pragma unroll
        for (int i = 0; i < kMaxWv; i++)
            if (i < w) vload<VEC>(R[i], a.rq + (size_t)ad[i] * a.F + f);
#pragma unroll
        for (int i = 0; i < kMaxWv; i++)
            if (i < w) {
#pragma unroll
                for (int v = 0; v < VEC; v++) S[v] += R[i][v];
            }
        float yv[VEC];
        vload<VEC>(yv, a.y + (size_t)n * a.F + f);
#pragma unroll
        for (int v = 0; v < VEC; v++) S[v] += yv[v];
#pragma unroll
        for (int i = 0; i < kMaxWv; i++)
            if (i < w) {
                float Q[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) Q[v] = S[v] - R[i][v];
                vstore<VEC>(a.rq + (size_t)ad[i] * a.F + f, Q);
            }
        bool frozen = false, stopped[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            stopped[v] = a.iters && a.iters[f + v] != 0;
            frozen = frozen || stopped[v];
        }
        if (a.D) {
            int d[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) d[v] = (S[v] < 0) ? 1 : 0;
            if (!frozen) vstore_i<VEC>(a.D + (size_t)n * a.F + f, d);
            else {
#pragma unroll
                for (int v = 0; v < VEC; v++)
                    if (!stopped[v]) a.D[(size_t)n * a.F + f + v] = d[v];
            }
        }
        if (a.app) {
            if (!frozen) vstore<VEC>(a.app + (size_t)n * a.F + f, S);
            else {
#pragma unroll
                for (int v = 0; v < VEC; v++)
                    if (!stopped[v]) a.app[(size_t)n * a.F + f + v] = S[v];
            }
        }
        if (a.bad && n < a.length) {
#pragma unroll
            for (int v = 0; v < VEC; v++)
                if (S[v] < 0) a.bad[f + v] = 1; // every writer stores the same value
        }
    }
}

// Check-node pass over all M rows (LDPC_Decoder.cu:262-315).
// grid = (ceil(F / (VEC*256)), min(M, 65535)); block = 256.
template <int VEC> __global__ __launch_bounds__(256) void k_table_cn(TableArgs a, int M)
{
    const int f = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (f >= a.F) return;
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        const int w = a.wc_blk[m / a.Z];
        float *row = a.rq + (size_t)m * a.Wc * a.F + f;
        float Q[kMaxWc][VEC];
        CnAcc acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v].init();
#pragma unroll
        for (int i = 0; i < kMaxWc; i++)
            if (i < w) vload<VEC>(Q[i], row + (size_t)i * a.F);
#pragma unroll
        for (int i = 0; i < kMaxWc; i++)
            if (i < w) {
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v].add(Q[i][v]);
            }
        uint32_t key[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) key[v] = acc[v].key();
#pragma unroll
        for (int i = 0; i < kMaxWc; i++)
            if (i < w) {
                float R[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) R[v] = cn_out(Q[i][v], acc[v].m2, key[v]);
                vstore<VEC>(row + (size_t)i * a.F, R);
            }
    }
}

// Per-frame termination bookkeeping after one iteration (LDPC_Decoder.cu:134-153):
//   flag = !bad; D[N][f] = flag; flag_hist bit; count frames whose flag is set; reset bad.
// Per-frame exit (iters != nullptr): a running frame whose flag comes up, or that reaches the last iteration, records
// `it` as its iteration count and stops; a stopped frame counts as flagged from then on.
__global__ __launch_bounds__(256) void k_flags(int *bad, int *D_flag_row, unsigned long long *flag_hist, int *ok_count,
                                               int F, int it, int *iters, int last)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    int flag = 0;
    if (f < F) {
        flag = bad[f] ? 0 : 1;
        bad[f] = 0;
        if (iters && iters[f] != 0) flag = 1; // stopped earlier: its flag row entry stays as written then
        else {
            if (D_flag_row) D_flag_row[f] = flag;
            if (iters && (flag || last)) iters[f] = it;
        }
        if (flag_hist && flag && it <= 64) flag_hist[f] |= (1ull << (it - 1));
    }
    if (ok_count) {
        unsigned long long b = __ballot(flag);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(ok_count, __popcll(b));
    }
}

// Device-side Statistic (Simulation.cu:245-262), two stages so that the N*F-word read of D streams at HBM
// rate: stage 1 counts message-bit errors per frame over a slice of rows (4 frames per lane, 16-byte loads,
// one atomic per frame and slice), stage 2 classifies each frame and reduces the five counters.
__global__ __launch_bounds__(256) void k_stat_errors(const int *D, const int *cw, int F, int length, int rows_per_slice, int *errs)
{
    const int f = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (f >= F) return;
    const int k0 = blockIdx.y * rows_per_slice, k1 = min(length, k0 + rows_per_slice);
    int e[4] = {0, 0, 0, 0};
    if (f + 3 < F && (F & 3) == 0) {
        int k = k0;
        if (!cw) { // all-zero codeword (the reference's PN_Message 0): eight rows in flight per lane
            for (; k + 8 <= k1; k += 8) {
                int4 d[8];
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = *reinterpret_cast<const int4 *>(D + (size_t)(k + i) * F + f);
#pragma unroll
                for (int i = 0; i < 8; i++) { e[0] += d[i].x != 0; e[1] += d[i].y != 0; e[2] += d[i].z != 0; e[3] += d[i].w != 0; }
            }
        }
        for (; k < k1; k++) {
            const int4 d = *reinterpret_cast<const int4 *>(D + (size_t)k * F + f);
            int4 c = make_int4(0, 0, 0, 0);
            if (cw) c = *reinterpret_cast<const int4 *>(cw + (size_t)k * F + f);
            e[0] += d.x != c.x; e[1] += d.y != c.y; e[2] += d.z != c.z; e[3] += d.w != c.w;
        }
    } else {
        for (int k = k0; k < k1; k++)
            for (int i = 0; i < 4 && f + i < F; i++) e[i] += D[(size_t)k * F + f + i] != (cw ? cw[(size_t)k * F + f + i] : 0);
    }
    for (int i = 0; i < 4 && f + i < F; i++)
        if (e[i]) atomicAdd(&errs[f + i], e[i]);
}

__global__ __launch_bounds__(256) void k_stat_final(int *errs, const int *flag_row, int F, int itera, const int *iters,
                                                    long long *counters)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    long long v[5] = {0, 0, 0, 0, 0};
    if (f < F) {
        const int err = errs[f], flag = flag_row[f];
        errs[f] = 0; // leave the scratch zeroed for the next call
        v[0] = (err != 0 || flag == 0) ? 1 : 0; // num_Error_Frames
        v[1] = err;                              // num_Error_Bits
        v[2] = iters ? iters[f] : itera;         // Total_Iteration += iteraTime per frame (Simulation.cu:262)
        v[3] = (err != 0 && flag == 1) ? 1 : 0; // num_False_Frames
        v[4] = (err == 0 && flag == 0) ? 1 : 0; // num_Alarm_Frames
    }
#pragma unroll
    for (int c = 0; c < 5; c++) {
        long long x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd((unsigned long long *)&counters[c], (unsigned long long)x);
    }
}

} // namespace cldpc
