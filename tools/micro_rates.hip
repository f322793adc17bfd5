// Experiment: sustained issue rates on gfx950 that the fused decoders' on-chip ceilings are computed from.
//   part 1  VALU: SIMD-cycles per wave-instruction for the instructions the min-sum loop uses, 1/2/4/6/8 waves per SIMD
//   part 2  LDS:  CU-cycles per wave-instruction for ds_read_b64 / ds_write_b64 / b32 / addtid and the loop's mixes
//   part 3  the loop's own mix: LDS operations and VALU instructions per edge pair side by side (do the pipes overlap?)
// Cycles are shader cycles measured in-kernel (s_memtime around the loop, median-free: max over waves of one CU is
// what bounds a CU), the clock the chip held is printed next to them (delta s_memtime / delta s_memrealtime x 100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro_rates.hip -o build/micro_rates ; output committed under profiles/.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { V_ADD, V_SUB, V_FMA, V_PK_ADD, V_PK_FMA, V_XOR, V_MIN, V_MAX, V_MED3, V_MED3_ABS, V_MIN_ABS, V_CNDMASK, V_DPP, V_LSHL_ADD, V_BFI, V_AND_OR, V_PERMLANE32, V_MIN3, V_MOV, V_NOPS, V_CNDMASK_S, V_AND, V_MAX3, V_CMP, V_SUB_NEG, V_XOR3, V_MIN_U32, V_MAX_U32, V_MIN_I32, V_SUB_U32, V_ADD_U32, V_ASHR, V_LSHR, V_BFE, V_PERM, V_MUL, V_OR, V_ADD3, V_MED3_U32, V_PK_MOV, V_MIN_U16, V_SUBREV, V_MAX_I32, V_BITOP3, V_XOR_CHAIN, V_BITOP3_CHAIN };

template <int OP> __global__ __launch_bounds__(256) void k_valu(float *out, Stamp *st, int iters, float seed)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    float a[8];
    uint32_t u[8];
    v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; u[i] = threadIdx.x * 7 + i; p[i] = {seed + i, seed - threadIdx.x}; }
    const unsigned mlo = __builtin_amdgcn_readfirstlane(0x55555555u + threadIdx.x / 64), mhi = __builtin_amdgcn_readfirstlane(0x33333333u + threadIdx.x / 64);
    const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == V_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == V_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == V_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == V_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == V_CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(mask));
                if (OP == V_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == V_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if (OP == V_SUB_NEG) asm volatile("v_sub_f32 %0, %0, -%1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_SUBREV) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_ASHR) asm volatile("v_ashrrev_i32 %0, 31, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_LSHR) asm volatile("v_lshrrev_b32 %0, 16, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_BFE) asm volatile("v_bfe_u32 %0, %1, 3, 11" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_MED3_U32) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_XOR_CHAIN) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[0]) : "v"(u[(i + 1) & 7]));                                  // every instruction waits for the one before
                if (OP == V_BITOP3_CHAIN) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[0]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_PK_MOV) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == V_MIN_U16) asm volatile("v_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_XOR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == V_MED3_ABS) asm volatile("v_med3_f32 %0, %0, %1, |%2|" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == V_MIN_ABS) asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
                if (OP == V_DPP) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 4) & 7]));
                if (OP == V_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == V_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (OP == V_PERMLANE32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
                if (OP == V_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == V_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == V_NOPS) asm volatile("s_nop 0");
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + u[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c0, c1, r0, r1};
}

// LDS mixes.  One group = what one wave issues per "edge pair" of the fused loop: NR ds_read_b64, NW ds_write_b64
// (or the b32 / addtid forms), NV VALU instructions of the loop's kinds; 8 groups, then s_waitcnt lgkmcnt(0).
enum { L_R64, L_W64, L_R32, L_W32, L_WADDTID, L_R128, L_W128, L_MIX31, L_MIX21, L_MIX31_V14, L_MIX21_V14, L_V14, L_MIX31_V14S, L_MIX21_V12S, L_V14S, L_MIX_ADDTID, L_R2_64, L_W2_64, L_R2ST64 };

template <int OP> __global__ __launch_bounds__(256) void k_lds(float *out, Stamp *st, int iters, float seed)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = wave * 4096 + lane * 8; // conflict-free, 8 x 512-byte rows per wave
    const unsigned base16 = wave * 4096 + lane * 16;
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float *>(lds)[i] = seed + i;
    __syncthreads();
    v2f d[8], p[8];
    v4f q[4];
    float a[8];
    uint32_t u[8];
    for (int i = 0; i < 8; i++) { d[i] = {seed + i, seed - i}; p[i] = {seed + i, seed - threadIdx.x}; a[i] = seed + threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
    for (int i = 0; i < 4; i++) q[i] = {seed, seed, seed, seed};
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(wave * 4096u)) : "memory");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define VALU2(i)                                                                      \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));                    \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define VALU14_SIMPLE(i) /* 14 two-operand instructions */                            \
    VALU2(i) VALU2((i + 1) & 7) VALU2((i + 2) & 7) VALU2((i + 3) & 7) VALU2((i + 4) & 7) VALU2((i + 5) & 7) VALU2((i + 6) & 7)
#define VALU14_LOOP(i) /* the loop's own mix for 2 frames: pk_add(Q), 2 xor, 2 med3|abs|, 2 min|abs|, 2 med3, 2 xor, pk_add(S) */ \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));                                                 \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                                                   \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 1) & 7]) : "v"(u[(i + 2) & 7]));                                         \
    asm volatile("v_med3_f32 %0, %0, %1, |%2|" : "+v"(a[(i + 2) & 7]) : "v"(seed), "v"(a[(i + 3) & 7]));                       \
    asm volatile("v_med3_f32 %0, %0, %1, |%2|" : "+v"(a[(i + 3) & 7]) : "v"(seed), "v"(a[(i + 4) & 7]));                       \
    asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(a[(i + 4) & 7]) : "v"(a[(i + 5) & 7]));                                       \
    asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(a[(i + 5) & 7]) : "v"(a[(i + 6) & 7]));                                       \
    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[(i + 6) & 7]) : "v"(seed), "v"(a[(i + 7) & 7]));                         \
    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[(i + 7) & 7]) : "v"(seed), "v"(a[i]));                                   \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 2) & 7]) : "v"(u[(i + 3) & 7]));                                         \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 3) & 7]) : "v"(u[(i + 4) & 7]));                                         \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i + 2) & 7]) : "v"(p[(i + 3) & 7]));
#define VALU14_SPLIT(i) /* the same arithmetic with no packed instruction: 14 single-frame instructions */                    \
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));                                                   \
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[(i + 1) & 7]) : "v"(a[(i + 2) & 7]));                                         \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                                                   \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 1) & 7]) : "v"(u[(i + 2) & 7]));                                         \
    asm volatile("v_med3_f32 %0, %0, %1, |%2|" : "+v"(a[(i + 2) & 7]) : "v"(seed), "v"(a[(i + 3) & 7]));                       \
    asm volatile("v_med3_f32 %0, %0, %1, |%2|" : "+v"(a[(i + 3) & 7]) : "v"(seed), "v"(a[(i + 4) & 7]));                       \
    asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(a[(i + 4) & 7]) : "v"(a[(i + 5) & 7]));                                       \
    asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(a[(i + 5) & 7]) : "v"(a[(i + 6) & 7]));                                       \
    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[(i + 6) & 7]) : "v"(seed), "v"(a[(i + 7) & 7]));                         \
    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[(i + 7) & 7]) : "v"(seed), "v"(a[i]));                                   \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 2) & 7]) : "v"(u[(i + 3) & 7]));                                         \
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(i + 3) & 7]) : "v"(u[(i + 4) & 7]));                                         \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(i + 4) & 7]) : "v"(a[(i + 5) & 7]));                                         \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(i + 5) & 7]) : "v"(a[(i + 6) & 7]));
#define RD64(i, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(d[i]) : "v"(base));
#define WR64(i, off) asm volatile("ds_write_b64 %0, %1 offset:" #off : : "v"(base), "v"(d[i]));
    for (int it = 0; it < iters; it++) {
        if (OP == L_R64) {
#define X(i) RD64(i, 0) RD64(i, 512)
            REP8(X)
#undef X
        }
        if (OP == L_W64) {
#define X(i) WR64(i, 0) WR64(i, 512)
            REP8(X)
#undef X
        }
        if (OP == L_R32) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(a[i]) : "v"(base)); asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(a[i]) : "v"(base));
            REP8(X)
#undef X
        }
        if (OP == L_W32) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:0" : : "v"(base), "v"(a[i])); asm volatile("ds_write_b32 %0, %1 offset:512" : : "v"(base), "v"(a[i]));
            REP8(X)
#undef X
        }
        if (OP == L_WADDTID) {
#define X(i) asm volatile("ds_write_addtid_b32 %0 offset:0" : : "v"(a[i]) : "memory"); asm volatile("ds_write_addtid_b32 %0 offset:512" : : "v"(a[i]) : "memory");
            REP8(X)
#undef X
        }
        if (OP == L_R128) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(q[i & 3]) : "v"(base16)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q[i & 3]) : "v"(base16));
            REP8(X)
#undef X
        }
        if (OP == L_W128) {
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:0" : : "v"(base16), "v"(q[i & 3])); asm volatile("ds_write_b128 %0, %1 offset:1024" : : "v"(base16), "v"(q[i & 3]));
            REP8(X)
#undef X
        }
        if (OP == L_R2_64) {
#define X(i) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(q[i & 3]) : "v"(base));
            REP8(X)
#undef X
        }
        if (OP == L_W2_64) {
#define X(i) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" : : "v"(base), "v"(d[i]), "v"(d[(i + 1) & 7]));
            REP8(X)
#undef X
        }
        if (OP == L_MIX31 || OP == L_MIX31_V14 || OP == L_MIX31_V14S) { // today's loop: S read, R read (CN), R read (VN), R write
#define X(i) RD64(i, 0) RD64((i + 1) & 7, 512) RD64((i + 2) & 7, 1024) if (OP == L_MIX31_V14) { VALU14_LOOP(i) } if (OP == L_MIX31_V14S) { VALU14_SPLIT(i) } WR64((i + 3) & 7, 1536)
            REP8(X)
#undef X
        }
        if (OP == L_MIX21 || OP == L_MIX21_V14 || OP == L_MIX21_V12S) { // previous outputs kept in registers: S read, R read (VN), R write
#define X(i) RD64(i, 0) RD64((i + 2) & 7, 1024) if (OP == L_MIX21_V14) { VALU14_LOOP(i) } if (OP == L_MIX21_V12S) { VALU14_SPLIT(i) } WR64((i + 3) & 7, 1536)
            REP8(X)
#undef X
        }
        if (OP == L_MIX_ADDTID) { // 2 b64 reads + 2 addtid b32 writes (what a plane-per-frame R array would cost to write)
#define X(i) RD64(i, 0) RD64((i + 2) & 7, 1024) asm volatile("ds_write_addtid_b32 %0 offset:1536" : : "v"(a[i]) : "memory"); asm volatile("ds_write_addtid_b32 %0 offset:2048" : : "v"(a[(i + 1) & 7]) : "memory");
            REP8(X)
#undef X
        }
        if (OP == L_V14) {
#define X(i) VALU14_LOOP(i)
            REP8(X)
#undef X
        }
        if (OP == L_V14S) {
#define X(i) VALU14_SPLIT(i)
            REP8(X)
#undef X
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + u[i] + d[i].x + d[i].y + p[i].x + p[i].y;
    for (int i = 0; i < 4; i++) s += q[i].x + q[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + wave] = {c0, c1, r0, r1};
}

static void report(const char *name, int w, const std::vector<Stamp> &st, double ms, double units, const char *unit)
{
    // Clock the chip held: per wave, delta s_memtime / delta s_memrealtime (100 MHz); the counters differ between XCDs, so
    // only differences taken inside one wave mean anything.  Throughput from the LAUNCH time (HIP events): with more
    // waves than a pipe needs the oldest waves are served first, so a wave's own lifetime is shorter than the launch.
    std::vector<double> clk;
    for (const Stamp &s : st) clk.push_back((double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0));
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[clk.size() / 2];
    printf("%-58s %d waves/SIMD: %6.2f %s (%.3f ms, %.2f GHz)\n", name, w, ms * 1e-3 * ghz * 1e9 / units, unit, ms, ghz);
}

template <int OP> void run_valu(const char *name, int w)
{
    const int blocks = 256 * w, iters = 8000;
    float *out; Stamp *st;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&st, (size_t)blocks * 4 * sizeof(Stamp));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, out, st, 100, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // a SIMD hosts w waves: SIMD-cycles per wave-instruction = wave duration / (instructions per wave * w)
    report(name, w, h, ms, (double)iters * 32 * w, "SIMD-cycles per wave-instruction");
    (void)hipFree(out); (void)hipFree(st);
}

template <int OP> void run_lds(const char *name, int w, int nlds, int nvalu)
{
    const int blocks = 256 * w, iters = 4000;
    float *out; Stamp *st;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&st, (size_t)blocks * 4 * sizeof(Stamp));
    (void)hipFuncSetAttribute((const void *)k_lds<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lds<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, 50, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // a CU hosts 4*w waves; CU-cycles per group (= one edge pair of one wave) = wave duration / (8 groups * iters * 4 * w)
    char nm[96];
    snprintf(nm, sizeof(nm), "%s [%d LDS + %d VALU per group]", name, nlds, nvalu);
    report(nm, w, h, ms, (double)iters * 8 * 4 * w, "CU-cycles per group");
    (void)hipFree(out); (void)hipFree(st);
}

int main()
{
    printf("== part 1: VALU issue, SIMD-cycles per wave-instruction (independent instructions, 8 registers rotating) ==\n");
    for (int w : {1, 2, 4, 6, 8}) {
        run_valu<V_ADD>("v_add_f32", w); run_valu<V_SUB>("v_sub_f32", w); run_valu<V_FMA>("v_fma_f32", w);
        run_valu<V_PK_ADD>("v_pk_add_f32", w); run_valu<V_PK_FMA>("v_pk_fma_f32", w); run_valu<V_XOR>("v_xor_b32", w);
        run_valu<V_MIN>("v_min_f32", w); run_valu<V_MAX>("v_max_f32", w); run_valu<V_MED3>("v_med3_f32", w);
        run_valu<V_MED3_ABS>("v_med3_f32 |src2|", w); run_valu<V_MIN_ABS>("v_min_f32 |src1| (VOP3)", w); run_valu<V_MIN3>("v_min3_f32", w);
        run_valu<V_CNDMASK>("v_cndmask_b32", w); run_valu<V_DPP>("s_nop1+v_mov_b32_dpp", w); run_valu<V_LSHL_ADD>("v_lshl_add_u32", w);
        run_valu<V_BFI>("v_bfi_b32", w); run_valu<V_AND_OR>("v_and_or_b32", w); run_valu<V_PERMLANE32>("v_permlane32_swap", w);
        run_valu<V_MOV>("v_mov_b32", w); run_valu<V_NOPS>("s_nop 0", w);
        run_valu<V_CNDMASK_S>("v_cndmask_b32 (sgpr mask)", w); run_valu<V_AND>("v_and_b32", w); run_valu<V_MAX3>("v_max3_f32", w);
        run_valu<V_CMP>("v_cmp_lt_f32 vcc", w); run_valu<V_SUB_NEG>("v_sub_f32 neg (VOP3)", w); run_valu<V_XOR3>("v_or3_b32", w);
        run_valu<V_MIN_U32>("v_min_u32", w); run_valu<V_MAX_U32>("v_max_u32", w); run_valu<V_MIN_I32>("v_min_i32", w); run_valu<V_MAX_I32>("v_max_i32", w);
        run_valu<V_SUB_U32>("v_sub_u32", w); run_valu<V_SUBREV>("v_subrev_u32", w); run_valu<V_ADD_U32>("v_add_u32", w); run_valu<V_ASHR>("v_ashrrev_i32", w);
        run_valu<V_LSHR>("v_lshrrev_b32", w); run_valu<V_BFE>("v_bfe_u32", w); run_valu<V_PERM>("v_perm_b32", w); run_valu<V_MUL>("v_mul_f32", w);
        run_valu<V_OR>("v_or_b32", w); run_valu<V_ADD3>("v_add3_u32", w); run_valu<V_MED3_U32>("v_med3_u32", w); run_valu<V_BITOP3>("v_bitop3_b32 (xor3)", w); run_valu<V_XOR_CHAIN>("v_xor_b32, dependent chain", w); run_valu<V_BITOP3_CHAIN>("v_bitop3_b32, dependent chain", w); run_valu<V_PK_MOV>("v_pk_mov_b32", w);
        run_valu<V_MIN_U16>("v_min_u16", w);
        printf("\n");
    }
    printf("== part 2/3: LDS, CU-cycles per group (2 instructions per group in the pure rows) ==\n");
    for (int w : {1, 2, 4, 6, 8}) {
        run_lds<L_R64>("ds_read_b64 x2", w, 2, 0); run_lds<L_W64>("ds_write_b64 x2", w, 2, 0);
        run_lds<L_R32>("ds_read_b32 x2", w, 2, 0); run_lds<L_W32>("ds_write_b32 x2", w, 2, 0);
        run_lds<L_WADDTID>("ds_write_addtid_b32 x2", w, 2, 0);
        run_lds<L_R128>("ds_read_b128 x2", w, 2, 0); run_lds<L_W128>("ds_write_b128 x2", w, 2, 0);
        run_lds<L_R2_64>("ds_read2_b64 x1 (= 2 values)", w, 1, 0); run_lds<L_W2_64>("ds_write2_b64 x1 (= 2 values)", w, 1, 0);
        run_lds<L_MIX31>("3 rd64 + 1 wr64", w, 4, 0); run_lds<L_MIX21>("2 rd64 + 1 wr64", w, 3, 0);
        run_lds<L_MIX_ADDTID>("2 rd64 + 2 wr_addtid32", w, 4, 0);
        run_lds<L_V14>("loop VALU mix (packed)", w, 0, 12); run_lds<L_V14S>("loop VALU mix (split)", w, 0, 14);
        run_lds<L_MIX31_V14>("3 rd64 + 1 wr64 + packed mix", w, 4, 12); run_lds<L_MIX31_V14S>("3 rd64 + 1 wr64 + split mix", w, 4, 14);
        run_lds<L_MIX21_V14>("2 rd64 + 1 wr64 + packed mix", w, 3, 12); run_lds<L_MIX21_V12S>("2 rd64 + 1 wr64 + split mix", w, 3, 14);
        printf("\n");
    }
    return 0;
}
