// nbldpc_api.hip -- host side of the GF(q) EMS decoder behind include/nbldpc.h.
//
// Readers restate myNBLDPC/src/Simulation.cpp:347-467 (Get_H) and src/GF.cpp:68-117 (GFInitial);
// nbldpc_ems_decode_batch replaces Decoding_EMS / Decoding_EMS_GPU (src/LDPC_Decoder.cpp:172-317,
// src/Decode_GPU.cu:138-356) for a batch of frames; the channel helpers restate
// src/LDPC_Encoder.cpp:41-79 and src/main.cu:203-228.
#include "../../include/nbldpc.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "common.hpp"
#include "nbldpc_kernel.hpp"
#include "nbldpc_pipe_kernel.hpp"
#include "nbldpc_tmm_kernel.hpp"
#include "nbldpc_wide_kernel.hpp"
#include "nbldpc_hbm_kernel.hpp"

using namespace cldpc;

struct nbldpc_code {
    int N = 0, M = 0, q = 0, m = 0, dv = 0, dc = 0;
    int *d_vn_w = nullptr, *d_vn_thr = nullptr, *d_vn_gf = nullptr;
    int *d_cn_w = nullptr, *d_cn_src = nullptr, *d_cn_gf = nullptr, *d_cn_vn = nullptr;
    unsigned char *d_mul = nullptr;
    size_t lds_bytes = 0;
    // trellis min-max decoders (nbldpc_tmm_decode_batch)
    int *d_cn_hinv = nullptr, *d_row_order = nullptr, *d_level_begin = nullptr;
    int levels = 0;
    bool tmm_ok = false;
    int zero_coeff = 0; // an edge with coefficient 0 exists (EMS only, see nbldpc_code_create)
    const char *last_kernel = "none"; // nbldpc_last_kernel
    int persist_grid = 0; // k_nb_ems / k_nb_ems_wide: workgroups that fill the chip once (CUs x workgroups per CU)
    bool hbm = false;   // decoded by k_nb_ems_hbm (state in a global-memory workspace): LDS too small or rows heavier than kNbMaxW
    int pipe_grid = 0;        // k_nb_ems2 (two frames in flight per workgroup): resident workgroups, 0 = kernel not offered for this code
    size_t pipe_lds = 0;
    int tmm_grid[2] = {0, 0}; // k_nb_tmm<q, layered>: the same, per schedule (0 = flooding, 1 = layered), fixed at create time
    bool no_persist = false;  // NBLDPC_NO_PERSIST, read once at create time (tests / experiments): one workgroup per frame
    // Frame counters of the persistent kernels: a ring of kWorkSlots words, one per decode call in flight (the call zeroes its
    // slot stream-ordered before the launch), instead of a hipMallocAsync / hipFreeAsync pair per call.  Calls on different
    // streams never share a slot unless more than kWorkSlots calls on this code object are in flight at once.
    int *d_work = nullptr;
    std::atomic<unsigned> work_next{0};
};
constexpr unsigned kWorkSlots = 1024, kWorkStride = 16; // 64 bytes apart: one counter per cache line

static int *next_work_slot(nbldpc_code *c) { return c->d_work + (size_t)(c->work_next.fetch_add(1) % kWorkSlots) * kWorkStride; }

extern "C" const char *nbldpc_last_error(void) { return err_buf(); }

extern "C" int nbldpc_read_matrix(const char *path, int dims[5], int *vn_w, int *vn_cn, int *vn_gf, int *cn_w, int *cn_vn, int *cn_gf)
{
    if (!path || !dims) return fail(NBLDPC_EINVAL, "nbldpc_read_matrix: null argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(NBLDPC_EIO, "can not open file: %s", path);
    int N, M, q, dv, dc, v;
    if (fscanf(fp, "%d %d %d %d %d", &N, &M, &q, &dv, &dc) != 5 || N <= 0 || M <= 0 || q < 2 || dv <= 0 || dc <= 0) {
        fclose(fp);
        return fail(NBLDPC_EIO, "%s: bad header", path);
    }
    dims[0] = N; dims[1] = M; dims[2] = q; dims[3] = dv; dims[4] = dc;
    if (!vn_w) { fclose(fp); return NBLDPC_OK; }
    if (!vn_cn || !vn_gf || !cn_w || !cn_vn || !cn_gf) { fclose(fp); return fail(NBLDPC_EINVAL, "nbldpc_read_matrix: null array"); }
    auto bad = [&](const char *what) { fclose(fp); return fail(NBLDPC_EIO, "%s: %s", path, what); };
    for (int i = 0; i < N; i++)
        if (fscanf(fp, "%d", &vn_w[i]) != 1 || vn_w[i] < 0 || vn_w[i] > dv) return bad("bad column weight");
    for (int i = 0; i < M; i++)
        if (fscanf(fp, "%d", &cn_w[i]) != 1 || cn_w[i] < 0 || cn_w[i] > dc) return bad("bad row weight");
    for (int i = 0; i < N * dv; i++) { vn_cn[i] = -1; vn_gf[i] = 0; }
    for (int i = 0; i < M * dc; i++) { cn_vn[i] = -1; cn_gf[i] = 0; }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < vn_w[i]; j++) {
            if (fscanf(fp, "%d", &v) != 1 || v < 1 || v > M) return bad("bad check index");
            vn_cn[i * dv + j] = v - 1;
            if (fscanf(fp, "%d", &v) != 1 || v < 0 || v >= q) return bad("bad field element");
            vn_gf[i * dv + j] = v;
        }
    for (int i = 0; i < M; i++)
        for (int j = 0; j < cn_w[i]; j++) {
            if (fscanf(fp, "%d", &v) != 1 || v < 1 || v > N) return bad("bad variable index");
            cn_vn[i * dc + j] = v - 1;
            if (fscanf(fp, "%d", &v) != 1 || v < 0 || v >= q) return bad("bad field element");
            cn_gf[i * dc + j] = v;
        }
    fclose(fp);
    return NBLDPC_OK;
}

extern "C" int nbldpc_gf_load(const char *path, int q, unsigned *mul, unsigned *add, unsigned *inv)
{
    if (!path || !mul || !add || !inv || q < 2) return fail(NBLDPC_EINVAL, "nbldpc_gf_load: bad argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(NBLDPC_EIO, "Cannot open %s", path);
    char word[256];
    int c;
    bool ok = true;
    while ((c = fgetc(fp)) != EOF && c != '\n') {} // title line (GF.cpp:90)
    ok = ok && fscanf(fp, "%255s %255s", word, word) == 2;
    for (int i = 0; ok && i < q * q; i++) ok = fscanf(fp, "%u", &mul[i]) == 1;
    ok = ok && fscanf(fp, "%255s %255s", word, word) == 2;
    for (int i = 0; ok && i < q * q; i++) ok = fscanf(fp, "%u", &add[i]) == 1;
    ok = ok && fscanf(fp, "%255s %255s", word, word) == 2;
    for (int i = 0; ok && i < q; i++) ok = fscanf(fp, "%u", &inv[i]) == 1;
    fclose(fp);
    return ok ? NBLDPC_OK : fail(NBLDPC_EIO, "%s: truncated GF(%d) table file", path, q);
}

extern "C" int nbldpc_gf_generate(int q, unsigned poly, unsigned *mul, unsigned *add, unsigned *inv)
{
    int m = 0;
    while ((1 << m) < q) m++;
    if (q < 2 || (1 << m) != q || q > 4096 || !mul || !add || !inv) return fail(NBLDPC_EINVAL, "nbldpc_gf_generate: q=%d must be 2^m", q);
    if ((poly >> m) != 1u) return fail(NBLDPC_EINVAL, "primitive polynomial %u does not have degree %d", poly, m);
    for (int a = 0; a < q; a++)
        for (int b = 0; b < q; b++) {
            unsigned r = 0, x = (unsigned)a;
            for (int i = 0; i < m; i++) { // carry-less multiply, reduced on the fly
                if ((b >> i) & 1) r ^= x;
                x <<= 1;
                if (x & (unsigned)q) x ^= poly;
            }
            mul[a * q + b] = r;
            add[a * q + b] = (unsigned)(a ^ b);
        }
    inv[0] = 0; // GF/Arith.Table: inv[0] = 0
    for (int a = 1; a < q; a++) {
        inv[a] = 0;
        for (int b = 1; b < q; b++)
            if (mul[a * q + b] == 1) { inv[a] = (unsigned)b; break; }
        if (!inv[a]) return fail(NBLDPC_EINVAL, "polynomial %u is not irreducible over GF(2): %d has no inverse", poly, a);
    }
    return NBLDPC_OK;
}

static int up(void **dst, const void *src, size_t bytes)
{
    CLDPC_HIP(hipMalloc(dst, bytes), NBLDPC_ENOMEM);
    CLDPC_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), NBLDPC_EHIP);
    return NBLDPC_OK;
}

using TmmKernel = void (*)(TmmArgs);
static TmmKernel tmm_kernel(int q, bool layered)
{
    if (q == 64) return layered ? k_nb_tmm<64, true> : k_nb_tmm<64, false>;
    if (q == 32) return layered ? k_nb_tmm<32, true> : k_nb_tmm<32, false>;
    return layered ? k_nb_tmm<16, true> : k_nb_tmm<16, false>;
}

using NbKernel = void (*)(NbArgs);
static NbKernel nb_kernel(int q, int dv)
{
    if (q == 256) return k_nb_ems_wide<256, 1024>; // fields wider than a wavefront (nbldpc_wide_kernel.hpp)
    if (q == 128) return k_nb_ems_wide<128, 1024>;
    if (q == 64) return dv <= 2 ? k_nb_ems<64, 2, nb_threads(64)> : k_nb_ems<64, kNbMaxDv, nb_threads(64)>;
    if (q == 32) return dv <= 2 ? k_nb_ems<32, 2, nb_threads(32)> : k_nb_ems<32, kNbMaxDv, nb_threads(32)>;
    return dv <= 2 ? k_nb_ems<16, 2, nb_threads(16)> : k_nb_ems<16, kNbMaxDv, nb_threads(16)>;
}

static size_t nb_lds_bytes(int N, int M, int q, int dv, int dc)
{
    return ((size_t)N * dv * nb_pair_stride(q) + (size_t)(q + 1) * M * dc + N + 4) * sizeof(float) + (size_t)q * q +
           ((size_t)N + 2 * (size_t)N * dv + (size_t)M + 3 * (size_t)M * dc + 2) * sizeof(unsigned short) + // graph tables
           (size_t)N * dv;                                                                                       // edge-liveness bytes
}

extern "C" int nbldpc_code_create(int N, int M, int q, int dv, int dc, const int *vn_w, const int *vn_cn, const int *vn_gf,
                                  const int *cn_w, const int *cn_vn, const int *cn_gf, const unsigned *mul, nbldpc_code **out)
{
    if (!vn_w || !vn_cn || !vn_gf || !cn_w || !cn_vn || !cn_gf || !mul || !out) return fail(NBLDPC_EINVAL, "nbldpc_code_create: null argument");
    int m = 0;
    while ((1 << m) < q) m++;
    if (N <= 0 || M <= 0 || q < 4 || (1 << m) != q) return fail(NBLDPC_EINVAL, "bad dimensions N=%d M=%d q=%d", N, M, q);
    if (q > 256) return fail(NBLDPC_EUNSUPPORTED, "EMS kernels support q <= 256 (got %d)", q);
    if (dv > kNbMaxDv) return fail(NBLDPC_EUNSUPPORTED, "dvmax=%d (<= %d) unsupported", dv, kNbMaxDv);
    // the fused kernels (state of one frame in LDS, walk unrolled per row weight) when the code fits them, else the workspace kernel
    const bool fused_q = q == 16 || q == 32 || q == 64 || q == 128 || q == 256;
    const size_t lds = !fused_q ? 0 : q > 64 ? nb_wide_lds_bytes(N, M, q, dv, dc, nb_threads(q)) : nb_lds_bytes(N, M, q, dv, dc);
    const bool hbm = !fused_q || dc > kNbMaxW || M * dc > nb_threads(q) || M > nb_threads(q) || lds > 160 * 1024 ||
                     getenv("NBLDPC_FORCE_HBM") != nullptr; // tests: the workspace kernel on a code the fused kernels take
    if (hbm && dc > kNbHbmMaxDc) return fail(NBLDPC_EUNSUPPORTED, "dcmax=%d (<= %d) unsupported", dc, kNbHbmMaxDc);
    // cross indices: index_in_CN / index_in_VN (LDPC_Decoder.cpp:106-130), first match
    std::vector<int> vn_thr((size_t)N * dv, 0), cn_src((size_t)M * dc, 0);
    for (int i = 0; i < N; i++)
        for (int d = 0; d < vn_w[i]; d++) {
            const int cn = vn_cn[i * dv + d];
            if (cn < 0 || cn >= M) return fail(NBLDPC_EINVAL, "VN %d edge %d: check index %d out of range", i, d, cn);
            int slot = -1;
            for (int t = 0; t < cn_w[cn]; t++)
                if (cn_vn[cn * dc + t] == i) { slot = t; break; }
            if (slot < 0) return fail(NBLDPC_EINVAL, "index_in_CN error: VN %d not listed by CN %d", i, cn);
            // 0 is accepted: the reference reads its exponent-format files (LDPC_N576_K288_GF64_d1_exp.txt) as field elements and
            // decodes with the zeros in place (such an edge sends nothing and adds nothing to a syndrome); EMS does the same here.
            // The trellis decoders need the inverse of every coefficient (GFInverse(0) exits in the reference): not offered then.
            if (vn_gf[i * dv + d] < 0 || vn_gf[i * dv + d] >= q) return fail(NBLDPC_EINVAL, "VN %d edge %d: coefficient %d", i, d, vn_gf[i * dv + d]);
            vn_thr[i * dv + d] = cn * dc + slot;
        }
    for (int r = 0; r < M; r++) {
        if (cn_w[r] < 2 || cn_w[r] > dc) return fail(NBLDPC_EUNSUPPORTED, "row %d weight %d outside [2,%d]", r, cn_w[r], dc);
        for (int t = 0; t < cn_w[r]; t++) {
            const int vn = cn_vn[r * dc + t];
            if (vn < 0 || vn >= N) return fail(NBLDPC_EINVAL, "CN %d slot %d: variable index %d out of range", r, t, vn);
            int idx = -1;
            for (int d = 0; d < vn_w[vn]; d++)
                if (vn_cn[vn * dv + d] == r) { idx = d; break; }
            if (idx < 0) return fail(NBLDPC_EINVAL, "index_in_VN error: CN %d not listed by VN %d", r, vn);
            if (cn_gf[r * dc + t] != vn_gf[vn * dv + idx]) return fail(NBLDPC_EINVAL, "CN %d slot %d: coefficient differs between the two views", r, t);
            cn_src[r * dc + t] = vn * dv + idx;
        }
    }
    std::vector<unsigned char> mulb((size_t)q * q);
    for (int i = 0; i < q * q; i++) {
        if (mul[i] >= (unsigned)q) return fail(NBLDPC_EINVAL, "TableMultiply[%d] = %u outside GF(%d)", i, mul[i], q);
        mulb[i] = (unsigned char)mul[i];
    }
    nbldpc_code *c = new (std::nothrow) nbldpc_code;
    if (!c) return fail(NBLDPC_ENOMEM, "out of host memory");
    c->N = N; c->M = M; c->q = q; c->m = m; c->dv = dv; c->dc = dc; c->lds_bytes = lds; c->hbm = hbm;
    int r = 0;
    if (!r) r = up((void **)&c->d_vn_w, vn_w, (size_t)N * sizeof(int));
    if (!r) r = up((void **)&c->d_vn_thr, vn_thr.data(), vn_thr.size() * sizeof(int));
    if (!r) r = up((void **)&c->d_vn_gf, vn_gf, (size_t)N * dv * sizeof(int));
    if (!r) r = up((void **)&c->d_cn_w, cn_w, (size_t)M * sizeof(int));
    if (!r) r = up((void **)&c->d_cn_src, cn_src.data(), cn_src.size() * sizeof(int));
    if (!r) r = up((void **)&c->d_cn_gf, cn_gf, (size_t)M * dc * sizeof(int));
    if (!r) r = up((void **)&c->d_cn_vn, cn_vn, (size_t)M * dc * sizeof(int));
    if (!r) r = up((void **)&c->d_mul, mulb.data(), mulb.size());
    // trellis min-max decoders: inverse of every edge coefficient; dependency levels of the rows for the layered schedule
    // (level of a row = 1 + the highest level among the EARLIER rows that share a variable node with it)
    if (!r) {
        std::vector<int> hinv((size_t)M * dc, 0), level(M, 0), last(N, -1), order(M), lbegin;
        bool inv_ok = true;
        for (int row = 0; row < M; row++)
            for (int t = 0; t < cn_w[row]; t++)
                if (cn_gf[row * dc + t] == 0) { inv_ok = false; c->zero_coeff = 1; } // no inverse (the reference's GFInverse exits)
        for (int i = 0; i < M * dc; i++) {
            const int h = cn_gf[i];
            if (h <= 0) continue;
            int b = 0;
            for (int x = 1; x < q && !b; x++)
                if (mul[(size_t)h * q + x] == 1) b = x;
            if (!b) inv_ok = false;
            hinv[i] = b;
        }
        int levels = 0;
        for (int row = 0; row < M; row++) {
            int lv = 0;
            for (int t = 0; t < cn_w[row]; t++) lv = std::max(lv, last[cn_vn[row * dc + t]] + 1);
            for (int t = 0; t < cn_w[row]; t++) last[cn_vn[row * dc + t]] = lv;
            level[row] = lv;
            levels = std::max(levels, lv + 1);
        }
        for (int i = 0; i < M; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return level[x] < level[y]; });
        lbegin.assign(levels + 1, 0);
        for (int i = 0; i < M; i++) lbegin[level[i] + 1]++;
        for (int l = 0; l < levels; l++) lbegin[l + 1] += lbegin[l];
        c->levels = levels;
        c->tmm_ok = (q == 16 || q == 32 || q == 64) && inv_ok && dc <= kTmmMaxW && levels <= 63 && M <= kTmmThreads && // the trellis kernels keep a vector in one wave
                    tmm_lds_bytes(N, M, q, dv, dc, false) <= 160 * 1024;
        if (c->tmm_ok) {
            if (!r) r = up((void **)&c->d_cn_hinv, hinv.data(), hinv.size() * sizeof(int));
            if (!r) r = up((void **)&c->d_row_order, order.data(), order.size() * sizeof(int));
            if (!r) r = up((void **)&c->d_level_begin, lbegin.data(), lbegin.size() * sizeof(int));
        }
    }
    if (!r) {
        hipError_t e = hipSuccess;
        // the attribute belongs to the kernel, not to this code: set it to the CU's whole LDS once and for all, so that
        // creating a second code with a smaller state never lowers the cap under the first one
        if (hbm) e = hipFuncSetAttribute((const void *)k_nb_ems_hbm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); // GF table + the max arrays of one pass (the kernel has two static words too)
        else e = hipFuncSetAttribute((const void *)nb_kernel(q, dv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) r = fail(NBLDPC_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        int dev = 0, ncu = 0;
        const bool have_cu = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
        if (!r && !hbm && have_cu) {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)nb_kernel(q, dv), nb_threads(q), lds) == hipSuccess && occ > 0)
                c->persist_grid = ncu * occ;
        }
        // the two-frame pipeline (nbldpc_pipe_kernel.hpp): GF(64), column weights <= 2, no zero coefficient, the columns fit its A/S/B waves
        const int ncw = (M * dc + 63) / 64;
        if (!r && !hbm && have_cu && q == 64 && dv <= 2 && !c->zero_coeff && (nb_threads(q) / 64 - ncw) * kNbPipeCpw >= N && !getenv("NBLDPC_NO_PIPE")) {
            const size_t pl = ((lds + 15) & ~(size_t)15) + nb_pipe_extra_lds(N, M, dc);
            int occ = 0;
            if (pl <= 160 * 1024 && hipFuncSetAttribute((const void *)k_nb_ems2<64, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_nb_ems2<64, 1024>, 1024, pl) == hipSuccess && occ > 0) {
                c->pipe_grid = ncu * occ;
                c->pipe_lds = pl;
            }
        }
        for (int layered = 0; !r && c->tmm_ok && layered < 2; layered++) { // per-kernel attributes and grids once, not per decode call
            const size_t tl = tmm_lds_bytes(N, M, q, dv, dc, layered != 0);
            TmmKernel k = tmm_kernel(q, layered != 0);
            e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { r = fail(NBLDPC_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); break; }
            int occ = 0;
            if (have_cu && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k, kTmmThreads, tl) == hipSuccess && occ > 0)
                c->tmm_grid[layered] = ncu * occ;
        }
        if (!r) {
            e = hipMalloc((void **)&c->d_work, (size_t)kWorkSlots * kWorkStride * sizeof(int));
            if (e != hipSuccess) r = fail(NBLDPC_ENOMEM, "hipMalloc(frame counters): %s", hipGetErrorString(e));
        }
    }
    c->no_persist = getenv("NBLDPC_NO_PERSIST") != nullptr;
    if (r) { nbldpc_code_destroy(c); return r; }
    *out = c;
    return NBLDPC_OK;
}

extern "C" const char *nbldpc_last_kernel(const nbldpc_code *c) { return c ? c->last_kernel : "none"; }

extern "C" int nbldpc_code_destroy(nbldpc_code *c)
{
    if (!c) return NBLDPC_OK;
    void *ptrs[] = {c->d_vn_w, c->d_vn_thr, c->d_vn_gf, c->d_cn_w, c->d_cn_src, c->d_cn_gf, c->d_cn_vn, c->d_mul, c->d_cn_hinv, c->d_row_order, c->d_level_begin, c->d_work};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete c;
    return NBLDPC_OK;
}

extern "C" int nbldpc_ems_decode_batch(nbldpc_code *c, const float *Lch, int B, int Nm, int Nc, int maxIT, int maxdc_cfg, int *out,
                                       int *iters, int *ok, float *LLR, float *c2v, void *stream)
{
    if (!c || !Lch || !out || !iters || !ok) return fail(NBLDPC_EINVAL, "nbldpc_ems_decode_batch: null argument");
    if (B <= 0 || maxIT <= 0) return fail(NBLDPC_EINVAL, "B=%d maxIT=%d must be positive", B, maxIT);
    if (Nm < 1 || Nm > c->q || Nc < 0) return fail(NBLDPC_EINVAL, "EMS_Nm=%d must be in [1,%d], EMS_Nc=%d >= 0", Nm, c->q, Nc);
    NbArgs a;
    a.Lch = Lch; a.out = out; a.iters = iters; a.ok = ok; a.LLR = LLR; a.c2v = c2v;
    a.vn_w = c->d_vn_w; a.vn_thr = c->d_vn_thr; a.vn_gf = c->d_vn_gf;
    a.cn_w = c->d_cn_w; a.cn_src = c->d_cn_src; a.cn_gf = c->d_cn_gf; a.cn_vn = c->d_cn_vn; a.mul = c->d_mul;
    a.N = c->N; a.M = c->M; a.q = c->q; a.dv = c->dv; a.dc = c->dc; a.B = B; a.Nm = Nm; a.Nc = Nc; a.max_iter = maxIT;
    a.dcmax_cfg = maxdc_cfg > 0 ? maxdc_cfg : c->dc;
    a.zero_coeff = c->zero_coeff;
    hipStream_t st = (hipStream_t)stream;
    if (c->hbm) {
        // one workspace slot per workgroup, stream-ordered so that calls on different streams do not share it; at most 2 GiB
        const size_t slot = nb_hbm_slot_floats(c->N, c->M, c->q, c->dv, c->dc);
        const size_t cap = std::max<size_t>(1, ((size_t)2 << 30) / (slot * sizeof(float)));
        const int slots = (int)std::min<size_t>({(size_t)B, (size_t)1024, cap});
        void *ws = nullptr;
        CLDPC_HIP(hipMallocAsync(&ws, (size_t)slots * slot * sizeof(float), st), NBLDPC_ENOMEM);
        a.ws = (float *)ws;
        a.ws_stride = slot;
        a.work = next_work_slot(c);
        CLDPC_HIP(hipMemsetAsync(a.work, 0, sizeof(int), st), NBLDPC_EHIP);
        c->last_kernel = "k_nb_ems_hbm";
        hipLaunchKernelGGL(k_nb_ems_hbm, dim3(slots), dim3(kNbHbmThreads), nb_hbm_lds_bytes(c->q), st, a);
        const hipError_t le = hipGetLastError();
        CLDPC_HIP(hipFreeAsync(ws, st), NBLDPC_EHIP);
        CLDPC_HIP(le, NBLDPC_EHIP);
        return NBLDPC_OK;
    }
    if (c->pipe_grid > 0 && !c2v && B >= 2) {
        // two frames in flight per workgroup, frames from a counter (k_nb_ems2); L_c2v is not offered there
        a.work = next_work_slot(c);
        CLDPC_HIP(hipMemsetAsync(a.work, 0, sizeof(int), st), NBLDPC_EHIP);
        c->last_kernel = "k_nb_ems2";
        hipLaunchKernelGGL((k_nb_ems2<64, 1024>), dim3(std::min(c->pipe_grid, (B + 1) / 2)), dim3(1024), c->pipe_lds, st, a);
        CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
        return NBLDPC_OK;
    }
    if (c->persist_grid > 0 && B > c->persist_grid && !c->no_persist) {
        // persistent workgroups and a frame counter (k_nb_ems), zeroed stream-ordered; one ring slot per call in flight
        a.work = next_work_slot(c);
        CLDPC_HIP(hipMemsetAsync(a.work, 0, sizeof(int), st), NBLDPC_EHIP);
        c->last_kernel = c->q > 64 ? "k_nb_ems_wide (frames from a counter)" : "k_nb_ems (frames from a counter)";
        hipLaunchKernelGGL(nb_kernel(c->q, c->dv), dim3(c->persist_grid), dim3(nb_threads(c->q)), c->lds_bytes, st, a);
        CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
        return NBLDPC_OK;
    }
    c->last_kernel = c->q > 64 ? "k_nb_ems_wide" : "k_nb_ems";
    hipLaunchKernelGGL(nb_kernel(c->q, c->dv), dim3(B), dim3(nb_threads(c->q)), c->lds_bytes, st, a);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

extern "C" int nbldpc_tmm_decode_batch(nbldpc_code *c, const float *Lch, int B, int layered, int maxIT, int *out, int *iters, int *ok,
                                       float *LLR, float *c2v, void *stream)
{
    if (!c || !Lch || !out || !iters || !ok) return fail(NBLDPC_EINVAL, "nbldpc_tmm_decode_batch: null argument");
    if (B <= 0 || maxIT <= 0) return fail(NBLDPC_EINVAL, "B=%d maxIT=%d must be positive", B, maxIT);
    if (!c->tmm_ok) return fail(NBLDPC_EUNSUPPORTED, "trellis min-max kernel unavailable for this code (dcmax > %d, > 63 row levels, or state exceeds LDS)", kTmmMaxW);
    TmmArgs a;
    a.Lch = Lch; a.out = out; a.iters = iters; a.ok = ok; a.LLR = LLR; a.c2v = c2v;
    a.vn_w = c->d_vn_w; a.vn_thr = c->d_vn_thr; a.cn_w = c->d_cn_w; a.cn_src = c->d_cn_src; a.cn_gf = c->d_cn_gf; a.cn_vn = c->d_cn_vn;
    a.cn_hinv = c->d_cn_hinv; a.row_order = c->d_row_order; a.level_begin = c->d_level_begin; a.mul = c->d_mul;
    a.N = c->N; a.M = c->M; a.q = c->q; a.dv = c->dv; a.dc = c->dc; a.B = B; a.max_iter = maxIT; a.levels = c->levels;
    const size_t lds = tmm_lds_bytes(c->N, c->M, c->q, c->dv, c->dc, layered != 0);
    TmmKernel k = tmm_kernel(c->q, layered != 0);
    hipStream_t st = (hipStream_t)stream;
    const int pgrid = c->tmm_grid[layered != 0];
    c->last_kernel = layered ? "k_nb_tmm (layered)" : "k_nb_tmm";
    if (pgrid > 0 && B > pgrid && !c->no_persist) { // persistent workgroups and a frame counter, as in nbldpc_ems_decode_batch
        c->last_kernel = layered ? "k_nb_tmm (layered, frames from a counter)" : "k_nb_tmm (frames from a counter)";
        a.work = next_work_slot(c);
        CLDPC_HIP(hipMemsetAsync(a.work, 0, sizeof(int), st), NBLDPC_EHIP);
        hipLaunchKernelGGL(k, dim3(pgrid), dim3(kTmmThreads), lds, st, a);
        CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
        return NBLDPC_OK;
    }
    hipLaunchKernelGGL(k, dim3(B), dim3(kTmmThreads), lds, st, a);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

extern "C" int nbldpc_demodulate_bpsk(const nbldpc_code *c, const float *rx, float sigma, int B, float *Lch, void *stream)
{
    if (!c) return fail(NBLDPC_EINVAL, "nbldpc_demodulate_bpsk: bad argument");
    return nbldpc_demodulate_bpsk_nq(c->N, c->q, rx, sigma, B, Lch, stream);
}

extern "C" int nbldpc_demodulate_bpsk_nq(int N, int q, const float *rx, float sigma, int B, float *Lch, void *stream)
{
    int m = 0;
    while ((1 << m) < q) m++;
    if (N <= 0 || q < 2 || (1 << m) != q || !rx || !Lch || B <= 0 || !(sigma > 0)) return fail(NBLDPC_EINVAL, "nbldpc_demodulate_bpsk: bad argument");
    const size_t total = (size_t)B * N * (q - 1);
    hipLaunchKernelGGL(k_nb_demod_bpsk, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rx, sigma, B, N, q, m, Lch);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

extern "C" int nbldpc_demodulate_qam(const nbldpc_code *c, const float *rx, const float *con, float sigma, int B, float *Lch, void *stream)
{
    if (!c) return fail(NBLDPC_EINVAL, "nbldpc_demodulate_qam: bad argument");
    return nbldpc_demodulate_qam_nq(c->N, c->q, rx, con, sigma, B, Lch, stream);
}

extern "C" int nbldpc_demodulate_qam_nq(int N, int q, const float *rx, const float *con, float sigma, int B, float *Lch, void *stream)
{
    if (N <= 0 || q < 2 || !rx || !con || !Lch || B <= 0 || !(sigma > 0)) return fail(NBLDPC_EINVAL, "nbldpc_demodulate_qam: bad argument");
    const size_t total = (size_t)B * N * (q - 1);
    hipLaunchKernelGGL(k_nb_demod_qam, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rx, con, sigma, B, N, q, Lch);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

extern "C" int nbldpc_read_constellation(const char *path, int n_points, float *con)
{
    if (!path || !con || n_points <= 0) return fail(NBLDPC_EINVAL, "nbldpc_read_constellation: bad argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(NBLDPC_EIO, "can not open file: %s", path);
    char tmp[100];
    for (int k = 0; k < n_points; k++) { // "Point: <idx> Real: <x> Imag: <y>" (Simulation.cpp:326-334)
        int idx = -1;
        float re = 0, im = 0;
        const bool ok = fscanf(fp, "%99s", tmp) == 1 && fscanf(fp, "%d", &idx) == 1 && fscanf(fp, "%99s", tmp) == 1 && fscanf(fp, "%f", &re) == 1 &&
                        fscanf(fp, "%99s", tmp) == 1 && fscanf(fp, "%f", &im) == 1;
        if (!ok || idx < 0 || idx >= n_points) {
            fclose(fp);
            return fail(NBLDPC_EIO, "%s: record %d is not 'Point: <0..%d> Real: <x> Imag: <y>'", path, k, n_points - 1);
        }
        con[2 * idx] = re;
        con[2 * idx + 1] = im;
    }
    fclose(fp);
    return NBLDPC_OK;
}

extern "C" int nbldpc_statistic(const nbldpc_code *c, const int *out, const int *iters, const int *ok, const int *cw, int B,
                                long long *counters, void *stream)
{
    if (!c || !out || !iters || !ok || !cw || !counters || B <= 0) return fail(NBLDPC_EINVAL, "nbldpc_statistic: bad argument");
    hipLaunchKernelGGL(k_nb_statistic, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, iters, ok, cw, B, c->N, counters);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

namespace {
inline float random_module(int *seed) // src/LDPC_Encoder.cpp:70-79
{
    seed[0] = (seed[0] * 249) % 61967;
    seed[1] = (seed[1] * 251) % 63443;
    seed[2] = (seed[2] * 252) % 63599;
    float t = ((float)seed[0] / 61967.0f) + ((float)seed[1] / 63443.0f) + ((float)seed[2] / 63599.0f);
    t -= (int)t;
    return t;
}
} // namespace

extern "C" float nbldpc_random_module(int seed[3]) { return random_module(seed); }

extern "C" int nbldpc_awgn_channel_host(int seed[3], float sigma, const int *cw, int N, int m, float *rx)
{
    if (!seed || !cw || !rx || N <= 0 || m <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_host: bad argument");
    const double two_pi = 2 * 3.1415926; // define.h:56
    for (int i = 0; i < N * m; i++) {
        const float tx = ((cw[i / m] >> (i % m)) & 1) ? -1.0f : 1.0f; // main.cu:203-209 + Constellation/BPSK.txt
        float u1 = random_module(seed), u2 = random_module(seed);
        const float amp = std::sqrt(-2.0f * std::log(1.0f - u1));
        rx[i] = (float)((double)sigma * std::cos(two_pi * (double)u2) * (double)amp + (double)tx);
        (void)random_module(seed); // the Image part draws two more numbers (LDPC_Encoder.cpp:62-66)
        (void)random_module(seed);
    }
    return NBLDPC_OK;
}

extern "C" int nbldpc_awgn_channel_host_qam(int seed[3], float sigma, const int *cw, int N, const float *con, int n_points, float *rx)
{
    if (!seed || !cw || !con || !rx || N <= 0 || n_points <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_host_qam: bad argument");
    const double two_pi = 2 * 3.1415926; // define.h:56
    for (int i = 0; i < N; i++) {
        if (cw[i] < 0 || cw[i] >= n_points) return fail(NBLDPC_EINVAL, "CodeWord_sym[%d]=%d outside the constellation", i, cw[i]);
        for (int c = 0; c < 2; c++) { // Real, then Image: two draws each (LDPC_Encoder.cpp:56-66), Modulate :22-26
            float u1 = random_module(seed), u2 = random_module(seed);
            const float amp = std::sqrt(-2.0f * std::log(1.0f - u1));
            rx[2 * i + c] = (float)((double)sigma * std::cos(two_pi * (double)u2) * (double)amp + (double)con[2 * cw[i] + c]);
        }
    }
    return NBLDPC_OK;
}

namespace {
constexpr unsigned kNbA[3] = {249u, 251u, 252u}, kNbM[3] = {61967u, 63443u, 63599u}; // src/LDPC_Encoder.cpp:72-74

// a^k mod m; the moduli are prime, so the exponent reduces mod (m - 1) and everything fits 32 bits (see bldpc_channel.hip)
__host__ __device__ inline unsigned nb_powmod(unsigned a, unsigned long long k, unsigned m)
{
    unsigned e = (unsigned)(k % (unsigned long long)(m - 1));
    unsigned r = 1, b = a % m;
    while (e) {
        if (e & 1) r = (r * b) % m;
        b = (b * b) % m;
        e >>= 1;
    }
    return r;
}

// One thread per (frame b, run of kNbRun consecutive bits): jump to draw 4*(b*N*m + i0), then step as RandomModule does.
constexpr int kNbRun = 16;
__global__ __launch_bounds__(256) void k_nb_awgn(unsigned s0, unsigned s1, unsigned s2, float sigma, const int *cw, int N, int m, int B, float *rx)
{
    const int runs = (N * m + kNbRun - 1) / kNbRun;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)B * runs) return;
    const int b = (int)(id / runs), i0 = (int)(id - (long long)b * runs) * kNbRun;
    const unsigned long long k = 4ull * ((unsigned long long)b * N * m + i0);
    unsigned s[3] = {s0, s1, s2};
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = (unsigned)(((unsigned long long)s[i] * nb_powmod(kNbA[i], k, kNbM[i])) % kNbM[i]);
    const double two_pi = 2 * 3.1415926; // define.h:56
    for (int i = i0; i < min(N * m, i0 + kNbRun); i++) {
        float u[4];
#pragma unroll
        for (int d = 0; d < 4; d++) { // Real part: draws 1-2; Image part (unused for BPSK): draws 3-4 (LDPC_Encoder.cpp:59-66)
#pragma unroll
            for (int j = 0; j < 3; j++) s[j] = (s[j] * kNbA[j]) % kNbM[j];
            // x / m for an integer 0 <= x < m, m an odd prime below 2^16: the correctly rounded float quotient equals the double
            // product x * (1/m) rounded to float (x/m is at least 2^-40 away, relatively, from every float rounding boundary;
            // all 3 x 63 599 cases checked in tests/test_host_cpu.py) -- three conversions and a multiply instead of a division
            float t = (float)((double)(int)s[0] * (1.0 / 61967.0)) + (float)((double)(int)s[1] * (1.0 / 63443.0)) + (float)((double)(int)s[2] * (1.0 / 63599.0));
            t -= (int)t;
            u[d] = t;
        }
        const float tx = ((cw[i / m] >> (i % m)) & 1) ? -1.0f : 1.0f; // main.cu:203-209 + Constellation/BPSK.txt
        const float amp = sqrtf(-2.0f * logf(1.0f - u[0]));
        rx[(size_t)b * N * m + i] = (float)((double)sigma * cos(two_pi * (double)u[1]) * (double)amp + (double)tx);
    }
}
} // namespace

extern "C" int nbldpc_awgn_channel_device(int seed[3], float sigma, const int *cw, int N, int m, int B, float *rx, void *stream)
{
    if (!seed || !cw || !rx || N <= 0 || m <= 0 || B <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_device: bad argument");
    for (int i = 0; i < 3; i++)
        if (seed[i] < 0 || (unsigned)seed[i] >= kNbM[i]) return fail(NBLDPC_EINVAL, "seed[%d]=%d outside [0,%u)", i, seed[i], kNbM[i]);
    const long long threads = (long long)B * ((N * m + kNbRun - 1) / kNbRun);
    hipLaunchKernelGGL(k_nb_awgn, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned)seed[0], (unsigned)seed[1],
                       (unsigned)seed[2], sigma, cw, N, m, B, rx);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    const unsigned long long draws = 4ull * (unsigned long long)N * m * B;
    for (int i = 0; i < 3; i++) seed[i] = (int)(((unsigned long long)seed[i] * nb_powmod(kNbA[i], draws, kNbM[i])) % kNbM[i]);
    return NBLDPC_OK;
}

namespace {
// QAM: one thread per (frame b, symbol i): jump to draw 4*(b*N + i), Real part from draws 1-2, Image part from draws 3-4.
__global__ __launch_bounds__(256) void k_nb_awgn_qam(unsigned s0, unsigned s1, unsigned s2, float sigma, const int *cw, const float *con, int N, int B,
                                                    float *rx)
{
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)B * N) return;
    const int i = (int)(id % N);
    unsigned s[3] = {s0, s1, s2};
#pragma unroll
    for (int j = 0; j < 3; j++) s[j] = (unsigned)(((unsigned long long)s[j] * nb_powmod(kNbA[j], 4ull * (unsigned long long)id, kNbM[j])) % kNbM[j]);
    const double two_pi = 2 * 3.1415926; // define.h:56
    float u[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
        for (int j = 0; j < 3; j++) s[j] = (s[j] * kNbA[j]) % kNbM[j];
        // x / m for an integer 0 <= x < m, m an odd prime below 2^16: the correctly rounded float quotient equals the double
            // product x * (1/m) rounded to float (x/m is at least 2^-40 away, relatively, from every float rounding boundary;
            // all 3 x 63 599 cases checked in tests/test_host_cpu.py) -- three conversions and a multiply instead of a division
            float t = (float)((double)(int)s[0] * (1.0 / 61967.0)) + (float)((double)(int)s[1] * (1.0 / 63443.0)) + (float)((double)(int)s[2] * (1.0 / 63599.0));
        t -= (int)t;
        u[d] = t;
    }
    const int sym = cw[i];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float amp = sqrtf(-2.0f * logf(1.0f - u[2 * c]));
        rx[2 * id + c] = (float)((double)sigma * cos(two_pi * (double)u[2 * c + 1]) * (double)amp + (double)con[2 * sym + c]);
    }
}
} // namespace

extern "C" int nbldpc_awgn_channel_device_qam(int seed[3], float sigma, const int *cw, int N, const float *con, int B, float *rx, void *stream)
{
    if (!seed || !cw || !con || !rx || N <= 0 || B <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_device_qam: bad argument");
    for (int i = 0; i < 3; i++)
        if (seed[i] < 0 || (unsigned)seed[i] >= kNbM[i]) return fail(NBLDPC_EINVAL, "seed[%d]=%d outside [0,%u)", i, seed[i], kNbM[i]);
    const long long threads = (long long)B * N;
    hipLaunchKernelGGL(k_nb_awgn_qam, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned)seed[0],
                       (unsigned)seed[1], (unsigned)seed[2], sigma, cw, con, N, B, rx);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    const unsigned long long draws = 4ull * (unsigned long long)N * B;
    for (int i = 0; i < 3; i++) seed[i] = (int)(((unsigned long long)seed[i] * nb_powmod(kNbA[i], draws, kNbM[i])) % kNbM[i]);
    return NBLDPC_OK;
}

// ---- the reference's AWGNChannel_CPU as it is declared: noise on a modulated frame (any constellation) -------------------------

extern "C" int nbldpc_awgn_channel_host_sym(int seed[3], float sigma, const float *tx, int len, float *rx)
{
    if (!seed || !tx || !rx || len <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_host_sym: bad argument");
    const double two_pi = 2 * 3.1415926; // define.h:56
    for (int i = 0; i < 2 * len; i++) { // sample i/2: Real from draws 1-2, Image from draws 3-4 (LDPC_Encoder.cpp:53-67)
        float u1 = random_module(seed), u2 = random_module(seed);
        const float amp = std::sqrt(-2.0f * std::log(1.0f - u1));
        rx[i] = (float)((double)sigma * std::cos(two_pi * (double)u2) * (double)amp + (double)tx[i]);
    }
    return NBLDPC_OK;
}

namespace {
// One thread per (frame b, run of kNbRun consecutive samples): jump to draw 4*(b*len + i0), then step as RandomModule does.
template <bool REAL_ONLY>
__global__ __launch_bounds__(256) void k_nb_awgn_sym(unsigned s0, unsigned s1, unsigned s2, float sigma, const float *tx, int len, int B, float *rx)
{
    const int runs = (len + kNbRun - 1) / kNbRun;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)B * runs) return;
    const int b = (int)(id / runs), i0 = (int)(id - (long long)b * runs) * kNbRun;
    const unsigned long long k = 4ull * ((unsigned long long)b * len + i0);
    unsigned s[3] = {s0, s1, s2};
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = (unsigned)(((unsigned long long)s[i] * nb_powmod(kNbA[i], k, kNbM[i])) % kNbM[i]);
    const double two_pi = 2 * 3.1415926; // define.h:56
    for (int i = i0; i < min(len, i0 + kNbRun); i++) {
        float u[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int j = 0; j < 3; j++) s[j] = (s[j] * kNbA[j]) % kNbM[j];
            // correctly rounded x / m as a double product (see k_nb_awgn)
            float t = (float)((double)(int)s[0] * (1.0 / 61967.0)) + (float)((double)(int)s[1] * (1.0 / 63443.0)) + (float)((double)(int)s[2] * (1.0 / 63599.0));
            t -= (int)t;
            u[d] = t;
        }
        const float a0 = sqrtf(-2.0f * logf(1.0f - u[0]));
        const float re = (float)((double)sigma * cos(two_pi * (double)u[1]) * (double)a0 + (double)tx[2 * i]);
        if (REAL_ONLY) {
            rx[(size_t)b * len + i] = re;
        } else {
            const float a1 = sqrtf(-2.0f * logf(1.0f - u[2]));
            const float im = (float)((double)sigma * cos(two_pi * (double)u[3]) * (double)a1 + (double)tx[2 * i + 1]);
            *reinterpret_cast<float2 *>(rx + ((size_t)b * len + i) * 2) = make_float2(re, im);
        }
    }
}
} // namespace

extern "C" int nbldpc_awgn_channel_device_sym(int seed[3], float sigma, const float *tx, int len, int B, int real_only, float *rx, void *stream)
{
    if (!seed || !tx || !rx || len <= 0 || B <= 0) return fail(NBLDPC_EINVAL, "nbldpc_awgn_channel_device_sym: bad argument");
    for (int i = 0; i < 3; i++)
        if (seed[i] < 0 || (unsigned)seed[i] >= kNbM[i]) return fail(NBLDPC_EINVAL, "seed[%d]=%d outside [0,%u)", i, seed[i], kNbM[i]);
    const long long threads = (long long)B * ((len + kNbRun - 1) / kNbRun);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (real_only) hipLaunchKernelGGL(k_nb_awgn_sym<true>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned)seed[0], (unsigned)seed[1], (unsigned)seed[2], sigma, tx, len, B, rx);
    else hipLaunchKernelGGL(k_nb_awgn_sym<false>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned)seed[0], (unsigned)seed[1], (unsigned)seed[2], sigma, tx, len, B, rx);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return nbldpc_seed_jump(seed, 4ull * (unsigned long long)len * B);
}

extern "C" int nbldpc_seed_jump(int seed[3], unsigned long long draws)
{
    if (!seed) return fail(NBLDPC_EINVAL, "nbldpc_seed_jump: null seed");
    for (int i = 0; i < 3; i++)
        if (seed[i] < 0 || (unsigned)seed[i] >= kNbM[i]) return fail(NBLDPC_EINVAL, "seed[%d]=%d outside [0,%u)", i, seed[i], kNbM[i]);
    for (int i = 0; i < 3; i++) seed[i] = (int)(((unsigned long long)seed[i] * nb_powmod(kNbA[i], draws, kNbM[i])) % kNbM[i]);
    return NBLDPC_OK;
}

namespace {
// errs[b] = number of symbols of frame b that differ from the transmitted word (Statistic, Simulation.cpp:264-267): one wave per frame
__global__ __launch_bounds__(256) void k_nb_frame_errors(const int *out, const int *cw, int B, int N, int *errs)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    int e = 0;
    for (int i = lane; i < N; i += 64) e += out[(size_t)b * N + i] != cw[i];
#pragma unroll
    for (int o = 32; o; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) errs[b] = e;
}
} // namespace

extern "C" int nbldpc_frame_errors(const nbldpc_code *c, const int *out, const int *cw, int B, int *errs, void *stream)
{
    if (!c || !out || !cw || !errs || B <= 0) return fail(NBLDPC_EINVAL, "nbldpc_frame_errors: bad argument");
    hipLaunchKernelGGL(k_nb_frame_errors, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, cw, B, c->N, errs);
    CLDPC_HIP(hipGetLastError(), NBLDPC_EHIP);
    return NBLDPC_OK;
}

extern "C" float nbldpc_sigma(float snr, int snrtype, int n_qam, float rate)
{
    if (snrtype == 0) return (float)std::sqrt(0.5 / (std::log((double)n_qam) / std::log(2.0) * rate * std::pow(10.0, (double)(snr / 10.0))));
    return (float)std::sqrt(0.5 / (std::log((double)n_qam) / std::log(2.0) * std::pow(10.0, (double)(snr / 10.0))));
}
