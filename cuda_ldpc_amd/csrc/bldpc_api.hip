// bldpc_api.hip -- host side of the binary QC-LDPC decoder behind include/bldpc.h.
//
// Graph builders restate bldpc_实习/Simulation.cu:292-387 (Get_H, Transform_H);
// bldpc_decode replaces LDPC_Decoder_GPU (LDPC_Decoder.cu:23-164): same inputs,
// same D/iteraTime outputs, but no per-call allocation, no per-iteration
// device->host copy of D and no host-side termination loop.
#include "../../include/bldpc.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "bldpc_qc_kernel.hpp"
#include "bldpc_table_kernels.hpp"
#include "common.hpp"

using namespace cldpc;

struct bldpc_code {
    int J = 0, L = 0, Z = 0, N = 0, M = 0, K = 0, Wc = 0, Wv = 0, nnz = 0, levels = 1;
    bool has_qc = false;
    std::vector<int> H, wc, wv;
    std::vector<int> level_begin; // node_list range of each VN level, size levels+1
    int *d_addr = nullptr, *d_node_list = nullptr;
    unsigned char *d_wv_blk = nullptr, *d_wc_blk = nullptr;
    DevBuf rq, bad, cnt, bits, yg, errs, itw;
    int *h_cnt = nullptr; // pinned
    QcPlan qc;            // fused LDS kernel description (frames_per_wg == 0: unavailable)
    const char *last_kernel = "none";
    bool profiling = false;
    // profiling: a ring of event pairs, one pair per decode call, so that a bench can average the dominant kernel over ALL of its
    // timed steps without synchronising after each (bldpc_kernel_ms_mean); ev0/ev1 point at the pair of the call in progress
    static constexpr unsigned kEvRing = 64;
    hipEvent_t evr[2 * kEvRing] = {};
    unsigned ev_n = 0; // decode calls recorded since the last bldpc_kernel_ms_mean
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" const char *bldpc_last_error(void) { return err_buf(); }

extern "C" int bldpc_read_blockh(const char *path, int J, int L, int *H, int *wc, int *wv)
{
    if (!path || !H || !wc || !wv || J <= 0 || L <= 0) return fail(BLDPC_EINVAL, "bldpc_read_blockh: bad argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(BLDPC_EIO, "can not open file: %s", path);
    for (int i = 0; i < J * L; i++) {
        int v;
        if (fscanf(fp, "%d", &v) != 1) {
            fclose(fp);
            return fail(BLDPC_EIO, "%s: expected %d shifts, got %d", path, J * L, i);
        }
        H[i] = v;
    }
    fclose(fp);
    std::fill(wc, wc + J + 1, 0);
    std::fill(wv, wv + L + 1, 0);
    for (int j = 0; j < J; j++) {
        for (int l = 0; l < L; l++) wc[j] += (H[j * L + l] != -1);
        wc[J] = std::max(wc[J], wc[j]);
    }
    for (int l = 0; l < L; l++) {
        for (int j = 0; j < J; j++) wv[l] += (H[j * L + l] != -1);
        wv[L] = std::max(wv[L], wv[l]);
    }
    return BLDPC_OK;
}

extern "C" int bldpc_transform_h(const int *H, int J, int L, int Z, const int *wc, const int *wv, int *addr, int as_written)
{
    if (!H || !wc || !wv || !addr || J <= 0 || L <= 0 || Z <= 0) return fail(BLDPC_EINVAL, "bldpc_transform_h: bad argument");
    const int Wv = wv[L], Wc = wc[J];
    std::fill(addr, addr + (size_t)L * Z * Wv, -1);
    for (int l = 0; l < L; l++) {
        int k = 0;
        for (int j = 0; j < J; j++) {
            const int s = H[j * L + l];
            if (s == -1) continue;
            if (s < 0 || s >= Z) return fail(BLDPC_EINVAL, "shift %d of block (%d,%d) outside [0,%d)", s, j, l, Z);
            int pos = 0; // ordinal of this block among the non-zero blocks of row j
            for (int t = 0; t < l; t++) pos += (H[j * L + t] != -1);
            for (int c = 0; c < Z; c++) {
                int row;
                if (as_written) // Simulation.cu:380, kept literally: the else branch is `c`
                    row = (((Z - s) % Z + c) >= Z) ? (Z - s) % Z + c - Z : c;
                else
                    row = (c - s + Z) % Z;
                addr[((size_t)l * Z + c) * Wv + k] = (j * Z + row) * Wc + pos;
            }
            k++;
        }
    }
    return BLDPC_OK;
}

static int upload(void **dst, const void *src, size_t bytes)
{
    CLDPC_HIP(hipMalloc(dst, bytes), BLDPC_ENOMEM);
    CLDPC_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), BLDPC_EHIP);
    return BLDPC_OK;
}

// Shared tail of the two constructors: validate the table, level-schedule the
// variable nodes, upload.
static int finish_code(bldpc_code *c, const int *addr)
{
    const int N = c->N, Wv = c->Wv, slots = c->M * c->Wc;
    if (c->Wv > kMaxWv || c->Wc > kMaxWc || c->Wv < 1 || c->Wc < 2)
        return fail(BLDPC_EUNSUPPORTED, "block weights Wc=%d Wv=%d outside supported [2,%d] / [1,%d]", c->Wc, c->Wv, kMaxWc, kMaxWv);
    std::vector<int> last(slots, 0), level(N, 1);
    int levels = 1;
    for (int n = 0; n < N; n++) {
        const int w = c->wv[n / c->Z];
        int lv = 1;
        for (int i = 0; i < w; i++) {
            const int s = addr[(size_t)n * Wv + i];
            if (s < 0 || s >= slots) return fail(BLDPC_EINVAL, "Address_Variablenode[%d][%d] = %d outside [0,%d)", n, i, s, slots);
            lv = std::max(lv, last[s] + 1);
        }
        for (int i = 0; i < w; i++) last[addr[(size_t)n * Wv + i]] = lv;
        level[n] = lv;
        levels = std::max(levels, lv);
    }
    c->levels = levels;
    std::vector<int> order(N);
    for (int n = 0; n < N; n++) order[n] = n;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
    c->level_begin.assign(levels + 1, 0);
    for (int n = 0; n < N; n++) c->level_begin[level[n]]++;
    for (int l = 1; l <= levels; l++) c->level_begin[l] += c->level_begin[l - 1];
    std::vector<unsigned char> wvb(c->L), wcb(c->J);
    for (int l = 0; l < c->L; l++) wvb[l] = (unsigned char)c->wv[l];
    for (int j = 0; j < c->J; j++) wcb[j] = (unsigned char)c->wc[j];
    int r;
    if ((r = upload((void **)&c->d_addr, addr, (size_t)N * Wv * sizeof(int)))) return r;
    if (levels > 1 && (r = upload((void **)&c->d_node_list, order.data(), (size_t)N * sizeof(int)))) return r;
    if ((r = upload((void **)&c->d_wv_blk, wvb.data(), wvb.size()))) return r;
    if ((r = upload((void **)&c->d_wc_blk, wcb.data(), wcb.size()))) return r;
    CLDPC_HIP(hipHostMalloc((void **)&c->h_cnt, sizeof(int), hipHostMallocDefault), BLDPC_ENOMEM);
    return BLDPC_OK;
}

static int set_dims(bldpc_code *c, int J, int L, int Z)
{
    if (J <= 0 || L <= 0 || Z <= 0 || J >= L) return fail(BLDPC_EINVAL, "need 0 < J < L and Z > 0 (J=%d L=%d Z=%d)", J, L, Z);
    if ((long long)L * Z > (1 << 24)) return fail(BLDPC_EUNSUPPORTED, "N = %lld too large", (long long)L * Z);
    c->J = J; c->L = L; c->Z = Z;
    c->N = L * Z; c->M = J * Z; c->K = c->N - c->M;
    return BLDPC_OK;
}

extern "C" int bldpc_code_create_qc(int J, int L, int Z, const int *H, bldpc_code **out)
{
    if (!H || !out) return fail(BLDPC_EINVAL, "bldpc_code_create_qc: null argument");
    bldpc_code *c = new (std::nothrow) bldpc_code;
    if (!c) return fail(BLDPC_ENOMEM, "out of host memory");
    int r = set_dims(c, J, L, Z);
    if (r) { delete c; return r; }
    c->H.assign(H, H + J * L);
    c->wc.assign(J + 1, 0);
    c->wv.assign(L + 1, 0);
    for (int j = 0; j < J; j++)
        for (int l = 0; l < L; l++)
            if (H[j * L + l] != -1) {
                if (H[j * L + l] < 0 || H[j * L + l] >= Z) { delete c; return fail(BLDPC_EINVAL, "shift %d outside [0,%d)", H[j * L + l], Z); }
                c->wc[j]++; c->wv[l]++; c->nnz++;
            }
    for (int j = 0; j < J; j++) c->wc[J] = std::max(c->wc[J], c->wc[j]);
    for (int l = 0; l < L; l++) c->wv[L] = std::max(c->wv[L], c->wv[l]);
    c->Wc = c->wc[J]; c->Wv = c->wv[L];
    std::vector<int> addr((size_t)c->N * std::max(c->Wv, 1));
    if (c->Wv < 1) { delete c; return fail(BLDPC_EINVAL, "empty matrix"); }
    r = bldpc_transform_h(H, J, L, Z, c->wc.data(), c->wv.data(), addr.data(), 0);
    if (!r) r = finish_code(c, addr.data());
    if (!r) {
        c->has_qc = true;
        r = qc_plan_build(&c->qc, J, L, Z, c->H.data());
    }
    if (r) { bldpc_code_destroy(c); return r; }
    *out = c;
    return BLDPC_OK;
}

extern "C" int bldpc_code_create_table(int J, int L, int Z, const int *wc, const int *wv, const int *addr, bldpc_code **out)
{
    if (!wc || !wv || !addr || !out) return fail(BLDPC_EINVAL, "bldpc_code_create_table: null argument");
    bldpc_code *c = new (std::nothrow) bldpc_code;
    if (!c) return fail(BLDPC_ENOMEM, "out of host memory");
    int r = set_dims(c, J, L, Z);
    if (r) { delete c; return r; }
    c->wc.assign(wc, wc + J + 1);
    c->wv.assign(wv, wv + L + 1);
    c->Wc = wc[J]; c->Wv = wv[L];
    for (int j = 0; j < J; j++) {
        if (wc[j] < 0 || wc[j] > c->Wc) { delete c; return fail(BLDPC_EINVAL, "Weight_Checknode[%d]=%d > max %d", j, wc[j], c->Wc); }
        c->nnz += wc[j];
    }
    for (int l = 0; l < L; l++)
        if (wv[l] < 0 || wv[l] > c->Wv) { delete c; return fail(BLDPC_EINVAL, "Weight_Variablenode[%d]=%d > max %d", l, wv[l], c->Wv); }
    r = finish_code(c, addr);
    if (r) { bldpc_code_destroy(c); return r; }
    *out = c;
    return BLDPC_OK;
}

extern "C" int bldpc_code_destroy(bldpc_code *c)
{
    if (!c) return BLDPC_OK;
    if (c->d_addr) (void)hipFree(c->d_addr);
    if (c->d_node_list) (void)hipFree(c->d_node_list);
    if (c->d_wv_blk) (void)hipFree(c->d_wv_blk);
    if (c->d_wc_blk) (void)hipFree(c->d_wc_blk);
    if (c->h_cnt) (void)hipHostFree(c->h_cnt);
    c->rq.release(); c->bad.release(); c->cnt.release(); c->bits.release(); c->yg.release(); c->errs.release(); c->itw.release();
    qc_plan_release(&c->qc);
    for (hipEvent_t e : c->evr)
        if (e) (void)hipEventDestroy(e);
    delete c;
    return BLDPC_OK;
}

extern "C" int bldpc_code_dims(const bldpc_code *c, int dims[8])
{
    if (!c || !dims) return fail(BLDPC_EINVAL, "bldpc_code_dims: null argument");
    dims[0] = c->N; dims[1] = c->M; dims[2] = c->K; dims[3] = c->Wc; dims[4] = c->Wv; dims[5] = c->nnz;
    dims[6] = c->levels; dims[7] = c->has_qc ? c->qc.frames_per_wg : 0;
    return BLDPC_OK;
}

extern "C" const char *bldpc_last_kernel(const bldpc_code *c) { return c ? c->last_kernel : "none"; }

extern "C" int bldpc_set_profiling(bldpc_code *c, int enable)
{
    if (!c) return fail(BLDPC_EINVAL, "bldpc_set_profiling: null code");
    if (enable && !c->evr[0]) {
        for (unsigned i = 0; i < 2 * bldpc_code::kEvRing; i++) CLDPC_HIP(hipEventCreate(&c->evr[i]), BLDPC_EHIP);
    }
    c->profiling = enable != 0;
    c->ev_n = 0;
    c->ev0 = c->ev1 = nullptr;
    return BLDPC_OK;
}

extern "C" int bldpc_kernel_ms_mean(bldpc_code *c, float *mean_ms, int *launches)
{
    if (!c || !mean_ms || !c->evr[0] || !c->ev_n) return fail(BLDPC_EINVAL, "bldpc_kernel_ms_mean: no profiled decode call since the last one");
    const unsigned n = std::min(c->ev_n, bldpc_code::kEvRing);
    double sum = 0;
    for (unsigned k = 0; k < n; k++) {
        const unsigned slot = (c->ev_n - 1 - k) % bldpc_code::kEvRing;
        float ms = 0;
        CLDPC_HIP(hipEventSynchronize(c->evr[2 * slot + 1]), BLDPC_EHIP);
        CLDPC_HIP(hipEventElapsedTime(&ms, c->evr[2 * slot], c->evr[2 * slot + 1]), BLDPC_EHIP);
        sum += ms;
    }
    *mean_ms = (float)(sum / n);
    if (launches) *launches = (int)n;
    c->ev_n = 0;
    return BLDPC_OK;
}

extern "C" int bldpc_last_kernel_ms(bldpc_code *c, float *ms)
{
    if (!c || !ms || !c->ev0) return fail(BLDPC_EINVAL, "bldpc_last_kernel_ms: profiling was not enabled");
    CLDPC_HIP(hipEventSynchronize(c->ev1), BLDPC_EHIP);
    CLDPC_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1), BLDPC_EHIP);
    return BLDPC_OK;
}

// ---------------------------------------------------------------------------
template <int VEC>
static int run_table(bldpc_code *c, const float *y, int F, int max_iter, int length, int exit_mode, int *D, float *app,
                     unsigned long long *flag_hist, int *itera, int *iters, hipStream_t st)
{
    const bool early = exit_mode != BLDPC_EXIT_FIXED; // the host watches the flags: stop when every frame is flagged / has stopped
    TableArgs a;
    a.rq = (float *)c->rq.p; a.y = y; a.addr = c->d_addr; a.node_list = c->d_node_list;
    a.wv_blk = c->d_wv_blk; a.wc_blk = c->d_wc_blk;
    a.F = F; a.Z = c->Z; a.Wv = c->Wv; a.Wc = c->Wc; a.length = length;
    a.iters = (exit_mode == BLDPC_EXIT_PER_FRAME) ? iters : nullptr;
    if (a.iters) CLDPC_HIP(hipMemsetAsync(iters, 0, (size_t)F * sizeof(int), st), BLDPC_EHIP);
    int *bad = (int *)c->bad.p, *cnt = (int *)c->cnt.p;
    const dim3 blk(256);
    const unsigned gx = (unsigned)((F + VEC * 256 - 1) / (VEC * 256));
    const bool per_iter_flags = early || flag_hist;
    CLDPC_HIP(hipMemsetAsync(c->rq.p, 0, (size_t)c->M * c->Wc * F * sizeof(float), st), BLDPC_EHIP); // LDPC_Decoder.cu:82
    CLDPC_HIP(hipMemsetAsync(bad, 0, (size_t)F * sizeof(int), st), BLDPC_EHIP);
    if (flag_hist) CLDPC_HIP(hipMemsetAsync(flag_hist, 0, (size_t)F * sizeof(unsigned long long), st), BLDPC_EHIP);
    int it = 0;
    while (it < max_iter) {
        it++;
        const bool last = (it == max_iter);
        const bool want_out = last || early; // D must be current whenever we may stop
        a.D = want_out ? D : nullptr;
        a.app = (want_out && app) ? app : nullptr;
        a.bad = (per_iter_flags || last) ? bad : nullptr;
        for (int lv = 0; lv < c->levels; lv++) { // one launch per collision level (1 for a conflict-free table)
            const int n0 = c->level_begin[lv], cntn = c->level_begin[lv + 1] - n0;
            if (cntn <= 0) continue;
            hipLaunchKernelGGL(k_table_vn<VEC>, dim3(gx, (unsigned)std::min(cntn, 65535)), blk, 0, st, a, n0, cntn);
        }
        if (!last || exit_mode == BLDPC_EXIT_BATCH_GLOBAL) // the CN pass after the final VN pass is unobservable (the reference runs it)
            hipLaunchKernelGGL(k_table_cn<VEC>, dim3(gx, (unsigned)std::min(c->M, 65535)), blk, 0, st, a, c->M);
        if (per_iter_flags || last) {
            const bool need_cnt = early;
            if (need_cnt) CLDPC_HIP(hipMemsetAsync(cnt, 0, sizeof(int), st), BLDPC_EHIP);
            hipLaunchKernelGGL(k_flags, dim3((F + 255) / 256), blk, 0, st, bad, want_out ? D + (size_t)c->N * F : nullptr,
                               flag_hist, need_cnt ? cnt : nullptr, F, it, const_cast<int *>(a.iters), last ? 1 : 0);
            if (need_cnt) {
                CLDPC_HIP(hipMemcpyAsync(c->h_cnt, cnt, sizeof(int), hipMemcpyDeviceToHost, st), BLDPC_EHIP);
                CLDPC_HIP(hipStreamSynchronize(st), BLDPC_EHIP);
                if (*c->h_cnt == F) break; // LDPC_Decoder.cu:150-153
            }
        }
    }
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    *itera = it;
    return BLDPC_OK;
}

static int decode_impl(bldpc_code *c, const float *y, int F, int max_iter, int length, int exit_mode, int kernel, int *D, float *app,
                       unsigned long long *flag_hist, int *itera, int *iters, void *stream, QcStat *stat = nullptr)
{
    if (!c || !y || !D || !itera) return fail(BLDPC_EINVAL, "bldpc_decode: null argument");
    if (F <= 0 || max_iter <= 0) return fail(BLDPC_EINVAL, "bldpc_decode: F=%d max_iter=%d must be positive", F, max_iter);
    if (length == 0) length = c->K;
    if (length < 0 || length > c->N) return fail(BLDPC_EINVAL, "bldpc_decode: length=%d outside [0,%d]", length, c->N);
    if (exit_mode != BLDPC_EXIT_FIXED && exit_mode != BLDPC_EXIT_BATCH_GLOBAL && exit_mode != BLDPC_EXIT_PER_FRAME)
        return fail(BLDPC_EINVAL, "unknown exit_mode %d", exit_mode);
    hipStream_t st = (hipStream_t)stream;
    if (c->profiling) { // this call's event pair
        const unsigned slot = c->ev_n++ % bldpc_code::kEvRing;
        c->ev0 = c->evr[2 * slot];
        c->ev1 = c->evr[2 * slot + 1];
    }
    const bool qc_ok = c->has_qc && c->qc.frames_per_wg > 0;
    // The fused kernels keep a frame's flag history in one 64-bit word: the reference's batch-global rule (which is found
    // from the histories) and a requested flag_hist need max_iter <= 64 there.  The reference takes any maxIT, and so do the
    // table kernels: AUTO goes to them; only an explicit QC_LDS request is refused.
    const bool needs_hist = exit_mode == BLDPC_EXIT_BATCH_GLOBAL || flag_hist != nullptr;
    if (kernel == BLDPC_KERNEL_AUTO) kernel = (qc_ok && !(needs_hist && max_iter > 64)) ? BLDPC_KERNEL_QC_LDS : BLDPC_KERNEL_TABLE;
    if (kernel == BLDPC_KERNEL_QC_LDS) {
        if (!qc_ok)
            return fail(BLDPC_EUNSUPPORTED, "QC_LDS kernel unavailable for this code (%s)",
                        c->has_qc ? "message state exceeds LDS" : "built from an address table");
        if (needs_hist && max_iter > 64)
            return fail(BLDPC_EUNSUPPORTED, "QC_LDS with a flag history (BATCH_GLOBAL exit or flag_hist) supports max_iter <= 64 (got %d); "
                        "BLDPC_KERNEL_AUTO or BLDPC_KERNEL_TABLE take any max_iter", max_iter);
        CLDPC_HIP(c->bad.reserve((size_t)F * sizeof(unsigned long long)), BLDPC_ENOMEM);
        CLDPC_HIP(c->cnt.reserve(64), BLDPC_ENOMEM);
        CLDPC_HIP(c->bits.reserve((size_t)F * (c->N / 32) * sizeof(unsigned)), BLDPC_ENOMEM);
        CLDPC_HIP(c->yg.reserve(((size_t)F + 2) * c->N * sizeof(float)), BLDPC_ENOMEM);
        if (exit_mode == BLDPC_EXIT_BATCH_GLOBAL) CLDPC_HIP(c->itw.reserve((size_t)F * sizeof(int)), BLDPC_ENOMEM);
        const char *used = c->qc.name;
        int r = qc_decode(&c->qc, y, F, max_iter, length, exit_mode, D, app, flag_hist, (unsigned long long *)c->bad.p,
                          (unsigned long long *)c->cnt.p, (unsigned *)c->bits.p, (float *)c->yg.p, itera, iters, (int *)c->itw.p, st,
                          c->profiling ? c->ev0 : nullptr, c->profiling ? c->ev1 : nullptr, stat, &used);
        c->last_kernel = used;
        return r;
    }
    if (kernel != BLDPC_KERNEL_TABLE) return fail(BLDPC_EINVAL, "unknown kernel %d", kernel);
    CLDPC_HIP(c->rq.reserve((size_t)c->M * c->Wc * F * sizeof(float)), BLDPC_ENOMEM);
    CLDPC_HIP(c->bad.reserve((size_t)F * sizeof(unsigned long long)), BLDPC_ENOMEM);
    CLDPC_HIP(c->cnt.reserve(64), BLDPC_ENOMEM);
    const bool a16 = ((uintptr_t)y % 16 == 0) && ((uintptr_t)D % 16 == 0) && (!app || (uintptr_t)app % 16 == 0);
    if (c->profiling) CLDPC_HIP(hipEventRecord(c->ev0, st), BLDPC_EHIP);
    int r;
    if (F % 4 == 0 && a16) {
        c->last_kernel = "table_vec4";
        r = run_table<4>(c, y, F, max_iter, length, exit_mode, D, app, flag_hist, itera, iters, st);
    } else {
        c->last_kernel = "table_vec1";
        r = run_table<1>(c, y, F, max_iter, length, exit_mode, D, app, flag_hist, itera, iters, st);
    }
    if (c->profiling) CLDPC_HIP(hipEventRecord(c->ev1, st), BLDPC_EHIP);
    return r;
}

extern "C" int bldpc_decode(bldpc_code *c, const float *y, int F, int max_iter, int length, int exit_mode, int kernel, int *D,
                            float *app, unsigned long long *flag_hist, int *itera, void *stream)
{
    if (exit_mode == BLDPC_EXIT_PER_FRAME)
        return fail(BLDPC_EINVAL, "bldpc_decode: per-frame exit returns one iteration count per frame, use bldpc_decode_per_frame");
    return decode_impl(c, y, F, max_iter, length, exit_mode, kernel, D, app, flag_hist, itera, nullptr, stream);
}

extern "C" int bldpc_decode_per_frame(bldpc_code *c, const float *y, int F, int max_iter, int length, int kernel, int *D, float *app,
                                      int *iters, void *stream)
{
    if (!iters) return fail(BLDPC_EINVAL, "bldpc_decode_per_frame: null iters");
    int unused = 0;
    return decode_impl(c, y, F, max_iter, length, BLDPC_EXIT_PER_FRAME, kernel, D, app, nullptr, &unused, iters, stream);
}

static int statistic_impl(const bldpc_code *cc, const int *D, const int *cw, int F, int length, int itera, const int *iters,
                          long long *counters, void *stream)
{
    bldpc_code *c = const_cast<bldpc_code *>(cc); // scratch only
    if (!c || !D || !counters || F <= 0) return fail(BLDPC_EINVAL, "bldpc_statistic: bad argument");
    if (length == 0) length = c->K;
    if (length < 0 || length > c->N) return fail(BLDPC_EINVAL, "bldpc_statistic: length=%d outside [0,%d]", length, c->N);
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)F * sizeof(int) > c->errs.cap) {
        CLDPC_HIP(c->errs.reserve((size_t)F * sizeof(int)), BLDPC_ENOMEM);
        CLDPC_HIP(hipMemsetAsync(c->errs.p, 0, (size_t)F * sizeof(int), st), BLDPC_EHIP);
    }
    const int slices = std::max(1, std::min(64, length / 32));
    const int rows = (length + slices - 1) / slices;
    if (length > 0)
        hipLaunchKernelGGL(k_stat_errors, dim3((unsigned)((F + 1023) / 1024), (unsigned)slices), dim3(256), 0, st, D, cw, F, length, rows,
                           (int *)c->errs.p);
    hipLaunchKernelGGL(k_stat_final, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, (int *)c->errs.p, D + (size_t)c->N * F, F, itera,
                       iters, counters);
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    return BLDPC_OK;
}

// bldpc_decode (or bldpc_decode_per_frame when iters != NULL) followed by bldpc_statistic against the all-zero codeword, as
// Simulation_GPU calls them back to back (Simulation.cu:143-145): same D, same counters.  On the fused kernels with a single
// launch (FIXED and PER_FRAME exits) the error counts come out of the pass that unpacks the hard bits into D, and the second
// pass over D's 4 N F bytes is not made.
extern "C" int bldpc_decode_statistic(bldpc_code *c, const float *y, int F, int max_iter, int length, int exit_mode, int kernel, int *D,
                                      int *iters, long long *counters, int *itera, void *stream)
{
    if (!c || !counters || !itera) return fail(BLDPC_EINVAL, "bldpc_decode_statistic: null argument");
    if ((exit_mode == BLDPC_EXIT_PER_FRAME) != (iters != nullptr))
        return fail(BLDPC_EINVAL, "bldpc_decode_statistic: iters goes with the per-frame exit and with nothing else");
    if (F <= 0) return fail(BLDPC_EINVAL, "bldpc_decode_statistic: F=%d must be positive", F);
    const int slen = length == 0 ? c->K : length;
    hipStream_t st = (hipStream_t)stream;
    const bool single = exit_mode == BLDPC_EXIT_FIXED || exit_mode == BLDPC_EXIT_PER_FRAME;
    QcStat stat; // per call: nothing of this lives in the shared code object
    if (single && c->has_qc && slen >= 0 && slen <= c->N) {
        if ((size_t)F * sizeof(int) > c->errs.cap) {
            CLDPC_HIP(c->errs.reserve((size_t)F * sizeof(int)), BLDPC_ENOMEM);
            CLDPC_HIP(hipMemsetAsync(c->errs.p, 0, (size_t)F * sizeof(int), st), BLDPC_EHIP);
        }
        stat.errs = (int *)c->errs.p;
        stat.length = slen;
    }
    const int r = decode_impl(c, y, F, max_iter, length, exit_mode, kernel, D, nullptr, nullptr, itera, iters, stream, &stat);
    const bool fused = stat.done;
    if (r) {
        // the unpack pass may have accumulated into errs before the failure: k_stat_final, which re-zeroes it, will not run
        if (fused) (void)hipMemsetAsync(c->errs.p, 0, (size_t)F * sizeof(int), st);
        return r;
    }
    if (!fused) return statistic_impl(c, D, nullptr, F, length, *itera, iters, counters, stream);
    hipLaunchKernelGGL(k_stat_final, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, (int *)c->errs.p, D + (size_t)c->N * F, F, *itera,
                       iters, counters);
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    return BLDPC_OK;
}

extern "C" int bldpc_statistic(const bldpc_code *cc, const int *D, const int *cw, int F, int length, int itera, long long *counters,
                               void *stream)
{
    return statistic_impl(cc, D, cw, F, length, itera, nullptr, counters, stream);
}

extern "C" int bldpc_statistic_per_frame(const bldpc_code *cc, const int *D, const int *cw, int F, int length, const int *iters,
                                         long long *counters, void *stream)
{
    if (!iters) return fail(BLDPC_EINVAL, "bldpc_statistic_per_frame: null iters");
    return statistic_impl(cc, D, cw, F, length, 0, iters, counters, stream);
}
