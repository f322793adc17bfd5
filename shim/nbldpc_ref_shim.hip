// nbldpc_ref_shim.hip -- Decoding_EMS / Decoding_TMM / Decoding_layered_TMM with the reference's signatures
// (myNBLDPC/include/LDPC_Decoder.h:13,23,25) as thin wrappers over include/nbldpc.h.  Kept: return value, iter_number,
// DecodeOutput, the state left in VN[].LLR and CN[].L_c2v (q-1 entries for EMS, q for the trellis decoders), one frame
// per call, callable from several host threads (the code object is shared and read-only, buffers are per thread).
// Not kept: the per-call mallocs of the GPU twin (Decode_GPU.cu:144-167); VN[].sort_L_v2c / sort_Entr_v2c, the scratch of the
// CPU decoder's sort (LDPC_Decoder.cpp:247-266) -- no caller of the reference reads them after the call (decode_once_cpu / _gpu
// read DecodeOutput and iter_number, Simulation.cpp:56-83,130-160) and the reference's own GPU twin leaves them at their
// initial value, the channel vector (Decode_GPU.cu:170-177): here they are left untouched.
#include "nbldpc_ref_shim.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../include/nbldpc.h"

namespace {
struct Cfg { int q = 0, dv = 0, dc = 0, maxIT = 0; std::vector<unsigned> mul; } g_cfg;
nbldpc_code *g_code = nullptr;
std::mutex g_mtx;

void die(const char *what)
{
    printf("%s: %s, exit!\n", what, nbldpc_last_error());
    exit(0);
}

void build_code(const LDPCCode *H, const VN *V, const CN *C)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_code) return;
    const Cfg &c = g_cfg;
    if (!c.q) { printf("nbldpc_shim_configure was not called, exit!\n"); exit(0); }
    const int N = H->Variablenode_num, M = H->Checknode_num;
    std::vector<int> vw(N), vc((size_t)N * c.dv, -1), vg((size_t)N * c.dv, 0), cw(M), cv((size_t)M * c.dc, -1), cg((size_t)M * c.dc, 0);
    for (int i = 0; i < N; i++) {
        vw[i] = V[i].weight;
        for (int d = 0; d < V[i].weight; d++) { vc[(size_t)i * c.dv + d] = V[i].linkCNs[d]; vg[(size_t)i * c.dv + d] = V[i].linkCNs_GF[d]; }
    }
    for (int r = 0; r < M; r++) {
        cw[r] = C[r].weight;
        for (int d = 0; d < C[r].weight; d++) { cv[(size_t)r * c.dc + d] = C[r].linkVNs[d]; cg[(size_t)r * c.dc + d] = C[r].linkVNs_GF[d]; }
    }
    if (nbldpc_code_create(N, M, c.q, c.dv, c.dc, vw.data(), vc.data(), vg.data(), cw.data(), cv.data(), cg.data(), c.mul.data(), &g_code))
        die("nbldpc_code_create");
}

struct Buffers {
    hipStream_t st = nullptr;
    float *Lch = nullptr, *LLR = nullptr, *c2v = nullptr;
    int *out = nullptr, *it = nullptr, *ok = nullptr;
    std::vector<float> h_Lch, h_LLR, h_c2v;
    void ensure(int N, int M, int q, int dc)
    {
        if (st) return;
        bool good = hipStreamCreate(&st) == hipSuccess;
        good = good && hipMalloc((void **)&Lch, (size_t)N * (q - 1) * 4) == hipSuccess && hipMalloc((void **)&LLR, (size_t)N * q * 4) == hipSuccess;
        good = good && hipMalloc((void **)&c2v, (size_t)M * dc * q * 4) == hipSuccess && hipMalloc((void **)&out, (size_t)N * 4) == hipSuccess;
        good = good && hipMalloc((void **)&it, 4) == hipSuccess && hipMalloc((void **)&ok, 4) == hipSuccess;
        if (!good) { printf("Cannot malloc the decoder buffers on device, exit!\n"); exit(0); }
        h_Lch.resize((size_t)N * (q - 1)); h_LLR.resize((size_t)N * q); h_c2v.resize((size_t)M * dc * q);
    }
};

// method: 0 EMS, 1 TMM, 3 layered TMM (define.h:37)
int decode_one(int method, const LDPCCode *H, VN *V, CN *C, int Nm, int Nc, int *DecodeOutput, int &iter_number)
{
    build_code(H, V, C); // takes the mutex and returns at once when the code exists: THREAD_NUM host threads call in here (Simulation.cpp:175-185)
    const Cfg &c = g_cfg;
    const int N = H->Variablenode_num, M = H->Checknode_num, q = c.q, nv = (method == 0) ? q - 1 : q;
    thread_local Buffers b;
    b.ensure(N, M, q, c.dc);
    for (int i = 0; i < N; i++) memcpy(&b.h_Lch[(size_t)i * (q - 1)], V[i].L_ch, (size_t)(q - 1) * sizeof(float));
    (void)hipMemcpyAsync(b.Lch, b.h_Lch.data(), b.h_Lch.size() * 4, hipMemcpyHostToDevice, b.st);
    int rc;
    if (method == 0) rc = nbldpc_ems_decode_batch(g_code, b.Lch, 1, Nm, Nc, c.maxIT, c.dc, b.out, b.it, b.ok, b.LLR, b.c2v, b.st);
    else rc = nbldpc_tmm_decode_batch(g_code, b.Lch, 1, method == 3, c.maxIT, b.out, b.it, b.ok, b.LLR, b.c2v, b.st);
    if (rc) die(method == 0 ? "nbldpc_ems_decode_batch" : "nbldpc_tmm_decode_batch");
    int ok = 0;
    (void)hipMemcpyAsync(DecodeOutput, b.out, (size_t)N * 4, hipMemcpyDeviceToHost, b.st);
    (void)hipMemcpyAsync(&iter_number, b.it, 4, hipMemcpyDeviceToHost, b.st);
    (void)hipMemcpyAsync(&ok, b.ok, 4, hipMemcpyDeviceToHost, b.st);
    (void)hipMemcpyAsync(b.h_LLR.data(), b.LLR, (size_t)N * nv * 4, hipMemcpyDeviceToHost, b.st);
    (void)hipMemcpyAsync(b.h_c2v.data(), b.c2v, (size_t)M * c.dc * nv * 4, hipMemcpyDeviceToHost, b.st);
    if (hipStreamSynchronize(b.st) != hipSuccess) { printf("decode failed on the device, exit!\n"); exit(0); }
    for (int i = 0; i < N; i++) memcpy(V[i].LLR, &b.h_LLR[(size_t)i * nv], (size_t)nv * sizeof(float));
    for (int r = 0; r < M; r++)
        for (int d = 0; d < C[r].weight; d++) memcpy(C[r].L_c2v[d], &b.h_c2v[((size_t)r * c.dc + d) * nv], (size_t)nv * sizeof(float));
    return ok; // 1 = zero syndrome reached, iter_number already decremented (LDPC_Decoder.cpp:232-238)
}
} // namespace

extern "C" int nbldpc_shim_configure(int GFQ, int maxdv, int maxdc, int maxIT, const unsigned *TableMultiply)
{
    if (GFQ < 4 || maxdv <= 0 || maxdc <= 0 || maxIT <= 0 || !TableMultiply) return NBLDPC_EINVAL;
    nbldpc_shim_reset();
    g_cfg.q = GFQ; g_cfg.dv = maxdv; g_cfg.dc = maxdc; g_cfg.maxIT = maxIT;
    g_cfg.mul.assign(TableMultiply, TableMultiply + (size_t)GFQ * GFQ);
    return NBLDPC_OK;
}

extern "C" void nbldpc_shim_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_code) nbldpc_code_destroy(g_code);
    g_code = nullptr;
}

int Decoding_EMS(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number)
{
    return decode_one(0, H, Variablenode, Checknode, EMS_Nm, EMS_Nc, DecodeOutput, iter_number);
}
int Decoding_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number)
{
    return decode_one(1, H, Variablenode, Checknode, EMS_Nm, EMS_Nc, DecodeOutput, iter_number);
}
int Decoding_layered_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number)
{
    return decode_one(3, H, Variablenode, Checknode, EMS_Nm, EMS_Nc, DecodeOutput, iter_number);
}
