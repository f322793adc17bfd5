// ldpc_ref_shim.hip -- LDPC_Decoder_GPU with the reference's signature (bldpc_实习/LDPC_Decoder.cuh:5) as a thin
// wrapper over include/bldpc.h.  Semantics kept: batch-global early exit (LDPC_Decoder.cu:150-153), host D with the
// flag row, LDPC->iteraTime, printf + exit(0) on failure.  Not kept: per-call cudaMalloc/cudaFree of the scratch.
#include "ldpc_ref_shim.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/bldpc.h"

namespace {
struct Cfg { int J = 0, L = 0, Z = 0, F = 0, length = 0, maxIT = 0; } g_cfg;
bldpc_code *g_code = nullptr;
int *g_D_dev = nullptr;
size_t g_D_cap = 0;

void die(const char *what)
{
    printf("%s: %s, exit!\n", what, bldpc_last_error());
    exit(0);
}
} // namespace

extern "C" int bldpc_shim_configure(int J, int L, int Z, int frames, int length, int maxIT)
{
    if (J <= 0 || L <= J || Z <= 0 || frames <= 0 || maxIT <= 0 || length < 0 || length > L * Z) return BLDPC_EINVAL;
    bldpc_shim_reset();
    g_cfg.J = J; g_cfg.L = L; g_cfg.Z = Z; g_cfg.F = frames; g_cfg.length = length; g_cfg.maxIT = maxIT;
    return BLDPC_OK;
}

extern "C" void bldpc_shim_reset(void)
{
    if (g_code) bldpc_code_destroy(g_code);
    g_code = nullptr;
}

void LDPC_Decoder_GPU(int *D, float *Channel_Out, hipDeviceProp_t prop, int *Address_Variablenode, int *Weight_Checknode,
                      int *Weight_Variablenode, LDPCCode *LDPC)
{
    (void)prop; // the reference reads only prop.maxThreadsPerBlock, to size its own launches (LDPC_Decoder.cu:97-124)
    const Cfg &c = g_cfg;
    if (!c.J) { printf("bldpc_shim_configure was not called, exit!\n"); exit(0); }
    const int N = c.L * c.Z;
    if (!g_code) {
        // the reference keeps the table on the device (main.cu:73,98); the builder wants it on the host once
        std::vector<int> addr((size_t)N * Weight_Variablenode[c.L]);
        if (hipMemcpy(addr.data(), Address_Variablenode, addr.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
            printf("Cannot copy Address_Variablenode to the host in LDPC_Decoder_GPU, exit!\n");
            exit(0);
        }
        if (bldpc_code_create_table(c.J, c.L, c.Z, Weight_Checknode, Weight_Variablenode, addr.data(), &g_code)) die("bldpc_code_create_table");
    }
    const size_t d_bytes = (size_t)(N + 1) * c.F * sizeof(int);
    if (d_bytes > g_D_cap) {
        if (g_D_dev) (void)hipFree(g_D_dev);
        if (hipMalloc((void **)&g_D_dev, d_bytes) != hipSuccess) { printf("Cannot malloc D_GPU in LDPC_Decoder_GPU on device, exit!\n"); exit(0); }
        g_D_cap = d_bytes;
    }
    if (bldpc_decode(g_code, Channel_Out, c.F, c.maxIT, c.length, BLDPC_EXIT_BATCH_GLOBAL, BLDPC_KERNEL_AUTO, g_D_dev, nullptr, nullptr,
                     &LDPC->iteraTime, nullptr))
        die("bldpc_decode");
    if (hipMemcpy(D, g_D_dev, d_bytes, hipMemcpyDeviceToHost) != hipSuccess) { printf("Cannot copy D_GPU to D, exit!\n"); exit(0); }
}
