// ldpc_ref_shim.hip -- the binary program's entry points with the reference's signatures (bldpc_实习/LDPC_Decoder.cuh:5,
// Simulation.cuh:4-12) as thin wrappers over include/bldpc.h.  Semantics kept: batch-global early exit
// (LDPC_Decoder.cu:150-153), host D with the flag row, LDPC->iteraTime, the counters and printed rows of Statistic,
// printf + exit(0) on failure.  Not kept: per-call cudaMalloc/cudaFree of the scratch, the dead BPSK kernel path.
#include "ldpc_ref_shim.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/bldpc.h"

namespace {
struct Cfg {
    int J = 0, L = 0, Z = 0, F = 0, length = 0, maxIT = 0;
    std::string path = "PON_LDPC.txt"; // Simulation.cu:296
    int as_written = 1;
    long leastErrorFrames = 50, leastTestFrames = 10000, displayStep = 40960; // define.cuh:52-54
    int device_channel = 0, device_statistics = 0, exit_mode = BLDPC_EXIT_BATCH_GLOBAL; // bldpc_shim_configure_fast
    long max_batches = 0;
} g_cfg;
bldpc_code *g_code = nullptr;
int *g_D_dev = nullptr;
size_t g_D_cap = 0;

void die(const char *what)
{
    printf("%s: %s, exit!\n", what, bldpc_last_error());
    exit(0);
}

// If `addr` is the intended circulant expansion (row = (c - s) mod Z) of some shift matrix, return that
// matrix in H: block (j, l) is present when column l*Z holds a slot of block row j, its shift is read from that slot,
// and the whole table is then rebuilt from H and compared.
bool table_is_qc(const Cfg &c, const int *wc, const int *wv, const std::vector<int> &addr, std::vector<int> &H)
{
    const int Wc = wc[c.J], Wv = wv[c.L];
    H.assign((size_t)c.J * c.L, -1);
    for (int l = 0; l < c.L; l++)
        for (int k = 0; k < wv[l]; k++) {
            const int slot = addr[((size_t)l * c.Z) * Wv + k];
            if (slot < 0 || slot >= c.J * c.Z * Wc) return false;
            const int row = slot / Wc, j = row / c.Z, r = row % c.Z;
            H[(size_t)j * c.L + l] = (c.Z - r) % c.Z; // row = (0 - s) mod Z at c = 0
        }
    std::vector<int> wc2(c.J + 1, 0), wv2(c.L + 1, 0), chk(addr.size());
    for (int j = 0; j < c.J; j++) {
        for (int l = 0; l < c.L; l++) wc2[j] += H[(size_t)j * c.L + l] != -1;
        if (wc2[j] != wc[j]) return false;
    }
    for (int l = 0; l < c.L; l++) {
        for (int j = 0; j < c.J; j++) wv2[l] += H[(size_t)j * c.L + l] != -1;
        if (wv2[l] != wv[l]) return false;
    }
    wc2[c.J] = Wc; wv2[c.L] = Wv;
    if (bldpc_transform_h(H.data(), c.J, c.L, c.Z, wc2.data(), wv2.data(), chk.data(), 0)) return false;
    for (int n = 0; n < c.L * c.Z; n++) // entries beyond a column's weight are unspecified in the reference's table
        for (int k = 0; k < wv[n / c.Z]; k++)
            if (chk[(size_t)n * Wv + k] != addr[(size_t)n * Wv + k]) return false;
    return true;
}
} // namespace

extern "C" int bldpc_shim_configure(int J, int L, int Z, int frames, int length, int maxIT)
{
    if (J <= 0 || L <= J || Z <= 0 || frames <= 0 || maxIT <= 0 || length < 0 || length > L * Z) return BLDPC_EINVAL;
    bldpc_shim_reset();
    g_cfg.J = J; g_cfg.L = L; g_cfg.Z = Z; g_cfg.F = frames; g_cfg.length = length; g_cfg.maxIT = maxIT;
    return BLDPC_OK;
}

extern "C" int bldpc_shim_configure_sim(const char *path, int as_written, long leastErrorFrames, long leastTestFrames, long displayStep)
{
    if (leastErrorFrames < 0 || leastTestFrames < 0 || displayStep <= 0) return BLDPC_EINVAL;
    if (path) g_cfg.path = path;
    g_cfg.as_written = as_written ? 1 : 0;
    g_cfg.leastErrorFrames = leastErrorFrames; g_cfg.leastTestFrames = leastTestFrames; g_cfg.displayStep = displayStep;
    return BLDPC_OK;
}

extern "C" int bldpc_shim_configure_fast(int device_channel, int device_statistics, int exit_mode, long max_batches)
{
    if (exit_mode != BLDPC_EXIT_FIXED && exit_mode != BLDPC_EXIT_BATCH_GLOBAL && exit_mode != BLDPC_EXIT_PER_FRAME) return BLDPC_EINVAL;
    if (exit_mode == BLDPC_EXIT_PER_FRAME && !device_statistics) return BLDPC_EINVAL; // the host Statistic takes ONE iteraTime per batch
    if (max_batches < 0) return BLDPC_EINVAL;
    g_cfg.device_channel = device_channel ? 1 : 0; g_cfg.device_statistics = device_statistics ? 1 : 0;
    g_cfg.exit_mode = exit_mode; g_cfg.max_batches = max_batches;
    return BLDPC_OK;
}

extern "C" void bldpc_shim_reset(void)
{
    if (g_code) bldpc_code_destroy(g_code);
    g_code = nullptr;
}

extern "C" const char *bldpc_shim_last_kernel(void) { return g_code ? bldpc_last_kernel(g_code) : "none"; }

void Get_H(int *H, int *Weight_Checknode, int *Weight_Variablenode)
{
    const Cfg &c = g_cfg;
    if (!c.J) { printf("bldpc_shim_configure was not called, exit!\n"); exit(0); }
    if (bldpc_read_blockh(c.path.c_str(), c.J, c.L, H, Weight_Checknode, Weight_Variablenode)) die("Get_H");
}

void Transform_H(int *H, int *Weight_Checknode, int *Weight_Variablenode, int *Address_Variablenode)
{
    const Cfg &c = g_cfg;
    if (!c.J) { printf("bldpc_shim_configure was not called, exit!\n"); exit(0); }
    if (bldpc_transform_h(H, c.J, c.L, c.Z, Weight_Checknode, Weight_Variablenode, Address_Variablenode, c.as_written)) die("Transform_H");
}

static void ensure_code(int *Address_Variablenode, int *Weight_Checknode, int *Weight_Variablenode)
{
    const Cfg &c = g_cfg;
    const int N = c.L * c.Z;
    if (g_code) return;
    // the reference keeps the table on the device (main.cu:73,98); the builder wants it on the host once
    std::vector<int> addr((size_t)N * Weight_Variablenode[c.L]), H;
    if (hipMemcpy(addr.data(), Address_Variablenode, addr.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
        printf("Cannot copy Address_Variablenode to the host in LDPC_Decoder_GPU, exit!\n");
        exit(0);
    }
    if (table_is_qc(c, Weight_Checknode, Weight_Variablenode, addr, H)) {
        if (bldpc_code_create_qc(c.J, c.L, c.Z, H.data(), &g_code)) die("bldpc_code_create_qc");
    } else if (bldpc_code_create_table(c.J, c.L, c.Z, Weight_Checknode, Weight_Variablenode, addr.data(), &g_code)) {
        die("bldpc_code_create_table");
    }
}

static void ensure_D(size_t d_bytes)
{
    if (d_bytes <= g_D_cap) return;
    if (g_D_dev) (void)hipFree(g_D_dev);
    if (hipMalloc((void **)&g_D_dev, d_bytes) != hipSuccess) { printf("Cannot malloc D_GPU in LDPC_Decoder_GPU on device, exit!\n"); exit(0); }
    g_D_cap = d_bytes;
}

void LDPC_Decoder_GPU(int *D, float *Channel_Out, cudaDeviceProp prop, int *Address_Variablenode, int *Weight_Checknode,
                      int *Weight_Variablenode, LDPCCode *LDPC)
{
    (void)prop; // the reference reads only prop.maxThreadsPerBlock, to size its own launches (LDPC_Decoder.cu:97-124)
    const Cfg &c = g_cfg;
    if (!c.J) { printf("bldpc_shim_configure was not called, exit!\n"); exit(0); }
    const int N = c.L * c.Z;
    ensure_code(Address_Variablenode, Weight_Checknode, Weight_Variablenode);
    const size_t d_bytes = (size_t)(N + 1) * c.F * sizeof(int);
    ensure_D(d_bytes);
    if (bldpc_decode(g_code, Channel_Out, c.F, c.maxIT, c.length, c.exit_mode == BLDPC_EXIT_FIXED ? BLDPC_EXIT_FIXED : BLDPC_EXIT_BATCH_GLOBAL, BLDPC_KERNEL_AUTO, g_D_dev, nullptr, nullptr,
                     &LDPC->iteraTime, nullptr))
        die("bldpc_decode");
    if (hipMemcpy(D, g_D_dev, d_bytes, hipMemcpyDeviceToHost) != hipSuccess) { printf("Cannot copy D_GPU to D, exit!\n"); exit(0); }
}

// The tail of Statistic (Simulation.cu:264-284): the result row every displayStep frames, again when the stop rule is met (the
// reference prints it twice when both fall on the same batch), 1 when the stop rule is met.  `last`: a max_batches stop.
static bool g_last_batch = false; // Simulation_GPU -> Statistic: this batch ends the point by max_batches
static int rows_and_stop(Simulation *SIM, bool last)
{
    const Cfg &c = g_cfg;
    const bool stop = SIM->num_Error_Frames >= c.leastErrorFrames && SIM->num_Frames >= c.leastTestFrames;
    auto row = [&]() {
        SIM->BER = (float)(((double)SIM->num_Error_Bits / (double)SIM->num_Frames) / (double)c.length);
        SIM->FER = (float)((double)SIM->num_Error_Frames / (double)SIM->num_Frames);
        SIM->AverageIT = (float)((double)SIM->Total_Iteration / (double)SIM->num_Frames);
        SIM->FER_Alarm = (float)((double)SIM->num_Alarm_Frames / (double)SIM->num_Frames);
        SIM->FER_False = (float)((double)SIM->num_False_Frames / (double)SIM->num_Frames);
        printf(" %.1f %8ld  %4ld  %6.4e  %6.4e  %.2f  %6.4e %6.4e\n", SIM->SNR, SIM->num_Frames, SIM->num_Error_Frames, SIM->FER, SIM->BER,
               SIM->AverageIT, SIM->FER_False, SIM->FER_Alarm);
    };
    if (SIM->num_Frames % c.displayStep == 0) row();
    if (stop || (last && SIM->num_Frames % c.displayStep != 0)) row();
    return stop ? 1 : 0;
}

int Statistic(Simulation *SIM, int *CodeWord_Frames, int *D, LDPCCode *LDPC)
{
    const Cfg &c = g_cfg;
    const int F = c.F, N = c.L * c.Z, Length = c.length;
    std::vector<int> err(F, 0); // message-bit errors per frame
    for (int n = 0; n < Length; n++) {
        const int *d = D + (size_t)n * F, *w = CodeWord_Frames + (size_t)n * F;
        for (int f = 0; f < F; f++) err[f] += d[f] != w[f];
    }
    const int *flag = D + (size_t)N * F; // row N: 1 = the frame passed the all-zero test (LDPC_Decoder.cu:134-147)
    for (int f = 0; f < F; f++) {
        SIM->num_Error_Bits += err[f];
        SIM->num_Error_Frames += (err[f] != 0 || flag[f] == 0);
        SIM->num_Alarm_Frames += (err[f] == 0 && flag[f] == 0);
        SIM->num_False_Frames += (err[f] != 0 && flag[f] == 1);
        SIM->Total_Iteration += LDPC->iteraTime; // batch-global count, once per frame (Simulation.cu:262)
    }
    return rows_and_stop(SIM, g_last_batch);
}

void Simulation_GPU(AWGNChannel *AWGN, float *sigma_GPU, Simulation *SIM, int *Address_Variablenode, int *Weight_Checknode,
                    int *Weight_Variablenode)
{
    (void)sigma_GPU;
    const Cfg &c = g_cfg;
    if (!c.J) { printf("bldpc_shim_configure was not called, exit!\n"); exit(0); }
    const size_t N = (size_t)c.L * c.Z, F = (size_t)c.F;
    cudaDeviceProp prop;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { printf("There is no GPU beyond 1.0, exit!\n"); exit(0); }
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("Cannot get device properties, exit!\n"); exit(0); }
    // host copies only where the host needs them: PN_Message 0, the all-zero codeword (Simulation.cu:96-106)
    std::vector<int> CodeWord(c.device_statistics ? 0 : N * F, 0), D(c.device_statistics ? 0 : (N + 1) * F);
    std::vector<float> Channel_Out(c.device_channel ? 0 : N * F);
    float *Channel_Out_GPU = nullptr;
    if (hipMalloc((void **)&Channel_Out_GPU, N * F * sizeof(float)) != hipSuccess) {
        printf("Cannot malloc Channel_Out_GPU in SNR_Simulation_GPU on device, exit!\n");
        exit(0);
    }
    LDPCCode LDPC;
    const bool fast_stats = c.device_statistics != 0;
    long long *cnt_dev = nullptr, *cnt_host = nullptr; // the five counters of one batch
    int *iters_dev = nullptr;
    if (fast_stats) {
        ensure_code(Address_Variablenode, Weight_Checknode, Weight_Variablenode);
        ensure_D((N + 1) * F * sizeof(int));
        bool good = hipMalloc((void **)&cnt_dev, 5 * sizeof(long long)) == hipSuccess && hipHostMalloc((void **)&cnt_host, 5 * sizeof(long long), hipHostMallocDefault) == hipSuccess;
        if (c.exit_mode == BLDPC_EXIT_PER_FRAME) good = good && hipMalloc((void **)&iters_dev, F * sizeof(int)) == hipSuccess;
        if (!good) { printf("Cannot malloc the counters on device, exit!\n"); exit(0); }
    }
    for (long batches = 1;; batches++) {
        SIM->num_Frames += c.F; // Simulation.cu:113
        if (c.device_channel) {
            if (bldpc_awgn_channel_device(AWGN->seed, AWGN->sigma, Channel_Out_GPU, nullptr, (int)N, c.F, nullptr)) die("AWGNChannel (device)");
        } else {
            if (bldpc_awgn_channel_host(AWGN->seed, AWGN->sigma, Channel_Out.data(), nullptr, (int)N, c.F)) die("AWGNChannel_CPU"); // NULL: all-zero word
            if (hipMemcpy(Channel_Out_GPU, Channel_Out.data(), N * F * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                printf("Cannot copy Channel_Out to the device, exit!\n");
                exit(0);
            }
        }
        const bool last = c.max_batches > 0 && batches >= c.max_batches;
        int stop;
        if (fast_stats) { // LDPC_Decoder_GPU + Statistic (Simulation.cu:143-145) as one device-side call; 40 bytes come back
            (void)hipMemsetAsync(cnt_dev, 0, 5 * sizeof(long long), nullptr);
            if (bldpc_decode_statistic(g_code, Channel_Out_GPU, c.F, c.maxIT, c.length, c.exit_mode, BLDPC_KERNEL_AUTO, g_D_dev, iters_dev, cnt_dev,
                                       &LDPC.iteraTime, nullptr))
                die("bldpc_decode_statistic");
            if (hipMemcpyAsync(cnt_host, cnt_dev, 5 * sizeof(long long), hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
                printf("Cannot copy the counters to the host, exit!\n");
                exit(0);
            }
            SIM->num_Error_Frames += cnt_host[0]; SIM->num_Error_Bits += cnt_host[1]; SIM->Total_Iteration += cnt_host[2];
            SIM->num_False_Frames += cnt_host[3]; SIM->num_Alarm_Frames += cnt_host[4];
            stop = rows_and_stop(SIM, last);
        } else {
            LDPC_Decoder_GPU(D.data(), Channel_Out_GPU, prop, Address_Variablenode, Weight_Checknode, Weight_Variablenode, &LDPC);
            g_last_batch = last;
            stop = Statistic(SIM, CodeWord.data(), D.data(), &LDPC);
            g_last_batch = false;
        }
        if (stop == 1 || last) break;
    }
    if (cnt_dev) (void)hipFree(cnt_dev);
    if (cnt_host) (void)hipHostFree(cnt_host);
    if (iters_dev) (void)hipFree(iters_dev);
    (void)hipFree(Channel_Out_GPU);
}
