// nbldpc_ref_shim.hip -- the non-binary program's entry points with the reference's signatures (myNBLDPC/include/Simulation.h:8-20,
// Decode_GPU.cuh:17,19, LDPC_Decoder.h:9-25, LDPC_Encoder.h:7-15, GF.h:7-19) as thin wrappers over include/nbldpc.h.
//
// Kept: return values, iter_number, DecodeOutput, the counters / printed rows / stop rule of the frame loop
// (Simulation.cpp:41-83,115-158,191-207,256-311), AWGN->seed as the reference leaves it, printf + exit(0) on failure; for the
// one-frame decoder entry points also the state left in VN[].LLR and CN[].L_c2v (q-1 entries for EMS, q for the trellis decoders).
// Changed on purpose: Simulation_CPU / Simulation_GPU decode cfg.batch frames per launch instead of one (same counters: the
// results are accounted frame by frame in stream order and the accounting stops where the reference's loop would have);
// the THREAD_NUM host threads are gone (the batch replaces them); no per-call cudaMalloc (Decode_GPU.cu:144-167).
// Not kept: VN[].sort_L_v2c / sort_Entr_v2c (scratch of the CPU decoder's sort, LDPC_Decoder.cpp:247-266; no caller reads them and
// the reference's own GPU twin leaves them at their initial value, Decode_GPU.cu:170-177); after Simulation_* the node arrays do
// not hold the last frame's LLR / L_c2v (no caller reads them: decode_once_* use DecodeOutput and iter_number only).
#include "nbldpc_ref_shim.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/nbldpc.h"

unsigned **TableAdd = nullptr; // GF.cpp:18-20
unsigned **TableMultiply = nullptr;
unsigned *TableInverse = nullptr;

namespace {
struct Cfg { int q = 0, dv = 0, dc = 0, maxIT = 0; std::vector<unsigned> mul; } g_cfg; // what the decoders need
struct SimCfg {                                                                       // define.h:23-59
    bool set = false;
    std::string Matrixfile = "BDS.576.288.GF.64.txt", Constellationfile = "./Constellation/BPSK.txt", GFTabledir = "./GF", results_file;
    int n_QAM = 2, THREAD_NUM = 1, EMS_NM = 2, EMS_NC = 2, decoder_method = 0;
    long leastErrorFrames = 50, leastTestFrames = 1000, displayStep = 100000;
    int batch = 4096, device_channel = 0;
} g_sim;
nbldpc_code *g_code = nullptr;
std::mutex g_mtx;
nbldpc_shim_frame_hook g_hook = nullptr;
void *g_hook_user = nullptr;

[[noreturn]] void die(const char *what)
{
    printf("%s: %s, exit!\n", what, nbldpc_last_error());
    exit(0);
}
[[noreturn]] void die_msg(const char *msg)
{
    printf("%s\n", msg);
    exit(0);
}
#define HIP_OR_DIE(call, msg)                      \
    do {                                           \
        if ((call) != hipSuccess) die_msg(msg);    \
    } while (0)

int thread_num() { return g_sim.THREAD_NUM > 0 ? g_sim.THREAD_NUM : 1; }

void build_code(const LDPCCode *H, const VN *V, const CN *C)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_code) return;
    const Cfg &c = g_cfg;
    if (!c.q) die_msg("nbldpc_shim_configure / nbldpc_shim_configure_sim was not called, exit!");
    const unsigned *mul = c.mul.empty() ? (TableMultiply ? TableMultiply[0] : nullptr) : c.mul.data(); // GFInitial's table (contiguous, malloc_2)
    if (!mul) die_msg("GFInitial was not called, exit!");
    const int N = H->Variablenode_num, M = H->Checknode_num;
    std::vector<int> vw(N), vc((size_t)N * c.dv, -1), vg((size_t)N * c.dv, 0), cw(M), cv((size_t)M * c.dc, -1), cg((size_t)M * c.dc, 0);
    for (int i = 0; i < N; i++) {
        vw[i] = V[i].weight;
        if (V[i].weight > c.dv) die_msg("a variable node is heavier than maxdv, exit!");
        for (int d = 0; d < V[i].weight; d++) { vc[(size_t)i * c.dv + d] = V[i].linkCNs[d]; vg[(size_t)i * c.dv + d] = V[i].linkCNs_GF[d]; }
    }
    for (int r = 0; r < M; r++) {
        cw[r] = C[r].weight;
        if (C[r].weight > c.dc) die_msg("a check node is heavier than maxdc, exit!");
        for (int d = 0; d < C[r].weight; d++) { cv[(size_t)r * c.dc + d] = C[r].linkVNs[d]; cg[(size_t)r * c.dc + d] = C[r].linkVNs_GF[d]; }
    }
    if (nbldpc_code_create(N, M, c.q, c.dv, c.dc, vw.data(), vc.data(), vg.data(), cw.data(), cv.data(), cg.data(), mul, &g_code))
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
        if (!good) die_msg("Cannot malloc the decoder buffers on device, exit!");
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
    if (hipStreamSynchronize(b.st) != hipSuccess) die_msg("decode failed on the device, exit!");
    for (int i = 0; i < N; i++) memcpy(V[i].LLR, &b.h_LLR[(size_t)i * nv], (size_t)nv * sizeof(float));
    for (int r = 0; r < M; r++)
        for (int d = 0; d < C[r].weight; d++) memcpy(C[r].L_c2v[d], &b.h_c2v[((size_t)r * c.dc + d) * nv], (size_t)nv * sizeof(float));
    return ok; // 1 = zero syndrome reached, iter_number already decremented (LDPC_Decoder.cpp:232-238)
}

// ---- the result row and the per-frame accounting of Statistic (Simulation.cpp:256-311) -----------------------------------------
void ratios(Simulation *SIM, const LDPCCode *H)
{
    SIM->BER = ((double)SIM->num_Error_Bits / (double)(SIM->num_Frames)) / (double)(H->Variablenode_num); // symbol errors / N (sic, :276)
    SIM->FER = (double)SIM->num_Error_Frames / (double)SIM->num_Frames;
    SIM->AverageIT = (double)SIM->Total_Iteration / (double)SIM->num_Frames;
}
void row(const Simulation *SIM, bool to_file)
{
    char line[256];
    snprintf(line, sizeof(line), " %.1f %8ld  %4ld  %6.4e  %6.4e  %.2f  %6.4esec\n", SIM->SNR, SIM->num_Frames, SIM->num_Error_Frames, SIM->FER,
             SIM->BER, SIM->AverageIT, SIM->sumTime / SIM->num_Frames / thread_num());
    fputs(line, stdout);
    if (to_file && !g_sim.results_file.empty()) { // Simulation.cpp:199-206: appended as well
        FILE *fp = fopen(g_sim.results_file.c_str(), "a");
        if (!fp) { printf("can not open file: %s\n", g_sim.results_file.c_str()); exit(0); }
        fputs(line, fp);
        fclose(fp);
    }
}
// One decoded frame with Error_msgBit wrong symbols; the caller has already advanced num_Frames and Total_Iteration (:149-153).
int account(Simulation *SIM, int Error_msgBit, const LDPCCode *H)
{
    SIM->num_Error_Bits += Error_msgBit;
    SIM->num_Error_Frames += (Error_msgBit != 0);
    if (SIM->num_Frames % g_sim.displayStep == 0) { ratios(SIM, H); row(SIM, true); }
    if (SIM->num_Error_Frames >= g_sim.leastErrorFrames && SIM->num_Frames >= g_sim.leastTestFrames) { ratios(SIM, H); return 1; }
    return 0;
}

// ---- batched frame loop behind Simulation_CPU / Simulation_GPU ---------------------------------------------------------------------
struct SimBuffers {
    hipStream_t st = nullptr;
    int cap = 0, len = 0, N = 0, q = 0;
    float *d_tx = nullptr, *d_rx = nullptr, *d_Lch = nullptr, *d_con = nullptr, *h_rx = nullptr;
    int *d_out = nullptr, *d_it = nullptr, *d_ok = nullptr, *d_errs = nullptr, *d_cw = nullptr;
    int *h_it = nullptr, *h_ok = nullptr, *h_errs = nullptr, *h_out = nullptr;
    void release()
    {
        void *dev[] = {d_tx, d_rx, d_Lch, d_con, d_out, d_it, d_ok, d_errs, d_cw};
        for (void *p : dev)
            if (p) (void)hipFree(p);
        void *host[] = {h_rx, h_it, h_ok, h_errs, h_out};
        for (void *p : host)
            if (p) (void)hipHostFree(p);
        d_tx = d_rx = d_Lch = d_con = h_rx = nullptr;
        d_out = d_it = d_ok = d_errs = d_cw = h_it = h_ok = h_errs = h_out = nullptr;
        cap = 0;
    }
    void ensure(int B, int len_, int N_, int q_)
    {
        if (!st) HIP_OR_DIE(hipStreamCreate(&st), "Cannot create a stream, exit!");
        if (B <= cap && len_ == len && N_ == N && q_ == q) return;
        release();
        cap = B; len = len_; N = N_; q = q_;
        bool good = hipMalloc((void **)&d_tx, (size_t)len * 8) == hipSuccess && hipMalloc((void **)&d_rx, (size_t)B * len * 8) == hipSuccess;
        good = good && hipMalloc((void **)&d_Lch, (size_t)B * N * (q - 1) * 4) == hipSuccess && hipMalloc((void **)&d_con, (size_t)q * 8) == hipSuccess;
        good = good && hipMalloc((void **)&d_out, (size_t)B * N * 4) == hipSuccess && hipMalloc((void **)&d_it, (size_t)B * 4) == hipSuccess;
        good = good && hipMalloc((void **)&d_ok, (size_t)B * 4) == hipSuccess && hipMalloc((void **)&d_errs, (size_t)B * 4) == hipSuccess;
        good = good && hipMalloc((void **)&d_cw, (size_t)N * 4) == hipSuccess;
        good = good && hipHostMalloc((void **)&h_rx, (size_t)B * len * 8, hipHostMallocDefault) == hipSuccess;
        good = good && hipHostMalloc((void **)&h_it, (size_t)B * 4, hipHostMallocDefault) == hipSuccess;
        good = good && hipHostMalloc((void **)&h_ok, (size_t)B * 4, hipHostMallocDefault) == hipSuccess;
        good = good && hipHostMalloc((void **)&h_errs, (size_t)B * 4, hipHostMallocDefault) == hipSuccess;
        good = good && hipHostMalloc((void **)&h_out, (size_t)B * N * 4, hipHostMallocDefault) == hipSuccess;
        if (!good) die_msg("Cannot malloc the simulation buffers, exit!");
    }
} g_simbuf;

void run_sim(const LDPCCode *H, AWGNChannel *AWGN, Simulation *SIM, const CComplex *CONSTELLATION, VN *V, CN *C, const CComplex *tx,
             const int *CodeWord_sym, bool gpu_entry)
{
    const SimCfg &s = g_sim;
    const Cfg &c = g_cfg;
    if (!s.set) die_msg("nbldpc_shim_configure_sim was not called, exit!");
    const int method = s.decoder_method;
    if (method < 0 || method > 3) die_msg("decoder_method must be 0, 1, 2 or 3, exit!");
    if (gpu_entry && method == 3) { printf("unfinished\n"); exit(0); } // Simulation.cpp:140-144
    build_code(H, V, C);
    const bool qam = s.n_QAM != 2;
    const int N = H->Variablenode_num, q = c.q, len = qam ? N : H->bit_length; // LDPC_Encoder.cpp:45-52
    SimBuffers &b = g_simbuf;
    b.ensure(std::max(s.batch, 1), len, N, q);
    hipStream_t st = b.st;
    HIP_OR_DIE(hipMemcpyAsync(b.d_cw, CodeWord_sym, (size_t)N * 4, hipMemcpyHostToDevice, st), "Cannot copy CodeWord_sym, exit!");
    HIP_OR_DIE(hipMemcpyAsync(b.d_tx, tx, (size_t)len * 8, hipMemcpyHostToDevice, st), "Cannot copy CComplex_sym, exit!");
    if (qam) HIP_OR_DIE(hipMemcpyAsync(b.d_con, CONSTELLATION, (size_t)q * 8, hipMemcpyHostToDevice, st), "Cannot copy CONSTELLATION, exit!");
    if (gpu_entry) SIM->sumTime = 0; // Simulation.cpp:113
    std::vector<int> seeds_after; // host channel: AWGN->seed after each frame of the batch
    long frame = 0;
    auto unfinished = [&]() { return SIM->num_Error_Frames < s.leastErrorFrames || SIM->num_Frames < s.leastTestFrames; }; // :41, :115
    while (unfinished()) {
        // frames for this launch: enough to meet the frame minimum and, at the error rate seen so far, the error minimum (x 1.25)
        long want = s.batch;
        if (frame == 0) want = std::max<long>(s.leastTestFrames - SIM->num_Frames, 256);
        else if (SIM->num_Error_Frames > 0) {
            const long need_e = std::max<long>(s.leastErrorFrames - SIM->num_Error_Frames, 0);
            want = std::max<long>(need_e * SIM->num_Frames / SIM->num_Error_Frames * 5 / 4 + 16, s.leastTestFrames - SIM->num_Frames);
        }
        const int B = (int)std::min<long>(std::max<long>(want, 64), b.cap);
        int seed0[3] = {AWGN->seed[0], AWGN->seed[1], AWGN->seed[2]};
        if (s.device_channel) {
            if (nbldpc_awgn_channel_device_sym(AWGN->seed, AWGN->sigma, b.d_tx, len, B, qam ? 0 : 1, b.d_rx, st)) die("AWGNChannel (device)");
        } else {
            // the reference's own stream, frame after frame (Simulation.cpp:118-122).  It is ONE serial stream, but an LCG can be
            // advanced k draws at once, so host threads each take a contiguous range of frames starting from the state the serial
            // loop would have there: same draws, same host libm, same bits; AWGN->seed after every frame is kept for the stop rule.
            seeds_after.resize((size_t)3 * B);
            int probe[3] = {seed0[0], seed0[1], seed0[2]};
            const bool jumpable = nbldpc_seed_jump(probe, 0) == NBLDPC_OK; // canonical LCG states only; else serial like the reference
            const unsigned hw = std::thread::hardware_concurrency();
            const int T = jumpable ? (int)std::max<long>(1, std::min<long>({(long)(hw ? hw : 1), 16L, (long)B / 8})) : 1;
            auto gen = [&](int f0, int f1) {
                int sd[3] = {seed0[0], seed0[1], seed0[2]};
                if (f0 && nbldpc_seed_jump(sd, 4ull * (unsigned long long)len * f0)) die("nbldpc_seed_jump");
                std::vector<float> frame_rx((size_t)len * 2);
                for (int f = f0; f < f1; f++) {
                    if (nbldpc_awgn_channel_host_sym(sd, AWGN->sigma, &tx[0].Real, len, frame_rx.data())) die("AWGNChannel_CPU");
                    memcpy(&seeds_after[(size_t)3 * f], sd, 3 * sizeof(int));
                    if (qam) memcpy(b.h_rx + (size_t)f * len * 2, frame_rx.data(), (size_t)len * 8);
                    else
                        for (int i = 0; i < len; i++) b.h_rx[(size_t)f * len + i] = frame_rx[2 * i]; // BPSK: Demodulate reads .Real only (LDPC_Decoder.cpp:142)
                }
            };
            if (T < 2) gen(0, B);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back(gen, (int)((long)B * t / T), (int)((long)B * (t + 1) / T));
                for (auto &x : th) x.join();
            }
            memcpy(AWGN->seed, &seeds_after[(size_t)3 * (B - 1)], 3 * sizeof(int));
            HIP_OR_DIE(hipMemcpyAsync(b.d_rx, b.h_rx, (size_t)B * len * (qam ? 8 : 4), hipMemcpyHostToDevice, st), "Cannot copy the channel output, exit!");
        }
        int rc = qam ? nbldpc_demodulate_qam(g_code, b.d_rx, b.d_con, AWGN->sigma, B, b.d_Lch, st) : nbldpc_demodulate_bpsk(g_code, b.d_rx, AWGN->sigma, B, b.d_Lch, st);
        if (rc) die("Demodulate");
        HIP_OR_DIE(hipStreamSynchronize(st), "channel / Demodulate failed on the device, exit!");
        const auto t0 = std::chrono::steady_clock::now(); // the reference times the decoder call alone (:126,:145)
        if (method == 0) rc = nbldpc_ems_decode_batch(g_code, b.d_Lch, B, s.EMS_NM, s.EMS_NC, c.maxIT, c.dc, b.d_out, b.d_it, b.d_ok, nullptr, nullptr, st);
        else if (method == 2) rc = nbldpc_ems_decode_batch(g_code, b.d_Lch, B, q, c.dc - 1, c.maxIT, c.dc, b.d_out, b.d_it, b.d_ok, nullptr, nullptr, st); // :63-66,:136-139
        else rc = nbldpc_tmm_decode_batch(g_code, b.d_Lch, B, method == 3, c.maxIT, b.d_out, b.d_it, b.d_ok, nullptr, nullptr, st);
        if (rc) die(method == 0 || method == 2 ? "nbldpc_ems_decode_batch" : "nbldpc_tmm_decode_batch");
        HIP_OR_DIE(hipStreamSynchronize(st), "decode failed on the device, exit!");
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / B;
        if (nbldpc_frame_errors(g_code, b.d_out, b.d_cw, B, b.d_errs, st)) die("nbldpc_frame_errors");
        (void)hipMemcpyAsync(b.h_it, b.d_it, (size_t)B * 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(b.h_ok, b.d_ok, (size_t)B * 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(b.h_errs, b.d_errs, (size_t)B * 4, hipMemcpyDeviceToHost, st);
        if (g_hook) (void)hipMemcpyAsync(b.h_out, b.d_out, (size_t)B * N * 4, hipMemcpyDeviceToHost, st);
        HIP_OR_DIE(hipStreamSynchronize(st), "Cannot copy the decoder outputs, exit!");
        int used = 0;
        for (; used < B && unfinished(); used++) { // Simulation.cpp:147-157, one frame at a time, in stream order
            SIM->num_Frames += 1;
            SIM->sumTime += dt;
            SIM->Total_Iteration += b.h_it[used];
            if (g_hook) g_hook(g_hook_user, frame, b.h_out + (size_t)used * N, b.h_it[used], b.h_ok[used]);
            frame++;
            account(SIM, b.h_errs[used], H);
        }
        if (used < B) { // the reference never drew the remaining frames: put AWGN->seed back where it would be
            if (s.device_channel) {
                memcpy(AWGN->seed, seed0, sizeof(seed0));
                if (nbldpc_seed_jump(AWGN->seed, 4ull * (unsigned long long)len * used)) die("nbldpc_seed_jump");
            } else {
                memcpy(AWGN->seed, &seeds_after[(size_t)3 * (used - 1)], 3 * sizeof(int));
            }
        }
    }
    if (SIM->num_Error_Frames >= s.leastErrorFrames && SIM->num_Frames >= s.leastTestFrames) { // Simulation.cpp:191-207, :234-250
        ratios(SIM, H);
        row(SIM, true);
    }
}
} // namespace

// ---- configuration --------------------------------------------------------------------------------------------------------------

extern "C" void nbldpc_shim_sim_defaults(nbldpc_shim_sim_config *cfg)
{
    if (!cfg) return;
    cfg->Matrixfile = "BDS.576.288.GF.64.txt";          // define.h:23
    cfg->Constellationfile = "./Constellation/BPSK.txt"; // :24
    cfg->GFTabledir = "./GF";                            // GF.cpp:81
    cfg->n_QAM = 2; cfg->GFQ = 64; cfg->maxdc = 4; cfg->maxdv = 2; cfg->THREAD_NUM = 1; // :25-29
    cfg->EMS_NM = 2; cfg->EMS_NC = 2; cfg->maxIT = 20; cfg->decoder_method = 0;         // :31-37
    cfg->leastErrorFrames = 50; cfg->leastTestFrames = 1000; cfg->displayStep = 100000;   // :52-54
    cfg->batch = 4096; cfg->device_channel = 0; cfg->results_file = nullptr;
}

extern "C" int nbldpc_shim_configure_sim(const nbldpc_shim_sim_config *cfg)
{
    if (!cfg || !cfg->Matrixfile || !cfg->Constellationfile || !cfg->GFTabledir) return NBLDPC_EINVAL;
    if (cfg->GFQ < 4 || cfg->maxdv <= 0 || cfg->maxdc <= 0 || cfg->maxIT <= 0 || cfg->batch <= 0 || cfg->displayStep <= 0 || cfg->leastErrorFrames < 0 ||
        cfg->leastTestFrames < 0 || (cfg->n_QAM != 2 && cfg->n_QAM != cfg->GFQ) || cfg->EMS_NM < 1 || cfg->EMS_NC < 0)
        return NBLDPC_EINVAL;
    nbldpc_shim_reset();
    g_cfg.q = cfg->GFQ; g_cfg.dv = cfg->maxdv; g_cfg.dc = cfg->maxdc; g_cfg.maxIT = cfg->maxIT;
    g_cfg.mul.clear(); // the table comes from GFInitial
    SimCfg &s = g_sim;
    s.set = true;
    s.Matrixfile = cfg->Matrixfile; s.Constellationfile = cfg->Constellationfile; s.GFTabledir = cfg->GFTabledir;
    s.results_file = cfg->results_file ? cfg->results_file : "";
    s.n_QAM = cfg->n_QAM; s.THREAD_NUM = cfg->THREAD_NUM; s.EMS_NM = cfg->EMS_NM; s.EMS_NC = cfg->EMS_NC; s.decoder_method = cfg->decoder_method;
    s.leastErrorFrames = cfg->leastErrorFrames; s.leastTestFrames = cfg->leastTestFrames; s.displayStep = cfg->displayStep;
    s.batch = cfg->batch; s.device_channel = cfg->device_channel;
    return NBLDPC_OK;
}

extern "C" int nbldpc_shim_configure(int GFQ, int maxdv, int maxdc, int maxIT, const unsigned *TableMultiply_)
{
    if (GFQ < 4 || maxdv <= 0 || maxdc <= 0 || maxIT <= 0 || !TableMultiply_) return NBLDPC_EINVAL;
    nbldpc_shim_reset();
    g_cfg.q = GFQ; g_cfg.dv = maxdv; g_cfg.dc = maxdc; g_cfg.maxIT = maxIT;
    g_cfg.mul.assign(TableMultiply_, TableMultiply_ + (size_t)GFQ * GFQ);
    return NBLDPC_OK;
}

extern "C" void nbldpc_shim_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_code) nbldpc_code_destroy(g_code);
    g_code = nullptr;
}

extern "C" void nbldpc_shim_set_frame_hook(nbldpc_shim_frame_hook hook, void *user)
{
    g_hook = hook;
    g_hook_user = user;
}

// ---- GF.h -------------------------------------------------------------------------------------------------------------------------

unsigned **malloc_2(int xDim, int yDim) // one contiguous block behind a row-pointer array (GF.cpp:22-33): a[0] is the flat table
{
    unsigned **a = (unsigned **)malloc((size_t)xDim * sizeof(unsigned *));
    a[0] = (unsigned *)calloc((size_t)xDim * yDim, sizeof(unsigned));
    for (int i = 1; i < xDim; i++) a[i] = a[0] + (size_t)i * yDim;
    return a;
}
float **malloc_2_float(int xDim, int yDim)
{
    float **a = (float **)malloc((size_t)xDim * sizeof(float *));
    a[0] = (float *)calloc((size_t)xDim * yDim, sizeof(float));
    for (int i = 1; i < xDim; i++) a[i] = a[0] + (size_t)i * yDim;
    return a;
}
int GFAdd(int ele1, int ele2) { return ele1 ^ ele2; }
int GFMultiply(int ele1, int ele2) { return (int)TableMultiply[ele1][ele2]; }
int GFInverse(int ele)
{
    if (ele == 0) { printf("Div 0 Error!\n"); exit(-1); } // GF.cpp:60-64
    return (int)TableInverse[ele];
}
bool GFInitial(int GFq)
{
    TableAdd = malloc_2(GFq, GFq);
    TableMultiply = malloc_2(GFq, GFq);
    TableInverse = (unsigned *)calloc((size_t)GFq, sizeof(unsigned));
    const std::string path = g_sim.GFTabledir + "/Arith.Table.GF." + std::to_string(GFq) + ".txt"; // GF.cpp:78-82
    if (nbldpc_gf_load(path.c_str(), GFq, TableMultiply[0], TableAdd[0], TableInverse)) { fprintf(stderr, "%s\n", nbldpc_last_error()); exit(-1); }
    return true;
}

// ---- Simulation.h / struct.h -----------------------------------------------------------------------------------------------------------

void Get_H(LDPCCode *H, VN *Variablenode, CN *Checknode)
{
    const SimCfg &s = g_sim;
    if (!s.set) die_msg("nbldpc_shim_configure_sim was not called, exit!");
    const int GFQ = g_cfg.q;
    int dims[5];
    if (nbldpc_read_matrix(s.Matrixfile.c_str(), dims, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) { printf("%s\n", nbldpc_last_error()); exit(0); }
    const int N = dims[0], M = dims[1], dv = dims[3], dc = dims[4];
    if (dims[2] != GFQ) { printf("%s is a GF(%d) matrix, configured GFQ is %d, exit!\n", s.Matrixfile.c_str(), dims[2], GFQ); exit(0); }
    std::vector<int> vw(N), vc((size_t)N * dv), vg((size_t)N * dv), cw(M), cv((size_t)M * dc), cg((size_t)M * dc);
    if (nbldpc_read_matrix(s.Matrixfile.c_str(), dims, vw.data(), vc.data(), vg.data(), cw.data(), cv.data(), cg.data())) { printf("%s\n", nbldpc_last_error()); exit(0); }
    H->Variablenode_num = N; H->Checknode_num = M; H->GF = dims[2];
    H->rate = (float)(N - M) / N; // Simulation.cpp:365
    int m = 0;
    while ((1 << m) < GFQ) m++;
    if ((1 << m) != GFQ || m < 2 || m > 8) { printf("error"); exit(0); } // :369-395
    H->q_bit = m;
    H->bit_length = N * m;
    H->maxWeight_variablenode = dv;
    H->maxWeight_checknode = dc;
    for (int t = 0; t < thread_num(); t++) { // one copy of the node arrays per host thread of the reference (:405-464)
        for (int i = 0; i < N; i++) {
            VN &v = Variablenode[(size_t)t * N + i];
            v.weight = vw[i];
            v.linkCNs = (int *)malloc((size_t)std::max(vw[i], 1) * sizeof(int));
            v.linkCNs_GF = (int *)malloc((size_t)std::max(vw[i], 1) * sizeof(int));
            for (int d = 0; d < vw[i]; d++) { v.linkCNs[d] = vc[(size_t)i * dv + d]; v.linkCNs_GF[d] = vg[(size_t)i * dv + d]; }
            v.L_ch = (float *)malloc((size_t)GFQ * sizeof(float));
            v.LLR = (float *)malloc((size_t)GFQ * sizeof(float));
            v.sort_L_v2c = malloc_2_float(std::max(vw[i], 1), GFQ);
            v.sort_Entr_v2c = malloc_2(std::max(vw[i], 1), GFQ);
        }
        for (int r = 0; r < M; r++) {
            CN &c = Checknode[(size_t)t * M + r];
            c.weight = cw[r];
            c.linkVNs = (int *)malloc((size_t)std::max(cw[r], 1) * sizeof(int));
            c.linkVNs_GF = (int *)malloc((size_t)std::max(cw[r], 1) * sizeof(int));
            for (int d = 0; d < cw[r]; d++) { c.linkVNs[d] = cv[(size_t)r * dc + d]; c.linkVNs_GF[d] = cg[(size_t)r * dc + d]; }
            c.L_c2v = malloc_2_float(std::max(cw[r], 1), GFQ);
        }
    }
}

CComplex *Get_CONSTELLATION(LDPCCode *H)
{
    (void)H;
    const SimCfg &s = g_sim;
    if (!s.set) die_msg("nbldpc_shim_configure_sim was not called, exit!");
    CComplex *con = (CComplex *)calloc((size_t)std::max(g_cfg.q, s.n_QAM), sizeof(CComplex)); // GFQ entries (Simulation.cpp:315); the caller frees it
    if (nbldpc_read_constellation(s.Constellationfile.c_str(), s.n_QAM, &con[0].Real)) { printf("%s\n", nbldpc_last_error()); exit(0); }
    return con;
}

void freeVN(const LDPCCode *H, VN *A)
{
    for (size_t i = 0; i < (size_t)H->Variablenode_num * thread_num(); i++) {
        free(A[i].linkCNs); free(A[i].linkCNs_GF); free(A[i].LLR); free(A[i].L_ch);
        free(A[i].sort_L_v2c[0]); free(A[i].sort_L_v2c); free(A[i].sort_Entr_v2c[0]); free(A[i].sort_Entr_v2c);
    }
    free(A);
}
void freeCN(const LDPCCode *H, CN *A)
{
    for (size_t i = 0; i < (size_t)H->Checknode_num * thread_num(); i++) {
        free(A[i].linkVNs); free(A[i].linkVNs_GF); free(A[i].L_c2v[0]); free(A[i].L_c2v);
    }
    free(A);
}

void Simulation_CPU(const LDPCCode *H, AWGNChannel *AWGN, Simulation *SIM, const CComplex *CONSTELLATION, VN *Variablenode, CN *Checknode,
                    const CComplex *CComplex_sym, const int *CodeWord_sym)
{
    run_sim(H, AWGN, SIM, CONSTELLATION, Variablenode, Checknode, CComplex_sym, CodeWord_sym, false);
}

void Simulation_GPU(const LDPCCode *H, AWGNChannel *AWGN, Simulation *SIM, const CComplex *CONSTELLATION, VN *Variablenode, CN *Checknode,
                    const CComplex *CComplex_sym, int *CodeWord_sym, const unsigned *, const unsigned *, const unsigned *, const int *, const int *,
                    const int *, const int *, const int *)
{
    run_sim(H, AWGN, SIM, CONSTELLATION, Variablenode, Checknode, CComplex_sym, CodeWord_sym, true);
}

int Statistic(Simulation *SIM, const int *CodeWord_Frames, int *D, const LDPCCode *H)
{
    int Error_msgBit = 0; // symbols, not bits (Simulation.cpp:264-267)
    for (int i = 0; i < H->Variablenode_num; i++) Error_msgBit += (D[i] != CodeWord_Frames[i]);
    return account(SIM, Error_msgBit, H);
}

// ---- LDPC_Encoder.h ------------------------------------------------------------------------------------------------------------------

void BitToSym(LDPCCode *H, int *CodeWord_sym, int *CodeWord_bit) // bit b of symbol s is CodeWord_bit[q_bit*s + b] (LDPC_Encoder.cpp:6-16)
{
    for (int s = 0; s < H->Variablenode_num; s++) {
        int v = 0;
        for (int b = 0; b < H->q_bit; b++) v |= (CodeWord_bit[H->q_bit * s + b] & 1) << b;
        CodeWord_sym[s] = v;
    }
}

void Modulate(const LDPCCode *H, CComplex *CONSTELLATION, CComplex *CComplex_sym, int *CodeWord_sym)
{
    const int len = g_sim.n_QAM != 2 ? H->Variablenode_num : H->bit_length; // symbols, or the bits the caller passes for BPSK (main.cu:198,211)
    for (int s = 0; s < len; s++) CComplex_sym[s] = CONSTELLATION[CodeWord_sym[s]];
}

void AWGNChannel_CPU(const LDPCCode *H, AWGNChannel *AWGN, CComplex *CComplex_sym_Channelout, const CComplex *CComplex_sym)
{
    const int len = g_sim.n_QAM != 2 ? H->Variablenode_num : H->bit_length;
    if (nbldpc_awgn_channel_host_sym(AWGN->seed, AWGN->sigma, &CComplex_sym[0].Real, len, &CComplex_sym_Channelout[0].Real)) die("AWGNChannel_CPU");
}

float RandomModule(int *seed) { return nbldpc_random_module(seed); }

// ---- LDPC_Decoder.h / Decode_GPU.cuh: one frame per call ------------------------------------------------------------------------------

void Demodulate(const LDPCCode *H, AWGNChannel *AWGN, const CComplex *CONSTELLATION, VN *Variablenode, CComplex *CComplex_sym_Channelout)
{
    const bool qam = g_sim.n_QAM != 2;
    const int N = H->Variablenode_num, q = g_cfg.q, len = qam ? N : H->bit_length;
    if (!q) die_msg("nbldpc_shim_configure / nbldpc_shim_configure_sim was not called, exit!");
    struct Dm {
        hipStream_t st = nullptr;
        float *d_rx = nullptr, *d_con = nullptr, *d_Lch = nullptr;
        std::vector<float> h_rx, h_Lch;
        int N = 0, q = 0, len = 0;
    };
    thread_local Dm d;
    const size_t need = (size_t)N * (q - 1);
    if (!d.st) HIP_OR_DIE(hipStreamCreate(&d.st), "Cannot create a stream, exit!");
    if (N != d.N || q != d.q || len != d.len) {
        if (d.d_rx) { (void)hipFree(d.d_rx); (void)hipFree(d.d_con); (void)hipFree(d.d_Lch); }
        bool good = hipMalloc((void **)&d.d_rx, (size_t)len * 8) == hipSuccess && hipMalloc((void **)&d.d_con, (size_t)q * 8) == hipSuccess &&
                    hipMalloc((void **)&d.d_Lch, need * 4) == hipSuccess;
        if (!good) die_msg("Cannot malloc the demodulator buffers on device, exit!");
        d.N = N; d.q = q; d.len = len;
        d.h_Lch.resize(need);
    }
    d.h_rx.resize((size_t)len * 2);
    int rc;
    if (qam) {
        (void)hipMemcpyAsync(d.d_rx, CComplex_sym_Channelout, (size_t)len * 8, hipMemcpyHostToDevice, d.st);
        (void)hipMemcpyAsync(d.d_con, CONSTELLATION, (size_t)q * 8, hipMemcpyHostToDevice, d.st);
        rc = nbldpc_demodulate_qam_nq(N, q, d.d_rx, d.d_con, AWGN->sigma, 1, d.d_Lch, d.st);
    } else {
        for (int i = 0; i < len; i++) d.h_rx[i] = CComplex_sym_Channelout[i].Real; // LDPC_Decoder.cpp:142
        (void)hipMemcpyAsync(d.d_rx, d.h_rx.data(), (size_t)len * 4, hipMemcpyHostToDevice, d.st);
        rc = nbldpc_demodulate_bpsk_nq(N, q, d.d_rx, AWGN->sigma, 1, d.d_Lch, d.st);
    }
    if (rc) die("Demodulate");
    (void)hipMemcpyAsync(d.h_Lch.data(), d.d_Lch, need * 4, hipMemcpyDeviceToHost, d.st);
    HIP_OR_DIE(hipStreamSynchronize(d.st), "Demodulate failed on the device, exit!");
    for (int i = 0; i < N; i++) memcpy(Variablenode[i].L_ch, &d.h_Lch[(size_t)i * (q - 1)], (size_t)(q - 1) * sizeof(float));
}

int index_in_VN(CN *Checknode, int CNnum, int index_in_linkVNS, VN *Variablenode) // which edge of that variable leads back to check CNnum
{
    const VN &v = Variablenode[Checknode[CNnum].linkVNs[index_in_linkVNS]];
    for (int i = 0; i < v.weight; i++)
        if (v.linkCNs[i] == CNnum) return i;
    printf("index_in_VN error\n");
    exit(0);
}
int index_in_CN(VN *Variablenode, int VNnum, int index_in_linkCNS, CN *Checknode)
{
    const CN &c = Checknode[Variablenode[VNnum].linkCNs[index_in_linkCNS]];
    for (int i = 0; i < c.weight; i++)
        if (c.linkVNs[i] == VNnum) return i;
    printf("index_in_CN error\n");
    exit(0);
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
int Decoding_EMS_GPU(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, const unsigned *, const unsigned *,
                     const int *, const int *, const int *, const int *, const int *, int &iter_number)
{
    return decode_one(0, H, Variablenode, Checknode, EMS_Nm, EMS_Nc, DecodeOutput, iter_number);
}
int Decoding_TMM_GPU(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, const unsigned *, const unsigned *,
                     const unsigned *, const int *, const int *, const int *, const int *, const int *, int &iter_number)
{
    return decode_one(1, H, Variablenode, Checknode, EMS_Nm, EMS_Nc, DecodeOutput, iter_number);
}
