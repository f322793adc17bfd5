// ldpc_ref_shim.hpp -- the binary program's own entry points, same C++ signatures, on top of the C ABI (include/bldpc.h).
//
// gsw4869/CUDA_LDPC's binary harness (bldpc_实习/main.cu:114-160) calls, with every shape a compile-time macro of define.cuh:
//     void Get_H(int* H, int* Weight_Checknode, int* Weight_Variablenode);                              (Simulation.cuh:10)
//     void Transform_H(int* H, int* Weight_Checknode, int* Weight_Variablenode, int* Address_Variablenode);   (:12)
//     void Simulation_GPU(AWGNChannel* AWGN, float* sigma_GPU, Simulation* SIM, int* Address_Variablenode,
//                         int* Weight_Checknode, int* Weight_Variablenode);                                    (:4)
//     int  Statistic(Simulation* SIM, int* CodeWord_Frames, int* D, LDPCCode* LDPC);                           (:8)
//     void LDPC_Decoder_GPU(int* D, float* Channel_Out, cudaDeviceProp prop, int* Address_Variablenode,
//                           int* Weight_Checknode, int* Weight_Variablenode, LDPCCode* LDPC);          (LDPC_Decoder.cuh:5)
// and the structs of struct.cuh:6-33.  This header declares exactly those, implemented in shim/ldpc_ref_shim.hip on
// bldpc_read_blockh / bldpc_transform_h / bldpc_awgn_channel_host / bldpc_decode.  The macros of define.cuh become run-time
// calls made once before the sweep:
//     bldpc_shim_configure(J, L, Z, Num_Frames_OneTime, msgLen-or-CW_Len, maxIT);
//     bldpc_shim_configure_sim(path, as_written, leastErrorFrames, leastTestFrames, displayStep);   (optional)
#pragma once
#include <hip/hip_runtime.h>

typedef hipDeviceProp_t cudaDeviceProp; // the one CUDA type name in the reference's signatures (only maxThreadsPerBlock is read there)

typedef struct
{
    int iteraTime; // iterations used by the last decode (struct.cuh:6-9)
} LDPCCode;

typedef struct
{
    int seed[3]; // RandomModule state (struct.cuh:10-14)
    float sigma;
} AWGNChannel;

typedef struct // struct.cuh:16-33, field for field
{
    float SNR;
    long num_Frames, num_Error_Frames, num_Error_Bits, Total_Iteration, num_False_Frames, num_Alarm_Frames;
    float FER, BER, AverageIT, FER_False, FER_Alarm;
} Simulation;

// J, L, Z: define.cuh:20-22; frames: Num_Frames_OneTime (:60); length: msgLen when Message_CW == 0 else CW_Len
// (LDPC_Decoder.cu:36); maxIT: define.cuh:35.  Returns 0, or a BLDPC_E* code (bldpc_last_error() has the text).
extern "C" int bldpc_shim_configure(int J, int L, int Z, int frames, int length, int maxIT);
// path: the BlockH file Get_H reads (the reference hard-codes "PON_LDPC.txt", Simulation.cu:296: the default here too);
// as_written: 1 = Transform_H reproduces Simulation.cu:380 literally (the reference's table, default), 0 = the intended
// circulant; leastErrorFrames / leastTestFrames / displayStep: define.cuh:52-54 (defaults 50 / 10000 / 40960).
extern "C" int bldpc_shim_configure_sim(const char *path, int as_written, long leastErrorFrames, long leastTestFrames, long displayStep);
// The fast path through the same Simulation_GPU (default: all zero = the reference's data flow, bit-exact host noise stream):
//   device_channel     1: bldpc_awgn_channel_device instead of the host AWGNChannel_CPU + upload (same RandomModule draws by LCG
//                         jump-ahead, device libm in the Box-Muller transform: a sample may differ by an ulp)
//   device_statistics  1: bldpc_decode_statistic -- decode and Statistic in one call on the device, only the five counters come
//                         back per batch (no 4 N F-byte copy of D, no host loop over it)
//   exit_mode          BLDPC_EXIT_BATCH_GLOBAL (the reference's rule, default), BLDPC_EXIT_FIXED, or BLDPC_EXIT_PER_FRAME (every
//                         frame stops on its own flag = the reference with Num_Frames_OneTime 1; needs device_statistics)
//   max_batches        stop a point after this many batches even if the stop rule is not met (0 = never; deep sweeps)
// Counters, rows and seeds are those of the Python mirror cuda_ldpc_amd.simulation.Simulation_GPU with the same switches.
extern "C" int bldpc_shim_configure_fast(int device_channel, int device_statistics, int exit_mode, long max_batches);
extern "C" void bldpc_shim_reset(void); // drop the cached code object (e.g. before switching matrices)
extern "C" const char *bldpc_shim_last_kernel(void); // which kernel tier the last LDPC_Decoder_GPU call ran on

void Get_H(int *H, int *Weight_Checknode, int *Weight_Variablenode);
void Transform_H(int *H, int *Weight_Checknode, int *Weight_Variablenode, int *Address_Variablenode);

// Pointer spaces as in the reference: D host [(N+1)*F]; Channel_Out and Address_Variablenode DEVICE; weights host.
// Failures print the message and exit(0), which is what the reference does on every error path.
// A table that equals the intended circulant expansion of some shift matrix runs on the fused on-chip kernels
// (the shifts are read back from it); any other table, the reference's as-written one included, on the table kernels.
void LDPC_Decoder_GPU(int *D, float *Channel_Out, cudaDeviceProp prop, int *Address_Variablenode, int *Weight_Checknode,
                      int *Weight_Variablenode, LDPCCode *LDPC);

// One SNR point (Simulation.cu:12-171): batches of Num_Frames_OneTime all-zero codewords through AWGNChannel_CPU (host,
// the reference's own noise stream, AWGN->seed advanced in place), LDPC_Decoder_GPU and Statistic until Statistic returns 1.
// sigma_GPU is accepted and unused, as in the reference (its BPSK kernel path is dead code with Add_noise 1).
void Simulation_GPU(AWGNChannel *AWGN, float *sigma_GPU, Simulation *SIM, int *Address_Variablenode, int *Weight_Checknode,
                    int *Weight_Variablenode);
// Simulation.cu:245-285: counters, the result row every displayStep frames, 1 when the stop rule is met.
int Statistic(Simulation *SIM, int *CodeWord_Frames, int *D, LDPCCode *LDPC);
