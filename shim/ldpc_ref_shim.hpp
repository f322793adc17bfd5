// ldpc_ref_shim.hpp -- the reference's own entry point, same C++ signature, on top of the C ABI.
//
// gsw4869/CUDA_LDPC's binary harness (bldpc_实习/Simulation.cu:143) calls
//     void LDPC_Decoder_GPU(int* D, float* Channel_Out, cudaDeviceProp prop, int* Address_Variablenode,
//                           int* Weight_Checknode, int* Weight_Variablenode, LDPCCode* LDPC);   (LDPC_Decoder.cuh:5)
// with every shape a compile-time macro of define.cuh.  Built with hipcc the harness spells cudaDeviceProp as
// hipDeviceProp_t; this header declares exactly that function, implemented in shim/ldpc_ref_shim.hip by one call
// to bldpc_decode.  The macros of define.cuh become one run-time call before the first decode:
//     bldpc_shim_configure(J, L, Z, Num_Frames_OneTime, msgLen-or-CW_Len, maxIT);
#pragma once
#include <hip/hip_runtime.h>

typedef struct
{
    int iteraTime; // iterations used by the last decode (struct.cuh:6-9)
} LDPCCode;

// J, L, Z: define.cuh:20-22; frames: Num_Frames_OneTime (:60); length: msgLen when Message_CW == 0 else CW_Len
// (LDPC_Decoder.cu:36); maxIT: define.cuh:35.  Returns 0, or a BLDPC_E* code (bldpc_last_error() has the text).
extern "C" int bldpc_shim_configure(int J, int L, int Z, int frames, int length, int maxIT);
extern "C" void bldpc_shim_reset(void); // drop the cached code object (e.g. before switching matrices)

// Pointer spaces as in the reference: D host [(N+1)*F]; Channel_Out and Address_Variablenode DEVICE; weights host.
// Failures print the message and exit(0), which is what the reference does on every error path.
void LDPC_Decoder_GPU(int *D, float *Channel_Out, hipDeviceProp_t prop, int *Address_Variablenode, int *Weight_Checknode,
                      int *Weight_Variablenode, LDPCCode *LDPC);
