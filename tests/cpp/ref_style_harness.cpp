// A harness written the way the reference's main()/Simulation_GPU drive the decoder (bldpc_实习/main.cu:92-98,
// Simulation.cu:74,138-143): host Get_H/Transform_H-style table, device Channel_Out and Address_Variablenode, then
// LDPC_Decoder_GPU(D, Channel_Out_GPU, prop, Address_Variablenode_GPU, Weight_Checknode, Weight_Variablenode, LDPC)
// with the reference's signature (shim/ldpc_ref_shim.hpp).  Prints the fold hash of D and iteraTime.
// usage: ref_style_harness <BlockH.txt> J L Z F SNR as_written(0|1)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bldpc.h"
#include "ldpc_ref_shim.hpp"

int main(int argc, char **argv)
{
    if (argc < 8) return 2;
    const int J = atoi(argv[2]), L = atoi(argv[3]), Z = atoi(argv[4]), F = atoi(argv[5]);
    const float snr = (float)atof(argv[6]);
    const int as_written = atoi(argv[7]);
    const int N = L * Z, K = N - J * Z;
    std::vector<int> H(J * L), Weight_Checknode(J + 1), Weight_Variablenode(L + 1);
    if (bldpc_read_blockh(argv[1], J, L, H.data(), Weight_Checknode.data(), Weight_Variablenode.data())) { printf("%s\n", bldpc_last_error()); return 1; }
    std::vector<int> Address_Variablenode((size_t)N * Weight_Variablenode[L]);
    bldpc_transform_h(H.data(), J, L, Z, Weight_Checknode.data(), Weight_Variablenode.data(), Address_Variablenode.data(), as_written);
    int *Address_Variablenode_GPU = nullptr;
    (void)hipMalloc((void **)&Address_Variablenode_GPU, Address_Variablenode.size() * sizeof(int));
    (void)hipMemcpy(Address_Variablenode_GPU, Address_Variablenode.data(), Address_Variablenode.size() * sizeof(int), hipMemcpyHostToDevice);

    int seed[3] = {173, 173, 173};
    std::vector<float> Channel_Out((size_t)N * F);
    bldpc_awgn_channel_host(seed, bldpc_sigma(snr, 1, 0.0f), Channel_Out.data(), nullptr, N, F);
    float *Channel_Out_GPU = nullptr;
    (void)hipMalloc((void **)&Channel_Out_GPU, Channel_Out.size() * sizeof(float));
    (void)hipMemcpy(Channel_Out_GPU, Channel_Out.data(), Channel_Out.size() * sizeof(float), hipMemcpyHostToDevice);

    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    std::vector<int> D((size_t)(N + 1) * F);
    LDPCCode LDPC;
    if (bldpc_shim_configure(J, L, Z, F, K, 50)) return 1;
    LDPC_Decoder_GPU(D.data(), Channel_Out_GPU, prop, Address_Variablenode_GPU, Weight_Checknode.data(), Weight_Variablenode.data(), &LDPC);

    unsigned h = 2166136261u;
    for (size_t i = 0; i < (size_t)N * F; i++) h = (h ^ (unsigned)D[i]) * 16777619u;
    int flags = 0;
    for (int f = 0; f < F; f++) flags += D[(size_t)N * F + f];
    printf("hash=%08x iteraTime=%d flags=%d/%d\n", h, LDPC.iteraTime, flags, F);
    return 0;
}
