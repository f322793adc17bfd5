// A sweep written the way the reference's main() is (bldpc_实习/main.cu:60-160): Get_H, Transform_H, the table copied to
// the device, then for every SNR point seeds reset to 173/173/173, sigma from snrtype 1, counters cleared, Simulation_GPU --
// all with the reference's signatures and structs (shim/ldpc_ref_shim.hpp).  The shapes that are macros of define.cuh
// come from the command line.  Prints the reference's result rows and, per point, a line "POINT snr frames error_frames
// error_bits total_iteration false alarm" for the test that compares it with cuda_ldpc_amd.simulation.
// usage: ref_main_style_sweep <BlockH.txt> J L Z F maxIT as_written startSNR stopSNR stepSNR leastErrorFrames leastTestFrames
//                             [device_channel device_statistics exit_mode max_batches]     (bldpc_shim_configure_fast; default 0 0 1 0)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ldpc_ref_shim.hpp"

int main(int argc, char **argv)
{
    if (argc < 13) return 2;
    const int J = atoi(argv[2]), L = atoi(argv[3]), Z = atoi(argv[4]), F = atoi(argv[5]), maxIT = atoi(argv[6]), as_written = atoi(argv[7]);
    const double startSNR = atof(argv[8]), stopSNR = atof(argv[9]), stepSNR = atof(argv[10]);
    const int CW_Len = L * Z, msgLen = CW_Len - J * Z;
    if (bldpc_shim_configure(J, L, Z, F, msgLen, maxIT)) return 1;
    if (bldpc_shim_configure_sim(argv[1], as_written, atol(argv[11]), atol(argv[12]), argc >= 17 ? (1L << 62) : 40960)) return 1;
    if (argc >= 17 && bldpc_shim_configure_fast(atoi(argv[13]), atoi(argv[14]), atoi(argv[15]), atol(argv[16]))) return 1;

    AWGNChannel *AWGN = (AWGNChannel *)malloc(sizeof(AWGNChannel));
    Simulation *SIM = (Simulation *)malloc(sizeof(Simulation));
    std::vector<int> H(J * L), Weight_Checknode(J + 1), Weight_Variablenode(L + 1);
    Get_H(H.data(), Weight_Checknode.data(), Weight_Variablenode.data());
    std::vector<int> Address_Variablenode((size_t)CW_Len * Weight_Variablenode[L]);
    Transform_H(H.data(), Weight_Checknode.data(), Weight_Variablenode.data(), Address_Variablenode.data());
    int *Address_Variablenode_GPU = nullptr;
    float *sigma_GPU = nullptr;
    if (hipMalloc((void **)&Address_Variablenode_GPU, Address_Variablenode.size() * sizeof(int)) != hipSuccess) return 1;
    if (hipMalloc((void **)&sigma_GPU, sizeof(float)) != hipSuccess) return 1;
    if (hipMemcpy(Address_Variablenode_GPU, Address_Variablenode.data(), Address_Variablenode.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return 1;

    for (SIM->SNR = (float)startSNR; SIM->SNR <= stopSNR; SIM->SNR += stepSNR) { // a float advanced by a double step (main.cu:114)
        AWGN->seed[0] = 173; AWGN->seed[1] = 173; AWGN->seed[2] = 173;
        AWGN->sigma = (float)sqrt(0.5 / (pow(10.0, (SIM->SNR / 10.0)))); // snrtype 1 (main.cu:126)
        SIM->num_Frames = 0; SIM->num_Error_Frames = 0; SIM->num_Error_Bits = 0;
        SIM->Total_Iteration = 0; SIM->num_False_Frames = 0; SIM->num_Alarm_Frames = 0;
        (void)hipMemcpy(sigma_GPU, &AWGN->sigma, sizeof(float), hipMemcpyHostToDevice);
        Simulation_GPU(AWGN, sigma_GPU, SIM, Address_Variablenode_GPU, Weight_Checknode.data(), Weight_Variablenode.data());
        (void)hipDeviceSynchronize();
        printf("POINT %.9g %ld %ld %ld %ld %ld %ld kernel=%s\n", SIM->SNR, SIM->num_Frames, SIM->num_Error_Frames, SIM->num_Error_Bits,
               SIM->Total_Iteration, SIM->num_False_Frames, SIM->num_Alarm_Frames, bldpc_shim_last_kernel());
    }
    free(AWGN); free(SIM);
    (void)hipFree(sigma_GPU); (void)hipFree(Address_Variablenode_GPU);
    printf("\ntask finish\n");
    return 0;
}
