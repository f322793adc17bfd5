// Host-side half of shim/nbldpc_ref_shim.hpp, runnable without a GPU: the class layouts of include/struct.h:9-71 and the
// reference-signature functions that never touch the device (Get_H, GFInitial, Get_CONSTELLATION, BitToSym, Modulate,
// AWGNChannel_CPU, RandomModule, index_in_VN / index_in_CN, Statistic, freeVN / freeCN).  Prints values that
// tests/test_host_cpu.py compares with the CPU oracle.  Run with cwd = data/nb (the reference's relative paths).
// usage: nb_shim_host_test <codeword.txt> <sigma>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nbldpc_ref_shim.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    printf("LAYOUT CComplex %zu LDPCCode %zu VN %zu CN %zu AWGNChannel %zu Simulation %zu\n", sizeof(CComplex), sizeof(LDPCCode), sizeof(VN), sizeof(CN),
           sizeof(AWGNChannel), sizeof(Simulation));
    printf("OFFSETS Simulation %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(Simulation, SNR), offsetof(Simulation, sumTime),
           offsetof(Simulation, num_Frames), offsetof(Simulation, num_Error_Frames), offsetof(Simulation, num_Error_Bits),
           offsetof(Simulation, Total_Iteration), offsetof(Simulation, num_False_Frames), offsetof(Simulation, num_Alarm_Frames),
           offsetof(Simulation, FER), offsetof(Simulation, BER), offsetof(Simulation, AverageIT), offsetof(Simulation, FER_False),
           offsetof(Simulation, FER_Alarm));
    printf("OFFSETS2 VN %zu %zu %zu %zu %zu %zu %zu CN %zu %zu %zu %zu LDPCCode %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(VN, linkCNs),
           offsetof(VN, linkCNs_GF), offsetof(VN, weight), offsetof(VN, LLR), offsetof(VN, L_ch), offsetof(VN, sort_L_v2c), offsetof(VN, sort_Entr_v2c),
           offsetof(CN, linkVNs), offsetof(CN, linkVNs_GF), offsetof(CN, weight), offsetof(CN, L_c2v), offsetof(LDPCCode, maxWeight_checknode),
           offsetof(LDPCCode, maxWeight_variablenode), offsetof(LDPCCode, GF), offsetof(LDPCCode, Variablenode_num), offsetof(LDPCCode, Checknode_num),
           offsetof(LDPCCode, rate), offsetof(LDPCCode, bit_length), offsetof(LDPCCode, q_bit));

    nbldpc_shim_sim_config cfg;
    nbldpc_shim_sim_defaults(&cfg); // define.h as committed: BDS.576.288.GF.64.txt, BPSK, GF(64), maxdc 4, maxdv 2
    cfg.leastErrorFrames = 2; cfg.leastTestFrames = 3; cfg.displayStep = 2;
    if (nbldpc_shim_configure_sim(&cfg)) return 1;
    LDPCCode *H = (LDPCCode *)malloc(sizeof(LDPCCode));
    FILE *fp = fopen(cfg.Matrixfile, "r");
    if (!fp || fscanf(fp, "%d %d", &H->Variablenode_num, &H->Checknode_num) != 2) return 1;
    fclose(fp);
    VN *V = (VN *)malloc((size_t)H->Variablenode_num * sizeof(VN));
    CN *C = (CN *)malloc((size_t)H->Checknode_num * sizeof(CN));
    Get_H(H, V, C);
    long sv = 0, sc = 0;
    for (int i = 0; i < H->Variablenode_num; i++)
        for (int d = 0; d < V[i].weight; d++) sv += (long)(i + 1) * (V[i].linkCNs[d] + 3 * V[i].linkCNs_GF[d] + d);
    for (int r = 0; r < H->Checknode_num; r++)
        for (int d = 0; d < C[r].weight; d++) sc += (long)(r + 1) * (C[r].linkVNs[d] + 3 * C[r].linkVNs_GF[d] + index_in_VN(C, r, d, V));
    printf("GET_H %d %d %d %.9g %d %d %d %d %ld %ld %d\n", H->Variablenode_num, H->Checknode_num, H->GF, H->rate, H->q_bit, H->bit_length,
           H->maxWeight_variablenode, H->maxWeight_checknode, sv, sc, index_in_CN(V, 5, 1, C));
    GFInitial(cfg.GFQ);
    printf("GF %u %u %u %d %d %d\n", TableMultiply[2][33], TableAdd[5][9], TableInverse[5], GFMultiply(7, 9), GFAdd(7, 9), GFInverse(13));
    CComplex *CON = Get_CONSTELLATION(H);
    printf("CON %.9g %.9g %.9g %.9g\n", CON[0].Real, CON[0].Image, CON[1].Real, CON[1].Image);

    int *cw = (int *)calloc((size_t)H->Variablenode_num, sizeof(int)), *bits = (int *)calloc((size_t)H->bit_length, sizeof(int));
    int *cw2 = (int *)calloc((size_t)H->Variablenode_num, sizeof(int));
    fp = fopen(argv[1], "r");
    for (int i = 0; fp && i < H->Variablenode_num; i++)
        if (fscanf(fp, "%d", &cw[i]) != 1) return 1;
    if (fp) fclose(fp);
    for (int i = 0; i < H->Variablenode_num; i++)
        for (int j = 0; j < H->q_bit; j++) bits[i * H->q_bit + j] = (cw[i] & (1 << j)) >> j; // main.cu:203-209
    BitToSym(H, cw2, bits);
    printf("BITTOSYM %d\n", memcmp(cw, cw2, (size_t)H->Variablenode_num * sizeof(int)) == 0);
    CComplex *tx = (CComplex *)malloc((size_t)H->bit_length * sizeof(CComplex)), *rx = (CComplex *)malloc((size_t)H->bit_length * sizeof(CComplex));
    Modulate(H, CON, tx, bits);
    AWGNChannel AWGN;
    AWGN.seed[0] = AWGN.seed[1] = AWGN.seed[2] = 173;
    AWGN.sigma = (float)atof(argv[2]);
    AWGNChannel_CPU(H, &AWGN, rx, tx);
    unsigned h = 2166136261u;
    for (int i = 0; i < H->bit_length; i++) { unsigned u; memcpy(&u, &rx[i].Real, 4); h = (h ^ u) * 16777619u; }
    printf("AWGN %.9g %.9g %.9g %08x %d %d %d\n", rx[0].Real, rx[1].Real, rx[H->bit_length - 1].Real, h, AWGN.seed[0], AWGN.seed[1], AWGN.seed[2]);
    int s2[3] = {173, 173, 173};
    printf("RANDOM %.9g %.9g\n", RandomModule(s2), RandomModule(s2));

    Simulation SIM;
    memset(&SIM, 0, sizeof(SIM));
    SIM.SNR = 3.0f;
    int rets[4];
    for (int f = 0; f < 4; f++) { // frames with 0, 3, 0, 1 wrong symbols: the stop rule (2 error frames, 3 frames) is met at the fourth
        memcpy(cw2, cw, (size_t)H->Variablenode_num * sizeof(int));
        if (f == 1) { cw2[0] ^= 1; cw2[7] ^= 5; cw2[95] ^= 2; }
        if (f == 3) cw2[40] ^= 9;
        SIM.num_Frames += 1; SIM.Total_Iteration += 2 + f; SIM.sumTime += 1e-3; // as decode_once_* does before the call (Simulation.cpp:149-153)
        rets[f] = Statistic(&SIM, cw, cw2, H);
    }
    printf("STAT %d %d %d %d %ld %ld %ld %ld %.9g %.9g %.9g\n", rets[0], rets[1], rets[2], rets[3], SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits,
           SIM.Total_Iteration, SIM.FER, SIM.BER, SIM.AverageIT);
    freeCN(H, C); freeVN(H, V);
    free(H); free(cw); free(cw2); free(bits); free(tx); free(rx); free(CON);
    return 0;
}
