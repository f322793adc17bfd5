// A sweep written the way the reference's NB main() is (myNBLDPC/src/main.cu:14-268): Get_H, GFInitial, Get_CONSTELLATION, the
// tables and the flattened graph copied to the device, the codeword modulated (:190-212), then for every Eb/N0 point seeds reset
// to 173/173/173, sigma from snrtype 0, counters cleared and Simulation_GPU (CPU_GPU 1, the reference's default) or Simulation_CPU
// -- all with the reference's signatures and classes (shim/nbldpc_ref_shim.hpp), linked against the shim and the library only.
// What define.h fixes at compile time comes from the command line.  Run with the reference's working directory layout as cwd
// (data/nb: Matrixfile, ./GF/, ./Constellation/ are relative paths as in define.h:23-24, GF.cpp:81).
// Prints the reference's result rows and, per point, "POINT snr frames error_frames symbol_errors total_iteration seed0 seed1 seed2";
// with dump > 0 also "FRAME i ok it hash(DecodeOutput)" for the first `dump` frames of the first point (test hook).
// usage: nb_ref_main_style_sweep <Matrixfile> <codeword.txt|zero> GFQ maxdc maxdv decoder_method CPU_GPU startSNR stopSNR stepSNR
//                                leastErrorFrames leastTestFrames batch device_channel [n_QAM Constellationfile] [dump]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nbldpc_ref_shim.hpp"

static int g_dump = 0;
static void frame_hook(void *, long frame, const int *DecodeOutput, int iter_number, int ok)
{
    if (frame >= g_dump) return;
    unsigned h = 2166136261u;
    const int N = 96; // hashed prefix: the whole word of the BDS code; enough to tell frames apart elsewhere
    for (int i = 0; i < N; i++) h = (h ^ (unsigned)DecodeOutput[i]) * 16777619u;
    printf("FRAME %ld %d %d %08x\n", frame, ok, iter_number, h);
}

int main(int argc, char **argv)
{
    if (argc < 15) return 2;
    nbldpc_shim_sim_config cfg;
    nbldpc_shim_sim_defaults(&cfg);
    cfg.Matrixfile = argv[1];
    const char *cwfile = argv[2];
    cfg.GFQ = atoi(argv[3]); cfg.maxdc = atoi(argv[4]); cfg.maxdv = atoi(argv[5]); cfg.decoder_method = atoi(argv[6]);
    const int CPU_GPU = atoi(argv[7]);
    const double startSNR = atof(argv[8]), stopSNR = atof(argv[9]), stepSNR = atof(argv[10]);
    cfg.leastErrorFrames = atol(argv[11]); cfg.leastTestFrames = atol(argv[12]); cfg.batch = atoi(argv[13]); cfg.device_channel = atoi(argv[14]);
    if (argc > 16) { cfg.n_QAM = atoi(argv[15]); cfg.Constellationfile = argv[16]; }
    g_dump = argc > 17 ? atoi(argv[17]) : (argc == 16 ? atoi(argv[15]) : 0);
    const int n_QAM = cfg.n_QAM, GFQ = cfg.GFQ, maxdc = cfg.maxdc, maxdv = cfg.maxdv, snrtype = 0;
    if (nbldpc_shim_configure_sim(&cfg)) { printf("bad configuration\n"); return 1; }
    if (g_dump) nbldpc_shim_set_frame_hook(frame_hook, nullptr);

    int Num_Device = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceCount(&Num_Device) != hipSuccess || Num_Device < 1) { printf("There is no GPU beyond 1.0, exit!\n"); exit(0); }
    if (hipGetDeviceProperties(&prop, Num_Device - 1) != hipSuccess) { printf("Cannot get device properties, exit!\n"); exit(0); }
    printf("Device Name : %s.\n", prop.name);

    AWGNChannel *AWGN = (AWGNChannel *)malloc(sizeof(AWGNChannel));
    Simulation *SIM = (Simulation *)malloc(sizeof(Simulation));
    LDPCCode *H = (LDPCCode *)malloc(sizeof(LDPCCode));
    FILE *fp_H = fopen(cfg.Matrixfile, "r"); // the node counts first, to size the arrays (main.cu:55-70)
    if (!fp_H) { printf("can not open file: %s\n", cfg.Matrixfile); exit(0); }
    if (fscanf(fp_H, "%d %d", &H->Variablenode_num, &H->Checknode_num) != 2) return 1;
    fclose(fp_H);
    const int threadNum = cfg.THREAD_NUM;
    VN *Variablenode = (VN *)malloc((size_t)H->Variablenode_num * threadNum * sizeof(VN));
    CN *Checknode = (CN *)malloc((size_t)H->Checknode_num * threadNum * sizeof(CN));
    Get_H(H, Variablenode, Checknode);
    GFInitial(GFQ);
    CComplex *CONSTELLATION = Get_CONSTELLATION(H);

    // tables and flattened graph on the device, as main.cu:89-188 prepares them for its own kernels (the shim ignores them)
    unsigned *TableMultiply_GPU, *TableAdd_GPU, *TableInverse_GPU;
    int *Checknode_weight, *Variablenode_weight, *Variablenode_linkCNs, *Checknode_linkVNs, *Checknode_linkVNs_GF;
    bool up = hipMalloc((void **)&TableMultiply_GPU, (size_t)GFQ * GFQ * sizeof(unsigned)) == hipSuccess &&
              hipMalloc((void **)&TableAdd_GPU, (size_t)GFQ * GFQ * sizeof(unsigned)) == hipSuccess &&
              hipMalloc((void **)&TableInverse_GPU, (size_t)GFQ * sizeof(unsigned)) == hipSuccess &&
              hipMalloc((void **)&Checknode_weight, (size_t)H->Checknode_num * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&Variablenode_weight, (size_t)H->Variablenode_num * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&Variablenode_linkCNs, (size_t)H->Variablenode_num * maxdv * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&Checknode_linkVNs, (size_t)H->Checknode_num * maxdc * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&Checknode_linkVNs_GF, (size_t)H->Checknode_num * maxdc * sizeof(int)) == hipSuccess;
    up = up && hipMemcpy(TableMultiply_GPU, TableMultiply[0], (size_t)GFQ * GFQ * sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(TableAdd_GPU, TableAdd[0], (size_t)GFQ * GFQ * sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(TableInverse_GPU, TableInverse, (size_t)GFQ * sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess;
    if (!up) { printf("Cannot copy the tables\n"); exit(0); }
    {
        int *tmp = (int *)calloc((size_t)H->Variablenode_num * maxdv + (size_t)H->Checknode_num * maxdc, sizeof(int));
        for (int i = 0; i < H->Variablenode_num; i++) // pre-multiplied offsets of the reference's message arrays (main.cu:137-143)
            for (int j = 0; j < Variablenode[i].weight; j++)
                tmp[i * maxdv + j] = Variablenode[i].linkCNs[j] * GFQ * maxdc + index_in_CN(Variablenode, i, j, Checknode) * GFQ;
        (void)hipMemcpy(Variablenode_linkCNs, tmp, (size_t)H->Variablenode_num * maxdv * sizeof(int), hipMemcpyHostToDevice);
        for (int i = 0; i < H->Checknode_num; i++) // :156-162
            for (int j = 0; j < Checknode[i].weight; j++)
                tmp[i * maxdc + j] = Checknode[i].linkVNs[j] * GFQ * maxdv + index_in_VN(Checknode, i, j, Variablenode) * GFQ;
        (void)hipMemcpy(Checknode_linkVNs, tmp, (size_t)H->Checknode_num * maxdc * sizeof(int), hipMemcpyHostToDevice);
        free(tmp);
    }

    // the transmitted word: codeword_test.h in the reference (main.cu:190-212), a text file here ("zero": the all-zero word)
    int *CodeWord_bit = (int *)calloc((size_t)H->bit_length, sizeof(int));
    int *CodeWord_sym = (int *)calloc((size_t)H->Variablenode_num, sizeof(int));
    int *CodeWord_sym_test = (int *)calloc((size_t)H->Variablenode_num, sizeof(int));
    if (strcmp(cwfile, "zero") != 0) {
        FILE *fp = fopen(cwfile, "r");
        if (!fp) { printf("can not open file: %s\n", cwfile); exit(0); }
        for (int i = 0; i < H->Variablenode_num; i++)
            if (fscanf(fp, "%d", &CodeWord_sym_test[i]) != 1) return 1;
        fclose(fp);
    }
    CComplex *CComplex_sym;
    if (n_QAM != 2) {
        CComplex_sym = (CComplex *)malloc((size_t)H->Variablenode_num * sizeof(CComplex));
        for (int i = 0; i < H->Variablenode_num; i++) CodeWord_sym[i] = CodeWord_sym_test[i];
        Modulate(H, CONSTELLATION, CComplex_sym, CodeWord_sym);
    } else {
        CComplex_sym = (CComplex *)malloc((size_t)H->bit_length * sizeof(CComplex));
        for (int i = 0; i < H->Variablenode_num; i++)
            for (int j = 0; j < H->q_bit; j++) CodeWord_bit[i * H->q_bit + j] = (CodeWord_sym_test[i] & (1 << j)) >> j;
        BitToSym(H, CodeWord_sym, CodeWord_bit);
        Modulate(H, CONSTELLATION, CComplex_sym, CodeWord_bit);
    }

    printf("sim start\n");
    for (SIM->SNR = (float)startSNR; SIM->SNR <= stopSNR; SIM->SNR += stepSNR) { // a float advanced by a double step (main.cu:215)
        AWGN->seed[0] = 173; AWGN->seed[1] = 173; AWGN->seed[2] = 173; // define.h:41-43
        if (snrtype == 0) AWGN->sigma = (float)sqrt(0.5 / (log(n_QAM) / log(2) * H->rate * (pow(10.0, (SIM->SNR / 10.0))))); // main.cu:223
        else AWGN->sigma = (float)sqrt(0.5 / (log(n_QAM) / log(2) * pow(10.0, (SIM->SNR / 10.0))));
        SIM->num_Frames = 0; SIM->num_Error_Frames = 0; SIM->num_Error_Bits = 0; SIM->Total_Iteration = 0;
        SIM->num_False_Frames = 0; SIM->num_Alarm_Frames = 0; SIM->sumTime = 0;
        SIM->FER = 0; SIM->BER = 0; SIM->AverageIT = 0; SIM->FER_False = 0; SIM->FER_Alarm = 0;
        if (!CPU_GPU)
            Simulation_CPU((const LDPCCode *)H, AWGN, SIM, (const CComplex *)CONSTELLATION, Variablenode, Checknode, (const CComplex *)CComplex_sym,
                           (const int *)CodeWord_sym);
        else
            Simulation_GPU((const LDPCCode *)H, AWGN, SIM, (const CComplex *)CONSTELLATION, Variablenode, Checknode, (const CComplex *)CComplex_sym,
                           CodeWord_sym, (const unsigned *)TableMultiply_GPU, (const unsigned *)TableAdd_GPU, (const unsigned *)TableInverse_GPU,
                           (const int *)Variablenode_weight, (const int *)Checknode_weight, (const int *)Variablenode_linkCNs,
                           (const int *)Checknode_linkVNs, (const int *)Checknode_linkVNs_GF);
        printf("POINT %.9g %ld %ld %ld %ld %d %d %d\n", SIM->SNR, SIM->num_Frames, SIM->num_Error_Frames, SIM->num_Error_Bits, SIM->Total_Iteration,
               AWGN->seed[0], AWGN->seed[1], AWGN->seed[2]);
        g_dump = 0; // frames of the first point only
    }
    (void)hipFree(TableMultiply_GPU); (void)hipFree(TableAdd_GPU); (void)hipFree(TableInverse_GPU); (void)hipFree(Checknode_weight);
    (void)hipFree(Variablenode_weight); (void)hipFree(Variablenode_linkCNs); (void)hipFree(Checknode_linkVNs); (void)hipFree(Checknode_linkVNs_GF);
    free(AWGN); free(SIM);
    freeCN(H, Checknode); freeVN(H, Variablenode);
    free(H); free(CodeWord_sym); free(CodeWord_bit); free(CodeWord_sym_test); free(CComplex_sym); free(CONSTELLATION);
    printf("\ntask finish\n");
    return 0;
}
