// nbldpc_ref_shim.hpp -- the non-binary program's own entry points, same C++ signatures, on top of the C ABI (include/nbldpc.h).
//
// gsw4869/CUDA_LDPC's NB program (myNBLDPC/src/main.cu:14-268) is: Get_H, GFInitial, Get_CONSTELLATION, Modulate, then per
// Eb/N0 point Simulation_CPU or -- its default, CPU_GPU 1 (include/define.h:59) -- Simulation_GPU, whose worker
// decode_once_gpu (src/Simulation.cpp:89-161) calls AWGNChannel_CPU, Demodulate, Decoding_EMS_GPU / Decoding_TMM_GPU and
// Statistic once per frame.  This header declares every one of those names with the reference's signature
// (include/Simulation.h:8-20, include/Decode_GPU.cuh:17,19, include/LDPC_Decoder.h:9-25, include/LDPC_Encoder.h:7-15,
// include/GF.h:7-19, include/struct.h:73-76) and the classes of include/struct.h:9-71 field for field, so that a main()
// written like the reference's links against shim/nbldpc_ref_shim.hip + libcuda_ldpc_amd.so and nothing else
// (tests/cpp/nb_ref_main_style_sweep.cpp).  What the reference fixes with macros of define.h at compile time is ONE run-time
// call made before Get_H:
//     nbldpc_shim_sim_config cfg; nbldpc_shim_sim_defaults(&cfg);   // the values of define.h as committed
//     cfg.Matrixfile = "..."; ...; nbldpc_shim_configure_sim(&cfg);
// The frame loop of the reference (one frame per decoder call, stop rule tested before every frame, Simulation.cpp:115) lives
// INSIDE Simulation_CPU / Simulation_GPU here, and that is where batching lives: cfg.batch frames are generated, demodulated
// and decoded per launch, their results are then accounted one by one in stream order and the accounting stops at the frame at
// which the reference's while-condition fails; AWGN->seed is left where the reference would have left it.  Counters, printed
// rows and seeds are those of a reference run with THREAD_NUM 1.
// Failures print the message and exit(0), which is what the reference does on every error path.
#pragma once

class CComplex // include/struct.h:9-14
{
public:
    float Real;
    float Image;
};
class LDPCCode // :15-26
{
public:
    int maxWeight_checknode;
    int maxWeight_variablenode;
    int GF;
    int Variablenode_num;
    int Checknode_num;
    float rate;
    int bit_length;
    int q_bit;
};
class VN // :27-37
{
public:
    int *linkCNs;
    int *linkCNs_GF;
    int weight;
    float *LLR;
    float *L_ch;
    float **sort_L_v2c;
    unsigned **sort_Entr_v2c;
};
class CN // :38-45
{
public:
    int *linkVNs;
    int *linkVNs_GF;
    int weight;
    float **L_c2v;
};
class AWGNChannel // :46-51
{
public:
    int seed[3];
    float sigma;
};
class Simulation // :53-71
{
public:
    float SNR;
    double sumTime;
    long num_Frames;
    long num_Error_Frames;
    long num_Error_Bits; // symbol errors (Simulation.cpp:264-268)
    long Total_Iteration;
    long num_False_Frames;
    long num_Alarm_Frames;
    float FER;
    float BER; // symbol errors / frames / Variablenode_num (sic, Simulation.cpp:193,276)
    float AverageIT;
    float FER_False;
    float FER_Alarm;
};

// ---- configuration: the macros of include/define.h:23-59 ------------------------------------------------------------------------
extern "C" {
typedef struct nbldpc_shim_sim_config {
    const char *Matrixfile;        // define.h:23  "BDS.576.288.GF.64.txt"
    const char *Constellationfile; // :24          "./Constellation/BPSK.txt"
    const char *GFTabledir;        // GF.cpp:81    "./GF" (the file is <dir>/Arith.Table.GF.<GFQ>.txt)
    int n_QAM, GFQ, maxdc, maxdv, THREAD_NUM; // :25-29  2, 64, 4, 2, 1 (THREAD_NUM only sizes Get_H's node arrays here)
    int EMS_NM, EMS_NC, maxIT, decoder_method; // :31-37 2, 2, 20, 0
    long leastErrorFrames, leastTestFrames, displayStep; // :52-54  50, 1000, 100000
    // not in the reference:
    int batch;                 // frames decoded per launch inside Simulation_* (default 4096; results do not depend on it)
    int device_channel;        // 0: the reference's host noise stream (bit-identical samples); 1: the same draws generated on the
                               // GPU by LCG jump-ahead (device libm: a sample may differ by an ulp), no host loop, no upload
    const char *results_file;  // NULL (default): no file; "results.txt" to append the rows as the reference does (Simulation.cpp:199-206)
} nbldpc_shim_sim_config;
void nbldpc_shim_sim_defaults(nbldpc_shim_sim_config *cfg);
int nbldpc_shim_configure_sim(const nbldpc_shim_sim_config *cfg); // 0 or an NBLDPC_E* code

// The older, narrower form (decoder entry points only): TableMultiply host unsigned [GFQ][GFQ] (GF.cpp:68-117).
int nbldpc_shim_configure(int GFQ, int maxdv, int maxdc, int maxIT, const unsigned *TableMultiply);
void nbldpc_shim_reset(void); // drop the cached code object (e.g. before switching matrices)

// Test hook: called for every frame Simulation_* accounts, in stream order, before Statistic (frame = 0-based index within the
// current Simulation_* call).
typedef void (*nbldpc_shim_frame_hook)(void *user, long frame, const int *DecodeOutput, int iter_number, int ok);
void nbldpc_shim_set_frame_hook(nbldpc_shim_frame_hook hook, void *user);
}

// ---- include/GF.h:7-19 --------------------------------------------------------------------------------------------------------------
extern unsigned **TableAdd;
extern unsigned **TableMultiply;
extern unsigned *TableInverse;
unsigned **malloc_2(int xDim, int yDim);
float **malloc_2_float(int xDim, int yDim);
int GFAdd(int ele1, int ele2);
int GFMultiply(int ele1, int ele2);
int GFInverse(int ele);
bool GFInitial(int GFq);

// ---- include/Simulation.h:8-20, include/struct.h:73-76 --------------------------------------------------------------------------------
void Get_H(LDPCCode *H, VN *Variablenode, CN *Checknode);
CComplex *Get_CONSTELLATION(LDPCCode *H);
void freeVN(const LDPCCode *H, VN *A);
void freeCN(const LDPCCode *H, CN *A);
void Simulation_CPU(const LDPCCode *H, AWGNChannel *AWGN, Simulation *SIM, const CComplex *CONSTELLATION, VN *Variablenode, CN *Checknode,
                    const CComplex *CComplex_sym, const int *CodeWord_sym);
// The eight device-pointer arguments (tables and flattened graph, main.cu:89-188) are accepted and ignored: the code object
// behind the shim is built from H / Variablenode / Checknode and TableMultiply.  decoder_method 3 prints "unfinished" and exits
// here as it does in the reference (Simulation.cpp:140-144); Simulation_CPU runs it.
void Simulation_GPU(const LDPCCode *H, AWGNChannel *AWGN, Simulation *SIM, const CComplex *CONSTELLATION, VN *Variablenode, CN *Checknode,
                    const CComplex *CComplex_sym, int *CodeWord_sym, const unsigned *TableMultiply_GPU, const unsigned *TableAdd_GPU,
                    const unsigned *TableInverse_GPU, const int *Variablenode_weight, const int *Checknode_weight,
                    const int *Variablenode_linkCNs, const int *Checknode_linkVNs, const int *Checknode_linkVNs_GF);
int Statistic(Simulation *SIM, const int *CodeWord_Frames, int *D, const LDPCCode *LDPC);

// ---- include/LDPC_Encoder.h:7-15 ------------------------------------------------------------------------------------------------------
void BitToSym(LDPCCode *H, int *CodeWord_sym, int *CodeWord_bit);
void Modulate(const LDPCCode *H, CComplex *CONSTELLATION, CComplex *CComplex_sym, int *CodeWord_sym);
void AWGNChannel_CPU(const LDPCCode *H, AWGNChannel *AWGN, CComplex *CComplex_sym_Channelout, const CComplex *CComplex_sym);
float RandomModule(int *seed);

// ---- include/LDPC_Decoder.h:9-25, include/Decode_GPU.cuh:17,19: ONE frame per call ----------------------------------------------------
void Demodulate(const LDPCCode *H, AWGNChannel *AWGN, const CComplex *CONSTELLATION, VN *Variablenode, CComplex *CComplex_sym_Channelout);
int index_in_VN(CN *Checknode, int CNnum, int index_in_linkVNS, VN *Variablenode);
int index_in_CN(VN *Variablenode, int VNnum, int index_in_linkCNS, CN *Checknode);
int Decoding_EMS(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
int Decoding_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
int Decoding_layered_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
// The GPU twins: same results as the functions above (the canonical semantics are the reference's CPU decoders, whose float
// operation order is the defined one); the device-pointer arguments are accepted and ignored.
int Decoding_EMS_GPU(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput,
                     const unsigned *TableMultiply_GPU, const unsigned *TableAdd_GPU, const int *Variablenode_weight, const int *Checknode_weight,
                     const int *Variablenode_linkCNs, const int *Checknode_linkVNs, const int *Checknode_linkVNs_GF, int &iter_number);
int Decoding_TMM_GPU(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput,
                     const unsigned *TableMultiply_GPU, const unsigned *TableAdd_GPU, const unsigned *TableInverse_GPU, const int *Variablenode_weight,
                     const int *Checknode_weight, const int *Variablenode_linkCNs, const int *Checknode_linkVNs, const int *Checknode_linkVNs_GF,
                     int &iter_number);
