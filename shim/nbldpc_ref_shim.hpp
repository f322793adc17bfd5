// nbldpc_ref_shim.hpp -- the non-binary reference's own decoder entry points, same C++ signatures, on top of the C ABI.
//
// gsw4869/CUDA_LDPC's NB harness (myNBLDPC/src/Simulation.cpp:54-70, :128-143) calls, one frame at a time,
//     int Decoding_EMS        (const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
//     int Decoding_TMM        (... same arguments ...);                                       (include/LDPC_Decoder.h:13,23,25)
//     int Decoding_layered_TMM(... same arguments ...);
// on pointer-rich node arrays.  The classes below restate the members the decoders touch, in the reference's order
// (include/struct.h:15-45), so that a harness built against this header lays its nodes out the same way; the macros
// of define.h (GFQ, maxdv, maxdc, maxIT) and the GF multiplication table become one run-time call before the first decode:
//     nbldpc_shim_configure(GFQ, maxdv, maxdc, maxIT, TableMultiply);
// The functions gather VN[].L_ch, decode a batch of one through nbldpc_ems_decode_batch / nbldpc_tmm_decode_batch and
// scatter DecodeOutput, iter_number and the last-iteration state the reference leaves behind (VN[].LLR, CN[].L_c2v).
// Failures print the message and exit(0), which is what the reference does on every error path.
#pragma once

class LDPCCode
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
class VN
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
class CN
{
public:
    int *linkVNs;
    int *linkVNs_GF;
    int weight;
    float **L_c2v;
};

// TableMultiply: host unsigned [GFQ][GFQ] (GF.cpp:68-117).  Returns 0 or an NBLDPC_E* code.
extern "C" int nbldpc_shim_configure(int GFQ, int maxdv, int maxdc, int maxIT, const unsigned *TableMultiply);
extern "C" void nbldpc_shim_reset(void); // drop the cached code object (e.g. before switching matrices)

int Decoding_EMS(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
int Decoding_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
int Decoding_layered_TMM(const LDPCCode *H, VN *Variablenode, CN *Checknode, int EMS_Nm, int EMS_Nc, int *DecodeOutput, int &iter_number);
