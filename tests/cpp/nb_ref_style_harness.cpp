// A harness written the way the reference's decode_once_cpu drives its decoders (myNBLDPC/src/Simulation.cpp:16-87):
// pointer-rich VN[] / CN[] node arrays as Get_H allocates them (:404-431), per frame L_ch written into the nodes, then
// Decoding_EMS / Decoding_TMM / Decoding_layered_TMM(H, Variablenode, Checknode, EMS_NM, EMS_NC, DecodeOutput, iter_number)
// with the reference's signatures (shim/nbldpc_ref_shim.hpp).  Prints, per frame, the return flag, iter_number and the
// fold hashes of DecodeOutput, of VN[].LLR and of CN[].L_c2v as the function leaves them.
// usage: nb_ref_style_harness <matrix.txt> <gf-table.txt> <Lch.bin: float [frames][N][q-1]> <frames> <method 0|1|3>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nbldpc.h"
#include "nbldpc_ref_shim.hpp"

static unsigned fold(unsigned h, const void *p, size_t words)
{
    const unsigned *u = (const unsigned *)p;
    for (size_t i = 0; i < words; i++) h = (h ^ u[i]) * 16777619u;
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int frames = atoi(argv[4]), method = atoi(argv[5]);
    int dims[5];
    if (nbldpc_read_matrix(argv[1], dims, 0, 0, 0, 0, 0, 0)) { printf("%s\n", nbldpc_last_error()); return 1; }
    const int N = dims[0], M = dims[1], q = dims[2], dv = dims[3], dc = dims[4], nv = method == 0 ? q - 1 : q;
    std::vector<int> vw(N), vc(N * dv), vg(N * dv), cw(M), cv(M * dc), cg(M * dc);
    if (nbldpc_read_matrix(argv[1], dims, vw.data(), vc.data(), vg.data(), cw.data(), cv.data(), cg.data())) return 1;
    std::vector<unsigned> mul(q * q), add(q * q), inv(q);
    if (nbldpc_gf_load(argv[2], q, mul.data(), add.data(), inv.data())) { printf("%s\n", nbldpc_last_error()); return 1; }

    LDPCCode H;
    H.Variablenode_num = N; H.Checknode_num = M; H.GF = q; H.maxWeight_variablenode = dv; H.maxWeight_checknode = dc;
    VN *Variablenode = (VN *)malloc(N * sizeof(VN));
    CN *Checknode = (CN *)malloc(M * sizeof(CN));
    for (int i = 0; i < N; i++) { // Simulation.cpp:404-418
        VN &v = Variablenode[i];
        v.weight = vw[i];
        v.linkCNs = (int *)malloc(v.weight * sizeof(int));
        v.linkCNs_GF = (int *)malloc(v.weight * sizeof(int));
        for (int d = 0; d < v.weight; d++) { v.linkCNs[d] = vc[i * dv + d]; v.linkCNs_GF[d] = vg[i * dv + d]; }
        v.L_ch = (float *)malloc(q * sizeof(float));
        v.LLR = (float *)malloc(q * sizeof(float));
        v.sort_L_v2c = 0; v.sort_Entr_v2c = 0; // message scratch of the CPU decoder: not part of what the callers read
    }
    for (int r = 0; r < M; r++) { // Simulation.cpp:420-431
        CN &c = Checknode[r];
        c.weight = cw[r];
        c.linkVNs = (int *)malloc(c.weight * sizeof(int));
        c.linkVNs_GF = (int *)malloc(c.weight * sizeof(int));
        c.L_c2v = (float **)malloc(c.weight * sizeof(float *));
        for (int d = 0; d < c.weight; d++) { c.linkVNs[d] = cv[r * dc + d]; c.linkVNs_GF[d] = cg[r * dc + d]; c.L_c2v[d] = (float *)calloc(q, sizeof(float)); }
    }
    if (nbldpc_shim_configure(q, dv, dc, 20, mul.data())) return 1;

    FILE *fp = fopen(argv[3], "rb");
    if (!fp) return 1;
    std::vector<int> DecodeOutput(N);
    for (int fr = 0; fr < frames; fr++) {
        for (int i = 0; i < N; i++)
            if (fread(Variablenode[i].L_ch, sizeof(float), q - 1, fp) != (size_t)(q - 1)) return 1;
        int iter_number = 0, ok;
        if (method == 1) ok = Decoding_TMM(&H, Variablenode, Checknode, 2, 2, DecodeOutput.data(), iter_number);
        else if (method == 3) ok = Decoding_layered_TMM(&H, Variablenode, Checknode, 2, 2, DecodeOutput.data(), iter_number);
        else ok = Decoding_EMS(&H, Variablenode, Checknode, 2, 2, DecodeOutput.data(), iter_number);
        unsigned ho = fold(2166136261u, DecodeOutput.data(), N), hl = 2166136261u, hc = 2166136261u;
        for (int i = 0; i < N; i++) hl = fold(hl, Variablenode[i].LLR, nv);
        static float zeros[4096];
        for (int r = 0; r < M; r++)
            for (int d = 0; d < dc; d++) hc = fold(hc, d < Checknode[r].weight ? Checknode[r].L_c2v[d] : zeros, nv);
        printf("frame %d ok=%d it=%d out=%08x LLR=%08x c2v=%08x\n", fr, ok, iter_number, ho, hl, hc);
    }
    fclose(fp);
    return 0;
}
