/*
 * ref_nb_driver.cpp -- driver around the UNMODIFIED reference NB-LDPC CPU
 * decoder.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * This file is ours; it is linked (by oracle/Makefile) against object files
 * compiled directly from /root/reference/myNBLDPC/src/{LDPC_Decoder,
 * LDPC_Encoder,Simulation,GF,struct}.cpp where they lie.  It mirrors the set-up
 * sequence of the reference's main() (src/main.cu:49-87, 203-223), which is a
 * CUDA translation unit and cannot be compiled here, then calls the reference's
 * own Get_H / GFInitial / Get_CONSTELLATION / Modulate / AWGNChannel_CPU /
 * Demodulate / Decoding_EMS exactly as decode_once_cpu does
 * (src/Simulation.cpp:41-83) and dumps inputs, outputs and the last-iteration
 * message state so the restatement in nbldpc_oracle.c can be compared bit for
 * bit.
 *
 * Must run with the working directory holding the reference's relative data
 * paths (define.h:23-24, GF.cpp:81): BDS.576.288.GF.64.txt,
 * Constellation/BPSK.txt, GF/Arith.Table.GF.64.txt  (here: data/nb/).
 *
 * usage: nb_ref dump  <snr_db> <nframes> <out.bin> [method]
 *        nb_ref time  <snr_db> <nframes> [method]    -> prints frames/s of the decoder
 * method = the reference's decoder_method (define.h:37): 0 Decoding_EMS (default), 1 Decoding_TMM,
 * 3 Decoding_layered_TMM; the trellis decoders keep q entries per vector (element 0 included), the dump follows.
 */
#include "define.h"
#include "LDPC_Decoder.h"
#include "LDPC_Encoder.h"
#include "GF.h"
#include "codeword_test.h"
#include <chrono>

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s dump|time snr_db nframes [out.bin]\n", argv[0]);
        return 2;
    }
    const bool dump = strcmp(argv[1], "dump") == 0;
    const float snr = (float)atof(argv[2]);
    const int nframes = atoi(argv[3]);
    const int method = dump ? (argc > 5 ? atoi(argv[5]) : 0) : (argc > 4 ? atoi(argv[4]) : 0);
    const int NV = (method == 0) ? GFQ - 1 : GFQ; // entries per message vector in the dump
    FILE *out = NULL;
    if (dump) {
        if (argc < 5) return 2;
        out = fopen(argv[4], "wb");
        if (!out) { perror("out"); return 2; }
    }

    LDPCCode *H = (LDPCCode *)malloc(sizeof(LDPCCode));
    AWGNChannel *AWGN = (AWGNChannel *)malloc(sizeof(AWGNChannel));
    FILE *fp = fopen(Matrixfile, "r");
    if (!fp) { fprintf(stderr, "cannot open %s (wrong cwd?)\n", Matrixfile); return 2; }
    if (fscanf(fp, "%d", &H->Variablenode_num) != 1 || fscanf(fp, "%d", &H->Checknode_num) != 1) return 2;
    fclose(fp);
    VN *Variablenode = (VN *)malloc(H->Variablenode_num * sizeof(VN));
    CN *Checknode = (CN *)malloc(H->Checknode_num * sizeof(CN));
    Get_H(H, Variablenode, Checknode);
    GFInitial(GFQ);
    CComplex *CONSTELLATION = Get_CONSTELLATION(H);

    int *CodeWord_bit = (int *)calloc(H->bit_length, sizeof(int));
    int *CodeWord_sym = (int *)calloc(H->Variablenode_num, sizeof(int));
    CComplex *CComplex_sym = (CComplex *)malloc(H->bit_length * sizeof(CComplex));
    for (int i = 0; i < H->Variablenode_num; i++)
        for (int j = 0; j < H->q_bit; j++)
#ifdef NB_REF_ZERO_CW /* codes other than the GF(64) one of codeword_test.h: the all-zero word, a codeword of every linear code */
            CodeWord_bit[i * H->q_bit + j] = 0;
#else
            CodeWord_bit[i * H->q_bit + j] = (CodeWord_sym_test[i] & (1 << j)) >> j;
#endif
    BitToSym(H, CodeWord_sym, CodeWord_bit);
    /* Modulate indexes the constellation with bits for BPSK and with symbols otherwise (LDPC_Encoder.cpp:18-36) */
    Modulate(H, CONSTELLATION, CComplex_sym, (n_QAM != 2) ? CodeWord_sym : CodeWord_bit);

    AWGN->seed[0] = ix_define;
    AWGN->seed[1] = iy_define;
    AWGN->seed[2] = iz_define;
    if (snrtype == 0)
        AWGN->sigma = (float)sqrt(0.5 / (log(n_QAM) / log(2) * H->rate * (pow(10.0, (snr / 10.0)))));
    else
        AWGN->sigma = (float)sqrt(0.5 / (log(n_QAM) / log(2) * pow(10.0, (snr / 10.0))));

    int *DecodeOutput = (int *)calloc(H->Variablenode_num, sizeof(int));
    CComplex *chan = (CComplex *)malloc(H->bit_length * sizeof(CComplex));
    int iter_number = 0;
    double secs = 0;
    long iters = 0;
    if (dump) {
        int hdr[8] = {H->Variablenode_num, H->Checknode_num, GFQ, maxdv, maxdc, maxIT, nframes, method | ((n_QAM != 2 ? n_QAM : 0) << 8)};
        float fh[2] = {AWGN->sigma, H->rate};
        fwrite(hdr, sizeof(int), 8, out);
        fwrite(fh, sizeof(float), 2, out);
        fwrite(CodeWord_sym, sizeof(int), H->Variablenode_num, out);
    }
    for (int fr = 0; fr < nframes; fr++) {
        AWGNChannel_CPU(H, AWGN, chan, (const CComplex *)CComplex_sym);
        Demodulate(H, AWGN, (const CComplex *)CONSTELLATION, Variablenode, chan);
        if (dump) {
            if (n_QAM != 2) /* one complex sample per code symbol (LDPC_Encoder.cpp:45-52) */
                for (int b = 0; b < H->Variablenode_num; b++) { fwrite(&chan[b].Real, sizeof(float), 1, out); fwrite(&chan[b].Image, sizeof(float), 1, out); }
            else
                for (int b = 0; b < H->bit_length; b++) fwrite(&chan[b].Real, sizeof(float), 1, out);
            for (int i = 0; i < H->Variablenode_num; i++) fwrite(Variablenode[i].L_ch, sizeof(float), GFQ - 1, out);
        }
        auto t0 = std::chrono::steady_clock::now();
        int ok = (method == 1)   ? Decoding_TMM(H, Variablenode, Checknode, EMS_NM, EMS_NC, DecodeOutput, iter_number)
                 : (method == 3) ? Decoding_layered_TMM(H, Variablenode, Checknode, EMS_NM, EMS_NC, DecodeOutput, iter_number)
                                 : Decoding_EMS(H, Variablenode, Checknode, EMS_NM, EMS_NC, DecodeOutput, iter_number);
        auto t1 = std::chrono::steady_clock::now();
        secs += std::chrono::duration<double>(t1 - t0).count();
        iters += iter_number;
        if (dump) {
            fwrite(DecodeOutput, sizeof(int), H->Variablenode_num, out);
            fwrite(&iter_number, sizeof(int), 1, out);
            fwrite(&ok, sizeof(int), 1, out);
            for (int i = 0; i < H->Variablenode_num; i++) fwrite(Variablenode[i].LLR, sizeof(float), NV, out);
            for (int r = 0; r < H->Checknode_num; r++)
                for (int d = 0; d < maxdc; d++) {
                    static float zeros[GFQ];
                    fwrite(d < Checknode[r].weight ? Checknode[r].L_c2v[d] : zeros, sizeof(float), NV, out);
                }
        }
    }
    if (dump) fclose(out);
    printf("{\"frames\": %d, \"seconds\": %.6f, \"frames_per_s\": %.4f, \"mean_iters\": %.3f, \"sigma\": %.9g}\n", nframes,
           secs, nframes / secs, (double)iters / nframes, AWGN->sigma);
    return 0;
}
