/*
 * nbldpc.h -- C ABI of the MI355X-native GF(q) non-binary LDPC EMS decoder.
 *
 * Drop-in boundary for the reference's NB hot path (gsw4869/CUDA_LDPC, directory
 * myNBLDPC/): Decoding_EMS (include/LDPC_Decoder.h:13, src/LDPC_Decoder.cpp:172-317)
 * and its GPU twin Decoding_EMS_GPU (include/Decode_GPU.cuh:17).  Canonical
 * semantics = the reference's CPU decoder (the only path of the reference whose float
 * operation order is defined); results are bit-identical to it.
 *
 * The reference decodes ONE frame per call out of pointer-rich VN/CN objects; here a
 * call decodes a batch of B independent frames from flat arrays:
 *   L_ch          float [B][N][q-1]   element k-1 <-> field element k (element 0 has LLR 0),
 *                                     what Demodulate leaves in VN[i].L_ch (LDPC_Decoder.cpp:139-157)
 *   DecodeOutput  int32 [B][N]        hard symbols (Decoding_EMS argument of the same name)
 *   iter_number   int32 [B]           as the reference leaves it: decremented once on success
 *                                     (LDPC_Decoder.cpp:232-238), == maxIT on failure
 *   ok            int32 [B]           the reference's return value (1 = zero syndrome reached)
 * Graph arrays flatten VN[]/CN[] (include/struct.h:27-45), -1 padded:
 *   vn_weight[N], vn_linkCNs[N][dvmax], vn_linkCNs_GF[N][dvmax]
 *   cn_weight[M], cn_linkVNs[M][dcmax], cn_linkVNs_GF[M][dcmax]
 * All functions return NBLDPC_OK or a negative code; nbldpc_last_error() has the text.
 */
#ifndef CUDA_LDPC_AMD_NBLDPC_H
#define CUDA_LDPC_AMD_NBLDPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define NBLDPC_OK 0
#define NBLDPC_EINVAL (-1)
#define NBLDPC_ENOMEM (-2)
#define NBLDPC_EHIP (-3)
#define NBLDPC_EIO (-4)
#define NBLDPC_EUNSUPPORTED (-5)

typedef struct nbldpc_code nbldpc_code;

/* Replaces Get_H (src/Simulation.cpp:347-467): "N M q / dvmax dcmax / N column weights / M row weights /
 * per VN (cn_1based, gf)*w / per CN (vn_1based, gf)*w".  Call with all arrays NULL to obtain
 * dims[0..4] = N, M, q, dvmax, dcmax, then again with arrays sized from them (host memory). */
int nbldpc_read_matrix(const char *path, int dims[5], int *vn_weight, int *vn_linkCNs, int *vn_linkCNs_GF, int *cn_weight,
                       int *cn_linkVNs, int *cn_linkVNs_GF);

/* Replaces GFInitial (src/GF.cpp:68-117): reads "Multiply Table" q*q, "Add Table" q*q, "Inverse Table" q. */
int nbldpc_gf_load(const char *path, int q, unsigned *TableMultiply, unsigned *TableAdd, unsigned *TableInverse);

/* The same tables computed from the primitive polynomial (e.g. 67 = x^6+x+1 for GF(64), first line of
 * GF/Arith.Table.GF.64.txt); polynomial-basis representation, add == XOR. */
int nbldpc_gf_generate(int q, unsigned primitive_poly, unsigned *TableMultiply, unsigned *TableAdd, unsigned *TableInverse);

/* Upload a code.  TableMultiply: host unsigned [q][q].  q a power of two, 4 <= q <= 256; column weights <= 8; row weights 2..64.
 * The fused kernels (one frame's message state inside one CU's LDS) take q in {16, 32, 64} (a message vector inside one wave, EMS
 * and trellis decoders) and q in {128, 256} (a vector over q/64 waves, EMS only) with row weights up to 6 (e.g.
 * BDS.576.288.GF.64.txt, LDPC_N96_K48_GF256_d1_exp.txt); every other code (Tanner_74_9_Z128_GF16.txt: row weight 21, 7.5 MB of
 * state; LDPC_N576_K480_GF256_exp.txt: row weight 12) is decoded by the workspace kernel (EMS only, same results, far slower).
 * nbldpc_tmm_decode_batch returns NBLDPC_EUNSUPPORTED where the trellis kernels do not apply.  A zero edge coefficient is
 * accepted for EMS (the reference reads its exponent-format files as field elements and decodes with the zeros in place); the
 * trellis decoders refuse such a code (the reference's GFInverse(0) exits). */
int nbldpc_code_create(int N, int M, int q, int dvmax, int dcmax, const int *vn_weight, const int *vn_linkCNs,
                       const int *vn_linkCNs_GF, const int *cn_weight, const int *cn_linkVNs, const int *cn_linkVNs_GF,
                       const unsigned *TableMultiply, nbldpc_code **code);
int nbldpc_code_destroy(nbldpc_code *code);

/* Name of the kernel the last decode call on this code launched (static string): "k_nb_ems2" = two frames in flight per workgroup,
 * "k_nb_ems" / "k_nb_ems_wide" one frame per workgroup (+ " (frames from a counter)": persistent workgroups), "k_nb_ems_hbm" the
 * workspace kernel, "k_nb_tmm…" the trellis decoders.  The counterpart of bldpc_last_kernel (include/bldpc.h); the reference has no
 * equivalent -- its kernels are fixed at compile time (myNBLDPC/src/Decode_GPU.cu:138-356). */
const char *nbldpc_last_kernel(const nbldpc_code *code);

/* Replaces Decoding_EMS / Decoding_EMS_GPU for a batch (all pointers DEVICE memory).
 *   EMS_Nm, EMS_Nc   reference macros EMS_NM / EMS_NC (define.h:31-32)
 *   maxIT            reference macro maxIT (define.h:35)
 *   maxdc_cfg        reference macro maxdc (define.h:28), used by the `EMS_Nc == maxdc - 1` test
 *                    (LDPC_Decoder.cpp:294); pass 0 for the code's dcmax
 *   LLR     optional float [B][N][q-1]       VN[].LLR of the last executed iteration
 *   L_c2v   optional float [B][M][dcmax][q-1] CN[].L_c2v as the reference leaves it
 *   stream  hipStream_t; the call is asynchronous. */
int nbldpc_ems_decode_batch(nbldpc_code *code, const float *L_ch, int B, int EMS_Nm, int EMS_Nc, int maxIT, int maxdc_cfg,
                            int *DecodeOutput, int *iter_number, int *ok, float *LLR, float *L_c2v, void *stream);

/* Trellis min-max decoders for a batch of frames: Decoding_TMM (decoder_method 1, myNBLDPC/src/LDPC_Decoder.cpp:361-558,
 * flooding) and Decoding_layered_TMM (decoder_method 3, :560-702; layered != 0), called from decode_once_cpu/gpu
 * (src/Simulation.cpp:58-70, :132-143) with the same arguments as Decoding_EMS.  Same outputs and return convention as
 * nbldpc_ems_decode_batch (ok = 1: zero syndrome reached, iter_number already decremented).  Optional state outputs hold
 * the reference's VN[].LLR and CN[].L_c2v after the call with ALL q entries per vector (element 0 included):
 * LLR DEVICE float [B][N][q], L_c2v DEVICE float [B][M][dcmax][q]; NULL to skip.  The layered schedule visits the
 * rows in dependency levels (rows that share no variable node commute), which leaves every value and its order of
 * computation per variable as in the reference's row-by-row loop. */
int nbldpc_tmm_decode_batch(nbldpc_code *code, const float *L_ch, int B, int layered, int maxIT, int *DecodeOutput, int *iter_number,
                            int *ok, float *LLR, float *L_c2v, void *stream);

/* Replaces Demodulate, BPSK branch (src/LDPC_Decoder.cpp:132-157), on the device:
 * rx float [B][N*m] (m = log2 q, bit b of symbol s at s*m+b) -> L_ch float [B][N][q-1]. */
int nbldpc_demodulate_bpsk(const nbldpc_code *code, const float *rx, float sigma, int B, float *L_ch, void *stream);

/* The same two without a code object (Demodulate needs the dimensions only): N symbols of GF(q). */
int nbldpc_demodulate_bpsk_nq(int N, int q, const float *rx, float sigma, int B, float *L_ch, void *stream);
int nbldpc_demodulate_qam_nq(int N, int q, const float *rx, const float *constellation, float sigma, int B, float *L_ch, void *stream);

/* Device-side Statistic (src/Simulation.cpp:256-279) over a decoded batch: counters device int64[4],
 * ACCUMULATED: [0] num_Error_Frames [1] num_Error_Bits (symbol errors, sic) [2] Total_Iteration [3] frames ok.
 * CodeWord_sym: device int32 [N]. */
int nbldpc_statistic(const nbldpc_code *code, const int *DecodeOutput, const int *iter_number, const int *ok,
                     const int *CodeWord_sym, int B, long long *counters, void *stream);

/* Host input generator, bit-identical to the reference (BPSK, n_QAM 2): Modulate of the codeword bits
 * (src/main.cu:203-211, Constellation/BPSK.txt: bit 0 -> +1, bit 1 -> -1) + AWGNChannel_CPU
 * (src/LDPC_Encoder.cpp:41-68: four RandomModule draws per bit, cos branch).  rx: HOST float [N*m] for
 * ONE frame; seed[3] advanced in place. */
int nbldpc_awgn_channel_host(int seed[3], float sigma, const int *CodeWord_sym, int N, int m, float *rx);

/* Device-side input generator: the same stream for B consecutive frames.  Every (frame, bit) jumps the
 * three LCGs of RandomModule (src/LDPC_Encoder.cpp:70-79) ahead to its own first draw (draw 4*(b*N*m + i):
 * seed * a^k mod m), so the uniforms are the reference's exactly; the samples are then formed with the device's
 * logf / sqrtf / cos, which may differ from the host libm by an ulp on a small fraction of arguments.
 * CodeWord_sym: DEVICE int32 [N]; rx: DEVICE float [B][N*m]; seed[3] (host) is advanced by B whole frames exactly
 * like B calls of nbldpc_awgn_channel_host. */
int nbldpc_awgn_channel_device(int seed[3], float sigma, const int *CodeWord_sym, int N, int m, int B, float *rx, void *stream);

/* ---- QAM constellations (n_QAM != 2 branches; n_QAM = q: one constellation point per code symbol).  The reference's define.h:25
 * fixes n_QAM 2; these entry points are checked against the reference BUILT with n_QAM 64 and Constellation/GRAY_64QAM.txt
 * (oracle/_ref/nb_ref_qam64, fixtures tests/golden/nb_ref_qam64_*.npz): channel samples, L_ch and decode results bit for bit. ---- */

/* Replaces Get_CONSTELLATION (src/Simulation.cpp:313-338): n_points records "Point: <idx> Real: <x> Imag: <y>" ->
 * constellation HOST float [n_points][2] (Real, Image), indexed by the record's own idx. */
int nbldpc_read_constellation(const char *path, int n_points, float *constellation);

/* Modulate (src/LDPC_Encoder.cpp:18-28: symbol s -> CONSTELLATION[CodeWord_sym[s]]) + AWGNChannel_CPU with
 * len = Variablenode_num (:41-68: two draws for the Real part, two for the Image part, cos branch).
 * rx: HOST float [N][2] for ONE frame; seed[3] advanced in place. */
int nbldpc_awgn_channel_host_qam(int seed[3], float sigma, const int *CodeWord_sym, int N, const float *constellation, int n_points,
                                 float *rx);

/* The same for B consecutive frames on the device (LCG jump-ahead per symbol, device libm as nbldpc_awgn_channel_device).
 * CodeWord_sym: DEVICE int32 [N]; constellation: DEVICE float [q][2]; rx: DEVICE float [B][N][2]. */
int nbldpc_awgn_channel_device_qam(int seed[3], float sigma, const int *CodeWord_sym, int N, const float *constellation, int B, float *rx,
                                   void *stream);

/* Replaces Demodulate, n_QAM != 2 branch (src/LDPC_Decoder.cpp:160-169), on the device:
 * rx float [B][N][2], constellation DEVICE float [q][2] -> L_ch float [B][N][q-1]. */
int nbldpc_demodulate_qam(const nbldpc_code *code, const float *rx, const float *constellation, float sigma, int B, float *L_ch,
                          void *stream);

/* ---- AWGNChannel_CPU exactly as the reference declares it (src/LDPC_Encoder.cpp:41-68): noise added to a MODULATED frame,
 * whatever produced it -- len samples (bit_length for BPSK, Variablenode_num for n_QAM != 2), each a (Real, Image) pair with
 * two RandomModule draws per part.  tx / rx: HOST float [len][2], i.e. the memory of a CComplex array (include/struct.h:9-14).
 * nbldpc_awgn_channel_host / _host_qam above are this function composed with Modulate. */
int nbldpc_awgn_channel_host_sym(int seed[3], float sigma, const float *tx, int len, float *rx);

/* The same for B consecutive frames on the device (LCG jump-ahead per run of samples, device libm).  tx: DEVICE float [len][2];
 * rx: DEVICE float [B][len] holding the Real parts only when real_only != 0 (what nbldpc_demodulate_bpsk takes), else
 * DEVICE float [B][len][2] (nbldpc_demodulate_qam).  seed[3] (host) is advanced by B whole frames. */
int nbldpc_awgn_channel_device_sym(int seed[3], float sigma, const float *tx, int len, int B, int real_only, float *rx, void *stream);

/* RandomModule (src/LDPC_Encoder.cpp:70-79): one uniform draw, seed[3] advanced in place. */
float nbldpc_random_module(int seed[3]);

/* Advance the three LCGs of RandomModule (src/LDPC_Encoder.cpp:70-79) by `draws` calls: seed_i * a_i^draws mod m_i.  One frame
 * of AWGNChannel_CPU is 4 * len draws. */
int nbldpc_seed_jump(int seed[3], unsigned long long draws);

/* Per-frame symbol errors against the transmitted word (the count Statistic makes, src/Simulation.cpp:264-267):
 * DecodeOutput DEVICE int32 [B][N], CodeWord_sym DEVICE int32 [N] -> errs DEVICE int32 [B]. */
int nbldpc_frame_errors(const nbldpc_code *code, const int *DecodeOutput, const int *CodeWord_sym, int B, int *errs, void *stream);

/* sigma of a sweep point (src/main.cu:221-228). */
float nbldpc_sigma(float SNR, int snrtype, int n_QAM, float rate);

const char *nbldpc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
