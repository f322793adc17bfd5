/*
 * bldpc.h -- C ABI of the MI355X-native binary QC-LDPC flooding min-sum decoder.
 *
 * This is the drop-in boundary for the reference's binary hot path
 * (gsw4869/CUDA_LDPC, directory bldpc_实习/).  Every entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++ or
 * torch types.  All functions return BLDPC_OK (0) or a negative error code and
 * never call exit() (the reference printf+exit(0)s on every failure,
 * LDPC_Decoder.cu:39-44); bldpc_last_error() returns a message for the calling
 * thread.
 *
 * Data layouts are the reference's:
 *   Channel_Out  float  [N][F]   frame-fastest, device        (LDPC_Decoder.cu:23, Simulation.cu:74)
 *   D            int32  [N+1][F] frame-fastest; row N = per-frame flag
 *                                "first `length` decoded bits are all zero"    (LDPC_Decoder.cu:134-147)
 *   Address_Variablenode int32 [N][Wv]: Memory_RQ slot (m*Wc+p) of each edge   (Simulation.cu:363-387)
 *   Weight_Checknode [J+1], Weight_Variablenode [L+1]: block weights, last = max (Simulation.cu:321-340)
 * with N = L*Z, M = J*Z, Wc/Wv the maximum block-row / block-column weights.
 * J, L, Z, the batch size F, maxIT and msgLen are compile-time macros in the
 * reference (define.cuh:20-61); here they are run-time arguments.
 */
#ifndef CUDA_LDPC_AMD_BLDPC_H
#define CUDA_LDPC_AMD_BLDPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define BLDPC_OK 0
#define BLDPC_EINVAL (-1)       /* bad argument / inconsistent shapes            */
#define BLDPC_ENOMEM (-2)       /* host or device allocation failed              */
#define BLDPC_EHIP (-3)         /* a HIP runtime call failed                     */
#define BLDPC_EIO (-4)          /* file could not be opened / parsed             */
#define BLDPC_EUNSUPPORTED (-5) /* requested kernel cannot run this code         */

/* early-exit modes of bldpc_decode */
#define BLDPC_EXIT_FIXED 0        /* run exactly max_iter iterations (benchmark mode)                       */
#define BLDPC_EXIT_BATCH_GLOBAL 1 /* reference rule: stop after the first iteration at which ALL F frames   */
                                  /* have their flag set (LDPC_Decoder.cu:150-153)                          */
#define BLDPC_EXIT_PER_FRAME 2    /* the same rule applied to every frame on its own (what the reference does with  */
                                  /* Num_Frames_OneTime = 1): bldpc_decode_per_frame only                           */

/* kernel selection */
#define BLDPC_KERNEL_AUTO 0   /* QC_LDS when the code has QC structure and fits LDS, else TABLE */
#define BLDPC_KERNEL_TABLE 1  /* generic address-table kernels, messages resident in HBM; any table      */
#define BLDPC_KERNEL_QC_LDS 2 /* fused QC kernels, all iterations on-chip: messages in LDS, or (larger codes)  *
                               * compressed check states in LDS / in registers with the a-posteriori values in LDS */

typedef struct bldpc_code bldpc_code; /* opaque: device-resident code tables + cached workspace.  A code object owns
                                        * its scratch buffers: use it from one host thread / one stream at a time (create
                                        * one object per concurrent stream, like a BLAS handle); different objects are
                                        * independent. */

/* -- graph builders ------------------------------------------------------- */

/* Replaces Get_H (Simulation.cu:292-354) without its hard-coded file name and
 * weight-range exits: reads J*L shifts (-1 = zero block) and fills the block
 * weights. H[J*L], Weight_Checknode[J+1], Weight_Variablenode[L+1]: host. */
int bldpc_read_blockh(const char *path, int J, int L, int *H, int *Weight_Checknode, int *Weight_Variablenode);

/* Replaces Transform_H (Simulation.cu:363-387). as_written = 0 builds the
 * intended circulant (column c of a block with shift s meets row (c-s) mod Z);
 * as_written = 1 reproduces the reference expression at :380 literally (its
 * else-branch maps columns c < s to row c).
 * Address_Variablenode[N*Wv]: host, unused entries -1 (main.cu:94). */
int bldpc_transform_h(const int *H, int J, int L, int Z, const int *Weight_Checknode, const int *Weight_Variablenode,
                      int *Address_Variablenode, int as_written);

/* -- code objects --------------------------------------------------------- */

/* Build a decoder for the QC code with block shifts H[J*L] (host).  Uploads the
 * circulant description for the fused LDS kernel AND the equivalent (correct)
 * address table for the generic kernels. */
int bldpc_code_create_qc(int J, int L, int Z, const int *H, bldpc_code **code);

/* Build a decoder from an arbitrary host address table with the reference's
 * semantics (what main.cu:98 uploads), including the as-written table whose
 * colliding slots make the reference's own output order-dependent; here the
 * order is defined as ascending variable-node index within each iteration
 * (the order of a sequential emulation of the reference launch), enforced by
 * level-scheduled launches.  Only BLDPC_KERNEL_TABLE can run such a code. */
int bldpc_code_create_table(int J, int L, int Z, const int *Weight_Checknode, const int *Weight_Variablenode,
                            const int *Address_Variablenode, bldpc_code **code);

int bldpc_code_destroy(bldpc_code *code);

/* dims[0..7] = N, M, K (= N-M), Wc, Wv, nnz blocks, VN levels (1 = conflict-free table),
 *              frames per workgroup of the QC_LDS kernel (0 = code does not fit LDS / no QC structure) */
int bldpc_code_dims(const bldpc_code *code, int dims[8]);

/* -- decode ---------------------------------------------------------------- */

/* Replaces LDPC_Decoder_GPU (LDPC_Decoder.cuh:5, LDPC_Decoder.cu:23-164).
 *   Channel_Out   device float [N][F]                         (in)
 *   F             frames in this call (reference: Num_Frames_OneTime)
 *   max_iter      reference: maxIT (define.cuh:35)
 *   length        bits examined by the termination test; reference: msgLen
 *                 (Message_CW == 0) or CW_Len; pass 0 for K = N - M
 *   exit_mode     BLDPC_EXIT_*
 *   kernel        BLDPC_KERNEL_*
 *   D             device int32 [N+1][F]                       (out) hard bits of the last executed
 *                 iteration + flag row, as the reference leaves them in its host D
 *   app           optional device float [N][F]                (out) a-posteriori sums (Add_result,
 *                 LDPC_Decoder.cu:201-204) of the last executed iteration; NULL to skip
 *   flag_hist     optional device uint64 [F]                  (out) bit (it-1) = flag of frame after
 *                 iteration it, it <= 64; NULL to skip
 *   iteraTime     host int                                    (out) iterations executed
 *                 (LDPC->iteraTime, batch-global like the reference)
 *   stream        hipStream_t (NULL = default stream).  The call is asynchronous in
 *                 BLDPC_EXIT_FIXED mode; BATCH_GLOBAL synchronises the stream (it reads flags).
 * No per-call allocation: scratch lives in the code object and grows on demand. */
int bldpc_decode(bldpc_code *code, const float *Channel_Out, int F, int max_iter, int length, int exit_mode, int kernel,
                 int *D, float *app, unsigned long long *flag_hist, int *iteraTime, void *stream);

/* Per-frame termination: frame f stops after the first iteration at which ITS flag is set -- the
 * reference's rule (LDPC_Decoder.cu:134-153) as it acts on a batch of one frame -- and column f of D (and of app) holds
 * the outputs of that iteration, exactly what LDPC_Decoder_GPU returns for that frame with Num_Frames_OneTime = 1;
 * a frame whose flag never comes up runs max_iter iterations.  Nothing is recomputed and nothing waits for the slowest
 * frame of the batch: on the fused kernels a workgroup leaves when its own (1-2) frames have stopped, so the cost of a
 * batch follows the MEAN iteration count (BER sweeps at operating SNR: several times the fixed-iteration rate).
 *   iters   device int32 [F]  (out) iterations executed by each frame (the reference's iteraTime of that frame)
 * Other arguments as bldpc_decode.  Asynchronous on `stream` with the fused kernels; the table kernels read one
 * counter per iteration, as for BATCH_GLOBAL. */
int bldpc_decode_per_frame(bldpc_code *code, const float *Channel_Out, int F, int max_iter, int length, int kernel, int *D,
                           float *app, int *iters, void *stream);

/* Device-side Statistic (Simulation.cu:245-262) over one decoded batch against
 * the all-zero codeword (PN_Message 0, define.cuh:26) or CodeWord (device int32
 * [N][F], may be NULL = all-zero).  counters: device int64[5], ACCUMULATED:
 *   [0] num_Error_Frames [1] num_Error_Bits [2] Total_Iteration (+= iteraTime per frame)
 *   [3] num_False_Frames [4] num_Alarm_Frames.   num_Frames is the caller's (+= F). */
int bldpc_statistic(const bldpc_code *code, const int *D, const int *CodeWord, int F, int length, int iteraTime,
                    long long *counters, void *stream);

/* The same with one iteration count per frame (bldpc_decode_per_frame): Total_Iteration += iters[f]. */
int bldpc_statistic_per_frame(const bldpc_code *code, const int *D, const int *CodeWord, int F, int length, const int *iters,
                              long long *counters, void *stream);

/* bldpc_decode (iters == NULL; FIXED or BATCH_GLOBAL exit) or bldpc_decode_per_frame (iters != NULL, exit_mode PER_FRAME)
 * followed by bldpc_statistic / bldpc_statistic_per_frame against the all-zero codeword, the pair of calls of the
 * reference's Simulation_GPU loop (Simulation.cu:143-145): D, *iteraTime / iters and the accumulated counters are the same
 * as from the two calls.  With the fused kernels and a single launch (FIXED, PER_FRAME) the message-bit errors are counted
 * from the packed hard bits while D is written, which saves the statistics pass over D. */
int bldpc_decode_statistic(bldpc_code *code, const float *Channel_Out, int F, int max_iter, int length, int exit_mode, int kernel,
                           int *D, int *iters, long long *counters, int *iteraTime, void *stream);

/* Host input generator, bit-identical to the reference's (the "identical AWGN inputs" of the parity
 * contract): AWGNChannel_CPU + RandomModule (LDPC_Encoder.cu:25-56).  seed[3] is advanced in place
 * (AWGNChannel.seed, struct.cuh:13), sigma as main.cu:120-127 computes it (bldpc_sigma below).
 * Channel_Out: HOST float [N][F], frame-outer / bit-inner draw order; CodeWord: host int32 [N][F] or
 * NULL for the all-zero codeword. */
int bldpc_awgn_channel_host(int seed[3], float sigma, float *Channel_Out, const int *CodeWord, int N, int F);

/* Device-side input generator: the same channel on the GPU.  The three LCGs of RandomModule are
 * advanced by modular exponentiation (seed * a^k mod m), so thread (f, n) produces exactly the draws u1, u2 the
 * serial host loop would have produced for that sample (integer and IEEE float arithmetic only: bit-identical);
 * the Box-Muller transform then uses the device math library, whose logf / sin differ from glibc's in the last
 * ulp on a small fraction of arguments -- samples are identical or 1 ulp apart, statistically the same channel.
 * Channel_Out: DEVICE float [N][F]; CodeWord: device int32 [N][F] or NULL.  seed[3] (host) is advanced by the
 * whole batch exactly like bldpc_awgn_channel_host. */
int bldpc_awgn_channel_device(int seed[3], float sigma, float *Channel_Out, const int *CodeWord, int N, int F, void *stream);

/* sigma of the sweep point (main.cu:120-127): snrtype 0 = Eb/N0 (uses rate), 1 = Es/N0. */
float bldpc_sigma(float SNR, int snrtype, float rate);

/* Timing of the dominant kernel (k_qc / k_qc2, or the VN+CN launch sequence of the table kernels) with
 * HIP events on the stream the kernel runs on.  enable != 0 makes every following bldpc_decode record a
 * pair of events around that kernel; bldpc_last_kernel_ms waits for the last pair and returns the
 * elapsed milliseconds in *ms.  Used by bench.py for the roofline line; off by default. */
int bldpc_set_profiling(bldpc_code *code, int enable);
int bldpc_last_kernel_ms(bldpc_code *code, float *ms);
/* Mean over the decode calls made since the previous bldpc_kernel_ms_mean / bldpc_set_profiling (at most the last 64): every
 * profiled call records its own event pair, nothing synchronises until this function is called.  *launches = calls averaged. */
int bldpc_kernel_ms_mean(bldpc_code *code, float *mean_ms, int *launches);

/* Name of the kernel variant the last bldpc_decode on this code used (static string). */
const char *bldpc_last_kernel(const bldpc_code *code);

const char *bldpc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
