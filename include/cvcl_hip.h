/* libcvcl_hip.so -- C ABI of the MI355X-native CVCL contrastive hot path.
 *
 * Drop-in boundary for the path SURVEY.md section 8 names.  The reference has no FFI: every
 * entry below replaces an *implicit ATen/cuDNN call* that the reference reaches through
 * torch.nn at the cited line (paths relative to the reference repo root), so a maintainer binds
 * it with ctypes from the same Python call site (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is CALLER-OWNED DEVICE memory,
 *    contiguous in the documented layout; the library never allocates, never synchronises and
 *    only enqueues work on the passed hipStream_t (passed as void*; NULL = default stream).
 *  - return 0 on success, negative CVCL_E* otherwise; cvcl_last_error() gives the thread-local text.
 *  - scratch comes from the caller: cvcl_*_workspace_bytes() sizes it.
 *  - ONE DEVICE PER PROCESS (the launch model of the path: one rank per GPU).  Occupancy queries, CU counts and the raised
 *    dynamic-LDS attributes are cached per process on first use, for the device that is current then; a process that
 *    switches devices afterwards gets the first device's grid and statistics-row counts and unraised LDS limits there.
 *  - dtype: CVCL_F32 = fp32 storage + exact-fp32 MFMA (parity mode, the reference numerics);
 *           CVCL_BF16 = bf16 storage + bf16 MFMA with fp32 accumulation/statistics (perf mode).
 *  - image activations inside the library are NHWC ("channels last"); the API takes the
 *    reference's NCHW fp32 images (multimodal_data_module.py:98-109) and hands back the layer4
 *    map in NHWC memory, which the host exposes as a logical NCHW tensor view.
 *
 * Run-time switches -- the ONLY environment variables the product library reads (once per process, csrc/api.cpp):
 *    variable               default  "0" means
 *    CVCL_CENTRED_STORAGE   on       plain (un-centred) bf16 storage of the raw convolution outputs (the round-2 numerics)
 *    CVCL_GEMM8W            on       cvcl_gemm never selects the 8-wave 256 x 256 kernel (everything on the 128 x 128 kernels)
 *    CVCL_GEMM_PRO          on       cvcl_gemm refuses the BN-prologue kernel (callers normalise the operand themselves)
 *    CVCL_F32_TILED         on       fp32 parity mode: the direct (one thread per output) stem / grouped-conv kernels
 *    CVCL_FINALIZE_ON_LOAD  on       bf16 train-mode trunk: partial rows + a cvcl_bn_finalize launch behind EVERY convolution (the launch
 *                                    sequence of rounds 1-5) instead of accumulated statistics that the consumer of the raw tensor turns
 *                                    into its channels' affine itself ("BatchNorm accumulators" below; tests/test_finalize_on_load_gpu.py)
 *  Experiment switches of earlier rounds (CVCL_FUSED_TAIL_STAGES, CVCL_CONV3_PRO_STAGES, CVCL_DS_RECOMPUTE,
 *  CVCL_PRO_DEPTH, CVCL_GCONV_LDS_KB, CVCL_GEMM_MINW, CVCL_GEMM_GLDS, CVCL_GCONV_WGRAD_BAND) exist only in a library built with
 *  -DCVCL_LAB (tools/README.md); the kernel-variant switches of the 8-wave GEMM live in tools/gemm_lab/.
 */
#ifndef CVCL_HIP_H
#define CVCL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVCL_ABI_VERSION 5

enum { CVCL_OK = 0, CVCL_EINVAL = -1, CVCL_ELAUNCH = -2, CVCL_EWORKSPACE = -3, CVCL_EUNSUPPORTED = -4 };
enum { CVCL_F32 = 0, CVCL_BF16 = 1 };
enum { CVCL_ACT_NONE = 0, CVCL_ACT_RELU = 1, CVCL_ACT_GELU = 2 };
/* BatchNorm accumulators (round 6).  A convolution hands its per-channel batch statistics on either as partial ROWS (one per
 * workgroup, reduced by cvcl_bn_finalize) or, with stats_rows == CVCL_STATS_ACCUMULATE, by atomically ADDING them to a caller-zeroed
 * accumulator: int64 [8][2][N] (8 rows = one per XCD; [0] sums, [1] sums of squares), fixed point with 24 fractional bits -- integer
 * addition commutes, so the totals are bit-deterministic.  sum = (acc[0][0][n] + ... + acc[7][0][n]) / 2^24.  The bf16 train-mode
 * trunk (cvcl_resnext50_fwd*) uses them so that the consumer of a raw tensor forms its BatchNorm affine itself instead of waiting
 * for a cvcl_bn_finalize launch ($CVCL_FINALIZE_ON_LOAD). */
#define CVCL_STATS_ACCUMULATE (-1)

int cvcl_abi_version(void);
const char* cvcl_last_error(void);

/* Optional measurement aid (bench.py roofline): while enabled, every kernel launch is bracketed by
 * hipEvents recorded on the stream the kernel is launched on; cvcl_prof_collect synchronises them and
 * returns the summed device time and launch count per kernel class.  Off by default (zero overhead).  */
enum { CVCL_K_GEMM = 0, CVCL_K_GCONV = 1, CVCL_K_STEM = 2, CVCL_K_BN_FINALIZE = 3, CVCL_K_BN_ADD_RELU = 4,
       CVCL_K_MAXPOOL = 5, CVCL_K_AVGPOOL = 6, CVCL_K_HEAD = 7, CVCL_K_OTHER = 8, CVCL_K_ATTENTION = 9,
       CVCL_K_LAYERNORM = 10, CVCL_K_LSTM = 11, CVCL_K_GEMM_F32 = 12, CVCL_K_BN_APPLY = 13, CVCL_K_BN_BWD = 14,
       CVCL_K_WGRAD = 15, CVCL_K_GEMM8W = 16, CVCL_K_GEMM_PRO = 17, CVCL_K_NCLASSES = 18 };
int cvcl_prof_enable(int on);
int cvcl_prof_collect(double* ms_per_class, long* launches_per_class, int n_classes);
/* average event-to-event time (us) of the same bracket around a kernel that does nothing, over n launches: the share of
 * every bracket that is dispatch gap rather than kernel (a rocprofv3 kernel duration = bracket - this, to first order).  */
int cvcl_prof_null_bracket_us(void* stream, int n, double* avg_us);

/* ------------------------------------------------------------------------------------------
 * Text encoder, "embedding" branch.  Replaces nn.Embedding + sum/len
 * (multimodal/multimodal.py:496-503; table built at :311-312, padding_idx = 0).
 *   table [V,E] f32, tok [B,L] i64, len [B] i64
 *   -> ret [B,E] f32 = sum_l table[tok[b,l]] / len[b];  out_ble [B,L,E] f32 = table[tok] (may be NULL)
 * bwd: d_table [V,E] f32 (fully overwritten; row 0 = 0) from d_ret [B,E]; deterministic
 * (b,l)-ordered accumulation, no atomics.  Token ids outside [0,V) are an error the kernel
 * reports by writing NaN into the affected rows (the reference raises an index error).        */
int cvcl_embed_meanpool_fwd(const float* table, const int64_t* tok, const int64_t* len, float* ret,
                            float* out_ble, int B, int L, int E, int V, void* stream);
int cvcl_embed_meanpool_bwd(const float* d_ret, const int64_t* tok, const int64_t* len, float* d_table,
                            int B, int L, int E, int V, void* stream);

/* F.normalize(x, p=2, dim=-1, eps) rows (multimodal/multimodal.py:736,743).
 *   x [N,E] f32 -> y [N,E], norm [N] (the un-clamped L2 norm, kept for bwd)
 * bwd: dx = (dy - y (y.dy)) / max(norm,eps)   (dy / eps where norm < eps).                     */
int cvcl_l2norm_fwd(const float* x, float* y, float* norm, int N, int E, float eps, void* stream);
int cvcl_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int N, int E, float eps,
                    void* stream);

/* Similarity logits (multimodal/multimodal.py:755,783-787):
 *   logits_per_image [Ni,Nt] = (img [Ni,E] . txt [Nt,E]^T) * exp(*neg_log_temp)
 * logits_per_text is the transposed view of the same buffer (bitwise: match.t()*s).  fp32 MFMA.
 * bwd: d_img = s * dS . txt, d_txt = s * dS^T . img, d_neg_log_temp = sum(dS * logits) (NULL to skip).
 * workspace: cvcl_sim_logits_bwd_workspace_bytes.                                               */
int cvcl_sim_logits_fwd(const float* img, const float* txt, const float* neg_log_temp, float* logits,
                        int Ni, int Nt, int E, void* stream);
size_t cvcl_sim_logits_bwd_workspace_bytes(int Ni, int Nt, int E);
int cvcl_sim_logits_bwd(const float* img, const float* txt, const float* neg_log_temp, const float* logits,
                        const float* d_logits, float* d_img, float* d_txt, float* d_neg_log_temp,
                        int Ni, int Nt, int E, void* workspace, size_t workspace_bytes, void* stream);
/* The same restricted to a range of rows (data-parallel global negatives, SURVEY.md 8e: every rank evaluates the replicated
 * N_g x N_g loss on the all-gathered features and back-propagates through ITS OWN rows only):
 *   d_img_rows [ni, E] = s * dS[i0 : i0 + ni, :] . txt          (image rows i0 .. i0 + ni - 1)
 *   d_txt_rows [nt, E] = s * dS[:, t0 : t0 + nt]^T . img        (text rows  t0 .. t0 + nt - 1)
 *   d_neg_log_temp = sum(dS * logits) over the WHOLE matrix (identical on every rank).
 * Operands are read in place (K-major GEMM operands): no workspace.  cvcl_sim_logits_bwd = the full ranges.                  */
int cvcl_sim_logits_bwd_rows(const float* img, const float* txt, const float* neg_log_temp, const float* logits,
                             const float* d_logits, float* d_img_rows, float* d_txt_rows, float* d_neg_log_temp,
                             int Ni, int Nt, int E, int i0, int ni, int t0, int nt, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Symmetric InfoNCE with arange labels + accuracies + entropies in one pass over the logits
 * (multimodal/multimodal.py:801-818, multimodal/utils.py:106-108).
 *   logits [N,N] f32 (= logits_per_image; logits_per_text is its transpose)
 *   -> scalars[5] = {infonce, image_accuracy, text_accuracy, image_entropy, text_entropy}
 *      row_lse [N], col_lse [N] (log-sum-exp per row / column, kept for bwd)
 * bwd: d_logits [N,N] = *d_loss/(2N) * (softmax_rows + softmax_cols - 2 I).
 * workspace (fwd): cvcl_infonce_workspace_bytes(N).                                             */
size_t cvcl_infonce_workspace_bytes(int N);
int cvcl_infonce_fwd(const float* logits, int N, float* scalars5, float* row_lse, float* col_lse,
                     void* workspace, size_t workspace_bytes, void* stream);
int cvcl_infonce_bwd(const float* logits, const float* row_lse, const float* col_lse, const float* d_loss,
                     float* d_logits, int N, void* stream);

/* get_entropy(logits, dim=-1) (multimodal/utils.py:106-108): out[r] = -sum_j p log p of softmax(x[r,:]). */
int cvcl_row_entropy(const float* x, float* out, int R, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * MFMA GEMM with fused prologue/epilogue -- the 1x1 convolutions of torchvision's Bottleneck
 * (call site multimodal/multimodal.py:101), nn.Linear fc/head (:190-192), ViT qkv/proj/fc1/fc2
 * (multimodal/vision_transformer_dino_mugs.py:92-94,113-115) and the text transformer linears.
 *
 *   C[M,N] = act( A'[M,K] . W[N,K]^T * (exp(*exp_scale)) + bias[N] ) (+ R[M,N])
 *   A'     = A, or relu?(A * a_scale[K] + a_shift[K])  (BatchNorm+ReLU of the producer fused into
 *            the operand load), rows optionally gathered with a spatial stride (1x1 stride-2 conv)
 *   stats  = per-column sum / sum-of-squares of the stored C (BatchNorm batch statistics of this
 *            conv's output), one partial row per persistent block: stats[grid_m][2][N] f32
 * All matrices row-major with the given leading dimensions (elements).  dtype selects the storage
 * type of A, W, C, R.                                                                            */
typedef struct {
    const void* A; const void* W; void* C;
    int M, N, K, lda, ldw, ldc;
    /* optional operand prologue (NULL = none) */
    const float* a_scale; const float* a_shift; int a_relu;
    /* optional row gather: output row m = (b, oy, ox) reads input row (b, oy*stride, ox*stride) */
    int gather_ho, gather_wo, gather_hi, gather_wi, gather_stride;   /* gather_stride 0/1 = off */
    /* optional epilogue */
    const float* exp_scale; const float* bias; int act;
    const void* R; int ldr;
    float* stats; int stats_rows;    /* stats_rows = capacity (>= cvcl_gemm_stats_rows(dtype, args)); rows written = that value.
                                      * stats_rows == CVCL_STATS_ACCUMULATE (bf16): stats points to int64 accumulators [8][2][N] that the
                                      * launch ADDS its per-channel (sum, sum of squares) to -- see "BatchNorm accumulators" below */
    /* Bottleneck tail (bf16): C = relu(round(A'W^T) * c_scale[N] + c_shift[N] + (R | R * r_scale[N] + r_shift[N])).
     * With C == NULL and stats != NULL the product is not stored, only its column statistics (same rounding). */
    const float* c_scale; const float* c_shift; const float* r_scale; const float* r_shift;
    /* training epilogues of the ViT MLP (bf16): C_pre != NULL with act = GELU also stores the pre-activation u = round(acc + bias)
     * (C = round(gelu(u))); G != NULL multiplies the rounded product by gelu'(G) (G [M][ldg] = the saved u): the data gradient
     * through the GELU, C = round(round(A W^T) * gelu'(G)).  Both NULL for every other use. */
    void* C_pre; const void* G; int ldg;
    /* centred storage of a raw convolution output (bf16 and fp32): with centre != NULL the product is stored (and its
     * statistics taken, and the Bottleneck tail's c_scale / c_shift applied) as round(A'W^T - centre[n]).  Train-mode
     * BatchNorm is invariant under a per-channel shift of its input, so consumers that apply (scale, shift) on load need no
     * change; cvcl_bn_finalize adds the centre back into the running mean.  Only the convolution epilogues honour it (no
     * bias / activation / residual-only epilogue).  See "Centred storage" below.                                          */
    const float* centre;
    /* Bottleneck tail with the downsample branch RECOMPUTED (bf16, the BN-prologue kernel, K = 128): instead of reading the
     * stored branch through R, identity = round(A2[M,K2] . W2[N,K2]^T - centre2[n]) * r_scale[n] + r_shift[n] with A2 = the block
     * input and W2 = the 1x1 downsample weight, K2 = 64 (torchvision Bottleneck.downsample of layer1.0: nn.Conv2d(64, 256, 1) +
     * BatchNorm).  R must be NULL; the statistics behind r_scale / r_shift come from cvcl_conv1x1_gram(A2) + cvcl_bn_from_gram(W2)
     * (or a statistics-only cvcl_gemm(A2, W2, C = NULL)). */
    const void* A2; const void* W2; int K2, lda2, ldw2; const float* centre2;
    /* LayerNorm folded into the linear (bf16, the 8-wave kernel only; reference vision_transformer_dino_mugs.py:136-149:
     * x + attn(norm1(x)), x + mlp(norm2(x)) -- nn.LayerNorm feeding nn.Linear).
     *   consumer (ln_stats != NULL; bias / activation epilogue, no residual): A = the RAW rows x, W = W diag(gamma) (rounded to
     *     bf16 by the caller), ln_colsum[n] = sum_k W'[n][k] of that rounded matrix, bias[n] = b[n] + sum_k W[n][k] beta[k]
     *     (required), ln_stats[m] = (rstd_m, -mean_m rstd_m):  C = round(act(rstd acc - mean rstd ln_colsum[n] + bias[n])).
     *     ln_stats is fetched in 16-byte granules: it must be readable for an EVEN number of rows (M + (M & 1)).
     *   producer (row_part != NULL; bias + residual epilogue): additionally row_part[m][N / 64][2] f32 = (sum, sum of squares) of
     *     the STORED row over each 64-column strip; cvcl_row_stats_finalize reduces them to the next consumer's ln_stats.
     * cvcl_gemm_ln_supported tells whether cvcl_gemm routes these arguments to the kernel that honours them. */
    const float* ln_stats; const float* ln_colsum; float* row_part;
    /* K-major operands and fused row sums (fp32 only, round 5): the gradient GEMMs of the trainable tail -- nn.Linear backward
     * (dW = dY^T X, dX = dY W, db = sum_m dY; reference call sites multimodal/multimodal.py:192 fc, :553-573 text transformer) and of
     * the similarity logits (:755) -- read their operands as they lie in memory instead of through transposed copies:
     *   a_trans != 0: element (m, k) of A is at A[k * lda + m]   (A is stored [K][M]);  w_trans != 0: W[k * ldw + n] (stored [K][N]);
     *   a_rowsum [M] f32 (needs a_trans): a_rowsum[m] = sum_k A'[m][k] -- with A' = dY^T this is the bias gradient, produced by the
     *   weight-gradient GEMM's own operand loads (fixed summation order, deterministic).
     * No prologue / gather / statistics / BN-tail options with these. */
    int a_trans, w_trans; float* a_rowsum;
    /* f32_split != 0 (fp32 only, round 5): fp32 operands, arithmetic on the bf16 MFMA with every element split into two bf16 parts
     * (hi.hi + hi.lo + lo.hi, fp32 accumulation): products good to ~2^-16 relative, ~4x the exact-fp32 kernel's speed.  For the
     * trainable tail of the bf16 configurations (the text transformer's linears and their gradients; the reference under bf16
     * autocast runs them as plain bf16 GEMMs); the fp32 parity mode never sets it. */
    int f32_split;
} cvcl_gemm_args;
int cvcl_gemm_grid_m(int dtype, int M, int N, int has_prologue);   /* needs a GPU (occupancy query) */
int cvcl_gemm(int dtype, const cvcl_gemm_args* args, void* stream);
/* exact number of statistics rows cvcl_gemm writes for these arguments when given a buffer of at least that many rows
 * (args->stats / stats_rows are ignored); ONLY those rows are defined afterwards -- reduce exactly that many.  With a smaller
 * buffer sized by cvcl_gemm_grid_m the 128-tile kernel runs and writes cvcl_gemm_grid_m rows.  (The 8-wave and the
 * BN-prologue kernels write fewer rows than cvcl_gemm_grid_m: always size and reduce by this function.)  */
int cvcl_gemm_stats_rows(int dtype, const cvcl_gemm_args* args);
/* The 8-wave 256 (224) x 256 bf16 kernel the dispatcher of cvcl_gemm selects for the MFMA-bound shapes (ResNeXt layers 2-4 1x1
 * convolutions, ViT linears; N % 256 == 0, K % 128 == 0), callable directly with the same argument block.
 * epi 0 = convolution epilogue (round + BN partial sums; C may be NULL), epi 1 = bias / activation / residual.
 * cvcl_gemm8w_tile_rows: the tile height (256 or 224) the epi-0 launch uses for an [M, N] output; cvcl_gemm8w_stats_rows: the
 * number of BN-statistics rows it writes.                                                                                */
int cvcl_gemm8w(int epi, const cvcl_gemm_args* args, void* stream);
/* The bandwidth-bound form cvcl_gemm selects when the operand carries the producer's BatchNorm + ReLU (a_scale / a_shift /
 * a_relu) and K = 128 | 256, N % 256 == 0 (conv3 of ResNeXt layers 1-2 on the raw grouped-convolution output: torchvision
 * Bottleneck.forward conv3(relu(bn2(.)))): statistics only (C NULL), C + statistics, or the Bottleneck tail (c_scale ...).
 * Round 6: the first two epilogues also take the operand as stored (a_scale == a_shift == NULL); cvcl_gemm routes such a product
 * here from 2^17 rows up (conv1 of ResNeXt layer2.0: K = 256, N = 256, M = 802 816 at B = 256).                            */
int cvcl_gemm_pro(const cvcl_gemm_args* args, void* stream);
int cvcl_gemm_pro_supported(const cvcl_gemm_args* args);
int cvcl_gemm_pro_stats_rows(int M, int N);
int cvcl_gemm8w_supported(int M, int N, int K, int lda, int ldw, int ldc);
/* Co-scheduling hint (round 4): the 8-wave GEMM kernels (bf16 and e4m3) run one 160-KiB-LDS workgroup per CU, so a launch that
 * fills the chip keeps every other stream's kernels out until its last round.  A host that keeps TWO passes of a frozen trunk in
 * flight on two streams sets cus = half the chip: each launch then takes twice as long on half the CUs and the two passes really run
 * side by side (ViT-B/16, B = 256, two trunk streams: 12.1 -> 11.9 ms per step; the ResNeXt trunk is faster with full grids).
 * Process-global, read at launch time (one host thread per process, as everything here); 0 = all CUs; returns the previous value.
 * Results do not depend on it (tile order only), except that BN-statistics row counts do: cvcl_gemm8w_stats_rows follows it. */
int cvcl_set_gemm_cu_share(int cus);
/* 1 when cvcl_gemm(CVCL_BF16, args) runs the 8-wave kernel for these arguments (shape policy included), i.e. ln_stats / row_part
 * would be honoured; 0 otherwise (cvcl_gemm then refuses arguments that carry them: CVCL_EUNSUPPORTED).  No GPU needed. */
int cvcl_gemm_ln_supported(const cvcl_gemm_args* args);
/* LayerNorm row statistics for the folded form above.  cvcl_row_stats: x [rows][D] (row stride in elements) ->
 * out[row] = (rstd, -mean rstd), biased variance + eps as nn.LayerNorm; cvcl_row_stats_finalize: the producer's strip partials
 * row_part[rows][strips][2] (strips = D / 64) -> the same.  fp64 combination of fp32 sums. */
int cvcl_row_stats(int dtype, const void* x, long x_row_stride, float* out, long rows, int D, float eps, void* stream);
int cvcl_row_stats_finalize(const float* row_part, int strips, float* out, long rows, int D, float eps, void* stream);
int cvcl_gemm8w_tile_rows(int M, int N);
int cvcl_gemm8w_stats_rows(int M, int N);

/* Train-mode BatchNorm statistics of a 1x1 convolution's output WITHOUT forming the output (round 4, csrc/bn_gram.hip): torchvision
 * Bottleneck.forward bn3(conv3(relu(bn2(.)))) / downsample[1](downsample[0](x)), reached from multimodal/multimodal.py:101, where the
 * convolution runs fused with its consumer and its statistics are needed first.  With a' = relu?(A * a_scale + a_shift) rounded to bf16
 * (a_scale NULL: A as stored), cvcl_conv1x1_gram leaves G = sum_m a'[m] a'[m]^T (only its upper-triangle 32 x 32 tiles are computed) and
 * s = sum_m a'[m] in fp64 inside the workspace (*gram_out: G [K][K] row-major, then s [K]); cvcl_bn_from_gram turns them into what
 * cvcl_bn_finalize produces for y = a' W^T (W bf16 [N][ldw]): mean = w.s / M, var = w^T (G - s s^T / M) w / M, (scale, shift) of the
 * stored tensor y - centre, running statistics or deferred moments.  bf16 rows, K = 64 | 128 | 256; deterministic.               */
size_t cvcl_conv1x1_gram_workspace_bytes(int K);
int cvcl_conv1x1_gram(const void* A, int lda, long M, int K, const float* a_scale, const float* a_shift, int a_relu, void* workspace,
                      size_t workspace_bytes, const double** gram_out, void* stream);
int cvcl_bn_from_gram(const double* gram, int K, long count, const void* W, int ldw, int N, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* moments, int moments_ld, const float* centre, void* stream);

/* out[c][r] = in[r][c], f32 (operand re-layout for the weight-gradient GEMMs). */
int cvcl_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
/* d_bias[n] = sum_m dY[m][n], f32. */
int cvcl_colsum_f32(const float* dY, float* d_bias, int M, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * ResNeXt-50 32x4d trunk = torchvision.models.resnext50_32x4d (third-party; reference call sites
 * multimodal/multimodal.py:96-102,155-158, multimodal/utils.py:207-209).  NHWC activations.
 * "raw" = convolution output before BatchNorm; every conv kernel emits per-channel sum / sumsq
 * partial rows stats[rows][2][C] of what it stored; cvcl_bn_finalize turns them into (scale, shift)
 * + running-stat EMA (nn.BatchNorm2d train mode: eps 1e-5, momentum 0.1, unbiased running var,
 * num_batches_tracked += 1); consumers apply scale/shift(+ReLU) on load.
 *
 * Centred storage.  A raw conv output whose per-channel batch mean is large next to its spread loses precision when it
 * is rounded to bf16 BEFORE BatchNorm subtracts the mean (the rounding error 2^-9 |y| becomes 2^-9 |y| / sigma of the
 * normalised value).  Every convolution entry therefore takes an optional per-output-channel `centre` c (fp32, device):
 * it stores round(y - c) and takes the statistics of THAT; BatchNorm(y - c) == BatchNorm(y), so (scale, shift) from
 * cvcl_bn_finalize apply to the stored tensor unchanged and the finalize adds c back where the true mean is needed
 * (running_mean, moments).  c only has to be within ~sigma of the batch mean: the host keeps the batch means of a
 * calibration pass (multimodal/resnext.py).  centre == NULL is c = 0 (plain storage).                               */
enum { CVCL_PACK_DENSE = 0, CVCL_PACK_STEM7 = 1, CVCL_PACK_GCONV3 = 2 };

typedef struct {
    const void* w;                 /* conv weight packed by cvcl_pack_conv_weight (dtype of the run) */
    const float* gamma; const float* beta;          /* BatchNorm weight / bias                       */
    float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* updated when training */
} cvcl_convbn_params;

/* centre (nullable): the c the producer stored its output with; running_mean receives batch mean + c.
 * cvcl_bn_eval_affine: shift = beta - (running_mean - c) * scale, the eval-mode affine of a tensor stored as y - c.  */
int cvcl_bn_finalize(const float* stats, int rows, long count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                     float eps, float* scale, float* shift, const float* centre, int C, void* stream);
int cvcl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* scale, float* shift, const float* centre, int C, void* stream);
int cvcl_col_stats_rows(long rows);
int cvcl_col_stats(int dtype, const void* x, long rows, int C, float* stats, int stats_rows, void* stream);

/* weights arrive in the reference layout (OIHW fp32, device) and are re-laid-out once per weight version */
size_t cvcl_packed_weight_bytes(int dtype, int kind, int cout, int cin_per_group, int k);
int cvcl_pack_conv_weight(int dtype, int kind, const float* w_oihw, void* out, int cout, int cin_per_group, int k,
                          void* stream);

/* conv1 7x7/2 pad 3, 3->64: x NCHW f32 [B,3,H,W] -> y NHWC raw [B,H/2,W/2,64] (+stats) */
int cvcl_stem_conv_stats_rows(int dtype, int B, int H, int W);
int cvcl_stem_conv7x7(int dtype, const float* x_nchw, const void* w_packed, void* y_nhwc, float* stats, int stats_rows,
                      const float* centre /* [64] or NULL */, int B, int H, int W, void* stream);
/* (bf16: y_nhwc == NULL = statistics only -- the first pass of the fused stem below.)
 * conv1 + bn1 + relu + maxpool in ONE pass (round 5; torchvision ResNet.forward conv1 -> bn1 -> relu -> maxpool, reached from
 * multimodal/multimodal.py:101): the raw [B,H/2,W/2,64] tensor is never written -- in train mode its BatchNorm statistics come from
 * a statistics-only cvcl_stem_conv7x7 pass, then this kernel recomputes the convolution and pools it out of LDS:
 * y NHWC bf16 [B, ceil(H/4), ceil(W/4), 64], bit-identical to cvcl_stem_conv7x7 + cvcl_bn_relu_maxpool.  bf16, W <= 252. */
int cvcl_stem_pool_supported(int dtype, int H, int W);
int cvcl_stem_pool(int dtype, const float* x_nchw, const void* w_packed, const float* scale, const float* shift,
                   const float* centre /* [64] or NULL */, void* y_nhwc, int B, int H, int W, void* stream);
/* relu(bn(x)) then maxpool 3x3/2 pad 1: [B,H,W,C] -> [B,ceil(H/2),ceil(W/2),C] */
int cvcl_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int B, int H, int W,
                         int C, void* stream);
/* grouped 3x3 conv pad 1 stride 1|2 on relu(x*a_scale+a_shift): [B,H,W,C] -> raw [B,Ho,Wo,C] (+stats).
 * a_scale == a_shift == NULL: plain convolution of x (no affine, no ReLU) -- the data-gradient form. */
int cvcl_gconv3x3_stats_rows(int dtype, int B, int H, int W, int C, int stride);
int cvcl_gconv3x3(int dtype, const void* x, const float* a_scale, const float* a_shift, const void* w_packed, void* y,
                  float* stats, int stats_rows, const float* centre /* [C] or NULL */, int B, int H, int W, int C, int groups,
                  int stride, void* stream);
/* out = relu(raw*scale+shift + (idn | idn*idn_scale+idn_shift)), [rows, C] */
int cvcl_bn_add_relu(int dtype, const void* raw, const float* scale, const float* shift, const void* idn,
                     const float* idn_scale, const float* idn_shift, void* out, long rows, int C, void* stream);
/* y = relu(x*scale+shift), [rows, C]; y may alias x */
int cvcl_bn_relu_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, long rows, int C,
                       void* stream);
/* adaptive avgpool to 1x1 + flatten: [B,HW,C] -> [B,C] f32 */
int cvcl_avgpool(int dtype, const void* x, float* out, int B, int HW, int C, void* stream);

/* Whole trunk in one call: conv1..layer4 + avgpool.  layers[53] in torchvision state_dict order
 * (conv1; per block conv1, conv2, conv3, [downsample.0]).  training != 0: batch statistics + running
 * stat updates (what the reference does even with a frozen CNN: Lightning keeps .train()).
 * -> layer4_out_nhwc [B,H/32,W/32,2048] (dtype), pooled [B,2048] f32.
 * centres (nullable): [53][2048] fp32 = cvcl_resnext50_centres_floats(), layer l's storage centre at centres + 2048 l
 * ("Centred storage" above).  NULL = plain storage, except eval mode in bf16, where NULL selects c = running_mean (the
 * stored tensors are then y - running_mean; $CVCL_CENTRED_STORAGE=0 turns that default off).                         */
size_t cvcl_resnext50_workspace_bytes(int dtype, int B, int H, int W);
size_t cvcl_resnext50_centres_floats(void);
int cvcl_resnext50_fwd(int dtype, int B, int H, int W, int training, const float* x_nchw,
                       const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                       void* layer4_out_nhwc, float* pooled, float momentum, float eps, const float* centres, void* stream);
/* ONE Bottleneck of that trunk (torchvision.models.resnet.Bottleneck.forward, reached from multimodal.py:101), enqueued as
 * exactly the launch sequence cvcl_resnext50_fwd uses for it -- the unit the teacher-forced parity tests drive: feed the
 * oracle's block input, compare the block output.  stage 0..3 = layer1..layer4; first != 0 for the stage's first block
 * (downsample branch, stride 2 when stage > 0); layers = conv1, conv2, conv3[, downsample] of the block (3 or 4 entries);
 * x_nhwc [B,h,w,Cin] -> out_nhwc [B,h/s,w/s,256 << stage].                                                              */
size_t cvcl_resnext50_block_workspace_bytes(int dtype, int B, int h, int w, int stage);
int cvcl_resnext50_block_fwd(int dtype, int B, int h, int w, int stage, int first, int training, const void* x_nhwc,
                             const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                             void* out_nhwc, float momentum, float eps, const float* centres /* [n_layers][2048] or NULL */,
                             void* stream);
/* The train-mode pass with its BatchNorm running-statistics update split off (no counterpart in the reference, which runs
 * one pass at a time; same results).  Consecutive passes of a FROZEN trunk are independent except for those 53 EMA updates
 * (torch.nn.BatchNorm2d train mode, reached from multimodal.py:88-104 because Lightning keeps .train()), so a host may enqueue
 * pass k+1 on a second stream beside pass k and apply the updates in pass order:
 *   cvcl_resnext50_fwd_deferred_stats  = cvcl_resnext50_fwd(training = 1) that writes every layer's batch (mean, unbiased
 *       variance) to moments ([53][2][2048] floats = cvcl_resnext50_moments_floats()) and touches no BatchNorm buffer;
 *   cvcl_resnext50_apply_moments       = the 53 updates r <- (1 - momentum) r + momentum x and num_batches_tracked += 1 in one
 *       launch; enqueue it behind the previous pass's apply.  Bit-identical to the updates of cvcl_resnext50_fwd.          */
size_t cvcl_resnext50_moments_floats(void);
int cvcl_resnext50_fwd_deferred_stats(int dtype, int B, int H, int W, const float* x_nchw,
                                      const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                                      void* layer4_out_nhwc, float* pooled, float eps, float* moments, const float* centres,
                                      void* stream);
int cvcl_resnext50_apply_moments(const cvcl_convbn_params* layers, int n_layers, const float* moments, float momentum,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * DINO ViT image encoder (multimodal/vision_transformer_dino_mugs.py:87-250), the one-layer text
 * transformer (multimodal/multimodal.py:553-573) and the LSTM text encoder (:513-552).  Linears run on
 * cvcl_gemm; these are the non-GEMM pieces.                                                        */
/* PatchEmbed unfold (vit:162,166): x NCHW f32 -> cols [B*np, Kpad] (k = c*p*p + ky*p + kx, zero padded) */
int cvcl_im2col_patches(int dtype, const float* x_nchw, void* cols, int B, int H, int W, int patch, int Kpad,
                        void* stream);
/* prepare_tokens (vit:232-243): h[b,0] = cls + pos[0]; h[b,1+i] = tok[b,i] + pos[1+i]; h [B,T,D]     */
int cvcl_vit_assemble_tokens(int dtype, const void* tok, const float* cls, const float* pos, void* h, int B, int T,
                             int D, void* stream);
/* nn.LayerNorm over the last dim; rows may be strided (x_row_stride elements); y is dtype or f32      */
int cvcl_layernorm(int dtype, const void* x, long x_row_stride, const float* gamma, const float* beta, float eps,
                   void* y, int y_is_f32, long rows, int D, void* stream);
/* softmax(q k^T * scale [+ key padding mask where key_tok == 0]) v;  qkv [B,T,3,heads,hd] -> out [B,T,heads*hd]
 * (vit:119-127; nn.MultiheadAttention inside nn.TransformerEncoderLayer).  bf16, hd = 64, no mask -> MFMA kernel. */
int cvcl_attention(int dtype, const void* qkv, const int64_t* key_tok, void* out, int B, int T, int heads, int head_dim,
                   float scale, void* stream);
/* same attention (bf16 qkv, head_dim 64, 32 < T), output written as e4m3 [B*T][heads*64] + e8m0 block scales tiled
 * [heads*64/128][B*T][4] -- cvcl_gemm_fp8_mx's MX input, so the fp8 projection needs no quantisation pass in between. */
int cvcl_attention_mx(const void* qkv, void* out8, void* out_block_scales, int B, int T, int heads, int head_dim, float scale,
                      void* stream);
/* Attention for a fine-tuned ViT (autograd of vision_transformer_dino_mugs.py:106-130): the forward above that also saves the
 * log-sum-exp of every row (lse [B][heads][T] fp32, log2 units), and the backward: d_qkv [B][T][3][heads][64] bf16 from
 * qkv, the forward output o and d_o ([B][T][heads*64] bf16).  Two kernels (queries own dQ; keys own dK, dV), probabilities
 * rebuilt from lse, no atomics: deterministic.  head_dim 64; 32 < T <= 288. */
int cvcl_attention_train(const void* qkv, void* out, float* lse, int B, int T, int heads, int head_dim, float scale, void* stream);
int cvcl_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* d_qkv, int B, int T, int heads,
                       int head_dim, float scale, void* stream);
/* LayerNorm backward on bf16 rows (32 lanes per row, D % 8 == 0, D <= 1024): dx = rstd (g - mean(g) - xhat mean(g xhat)) with
 * g = dy gamma, plus `add` (nullable: the residual-stream gradient that bypasses the norm); dy bf16 or fp32; per-group partial
 * sums of dgamma = sum dy xhat and dbeta = sum dy in partial [cvcl_layernorm_bwd_rows_partials(rows)][2][D] (reduce with
 * cvcl_colsum_f32: fixed order, deterministic).                                                                          */
int cvcl_layernorm_bwd_rows_partials(long rows);
int cvcl_layernorm_bwd_rows(const void* x, long x_row_stride, const float* gamma, const void* dy, int dy_is_f32, long dy_row_stride,
                            float eps, const void* add, void* dx, long dx_row_stride, float* partial, long rows, int D, void* stream);
/* GELU (erf form) on bf16: d_y == NULL -> y = gelu(u); else y = d_y * gelu'(u) (u = the saved pre-activation).  n % 8 == 0 */
int cvcl_gelu_bf16(const void* u, const void* d_y, void* y, long n, void* stream);
/* backward of cvcl_vit_assemble_tokens: d_tok [B][T-1][D] bf16 = the patch rows of dh [B][T][D]; d_pos [T][D] fp32 = sum over
 * the batch (its row 0 is also d_cls)                                                                                     */
int cvcl_vit_tokens_bwd(const void* dh, void* d_tok, float* d_pos, int B, int T, int D, void* stream);
/* x[b,l,:] = table[tok[b,l]] (+ pos[l]) (multimodal.py:496, 561-563) */
int cvcl_embed_gather_pos(const float* table, const int64_t* tok, const float* pos, float* x, int B, int L, int E, int V,
                          void* stream);
/* ret[b,:] = sum over ALL L positions of x[b,l,:] / len[b] (multimodal.py:573) */
int cvcl_seq_sum_div(const float* x, const int64_t* len, float* ret, int B, int L, int E, void* stream);
/* one nn.LSTM step on precomputed gates [B,4H] (order i,f,g,o); rows with len <= t keep (h,c) and emit zeros */
int cvcl_lstm_cell(const float* gates, const int64_t* len, int t, float* h, float* c, float* out, int B, int L, int Hd,
                   void* stream);

/* Training side of the text encoders (the reference trains them under Lightning's .train(): dropout 0.1 in
 * nn.TransformerEncoderLayer, LockedDropout(dropout_i) ahead of the LSTM; multimodal/multimodal.py:46-53,513-573).
 * All f32, deterministic.  Dropout uses a counter-based hash of (seed, element index): the same call is the
 * backward.  shared_period > 0 shares the mask along a dimension of that extent (LockedDropout's [B,1,E] mask).  */
int cvcl_dropout(const float* x, const float* residual, float* y, long n, float p, unsigned long long seed,
                 long shared_period, long inner, void* stream);      /* y = dropout(x) (+ residual) */
/* LayerNorm backward: dx, plus dy*xhat per element (column-summed by cvcl_colsum_f32 into dgamma; dbeta = colsum(dy)) */
int cvcl_layernorm_bwd(const float* x, const float* gamma, const float* dy, float eps, float* dx, float* dy_xhat, long rows,
                       int D, void* stream);
int cvcl_relu_bwd(const float* y, const float* dy, float* dx, long n, void* stream);
/* d_table[v] = sum of dx rows whose token is v, in position order; row 0 (padding_idx) = 0; fully overwritten */
int cvcl_embed_rows_bwd(const float* dx, const int64_t* tok, float* d_table, int n_pos, int E, int V, void* stream);
int cvcl_seq_sum_div_bwd(const float* d_ret, const int64_t* len, float* dx, int B, int L, int E, void* stream);
/* nn.MultiheadAttention core for short sequences (T <= 32) with key padding mask and probability dropout:
 * forward when out != NULL, backward when d_qkv != NULL (P is recomputed).  qkv [B,T,3,heads,hd] f32.            */
int cvcl_attention_small(const float* qkv, const int64_t* key_tok, const float* d_out, float* out, float* d_qkv, int B,
                         int T, int heads, int head_dim, float scale, float dropout_p, unsigned long long seed,
                         void* stream);
/* nn.LSTM step t that saves, in [B,L,.] layout (row b*L+t), the gate activations, c_t and h_{t-1} for BPTT; and the
 * BPTT step: d_gates rows b*L+t of [B,L,4H], dc updated in place, dh_carry = dh where the step was masked        */
int cvcl_lstm_cell_train(const float* gates, const int64_t* len, int t, float* h, float* c, float* out /*[B,L,H] or NULL*/,
                         float* gates_act, float* c_save, float* h_prev_save, int B, int L, int Hd, void* stream);
int cvcl_lstm_cell_bwd(const float* gates_act, const float* c_save, const int64_t* len, int t, const float* dh, float* dc,
                       float* d_gates, float* dh_carry, int B, int L, int Hd, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward side of the ResNeXt trunk for --finetune_cnn (multimodal/multimodal.py:175-179: the CNN's parameters keep
 * requires_grad, so autograd runs through torchvision's Bottlenecks).  Used by the autograd-composed fine-tuning
 * path; [rows, C] = NHWC activations flattened over pixels.                                              */
int cvcl_bn_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, long rows, int C, int relu,
                  void* stream);
/* train-mode BatchNorm backward, g = dy * mask:
 *   mode 0: no mask | mode 1: [x*scale+shift > 0] (ReLU right after the BN; recomputed, y is not read) |
 *   mode 2: [out > 0], out = relu(bn(x) + identity) the Bottleneck output; g_out (optional) receives g = d identity.
 *   dbeta = sum g;  dgamma = sum g * xhat;  dx = gamma * rstd * (g - dbeta/n - xhat * dgamma/n)
 * partial: scratch [partial_rows >= cvcl_bn_bwd_partial_rows(dtype, rows, C)][2][C] f32;  coef: scratch [3][C] f32 */
int cvcl_bn_bwd_partial_rows(int dtype, long rows, int C);
int cvcl_bn_bwd(int dtype, int mode, const void* x, const void* out, const void* dy, const float* scale, const float* shift,
                const float* mean, const float* rstd, const float* gamma, float* dgamma, float* dbeta, void* dx, void* g_out,
                long rows, int C, float* partial, int partial_rows, float* coef, void* stream);
/* batch mean and 1/sqrt(var_biased + eps) from the forward statistics rows (what bn_finalize normalised with; with centred
 * storage they are the moments of the STORED tensor y - c, which is what cvcl_bn_bwd needs next to that tensor).
 * centre_track (nullable, [C]): the c the producer used, updated in place to c + mean = the batch mean of y -- the storage
 * centre of the next step (a trunk that is being fine-tuned re-centres every step: its weights move).                 */
int cvcl_bn_batch_moments(const float* stats, int stats_rows, long count, float eps, float* mean, float* rstd,
                          float* centre_track, int C, void* stream);
/* OIHW f32 weight of the convolution that computes the data gradient of a grouped 3x3 conv (flip + in/out swap per group) */
int cvcl_gconv_weight_dgrad(const float* w, float* out, int C, int cin_per_group, void* stream);
int cvcl_transpose(int dtype, const void* in, void* out, long rows, int cols, void* stream);     /* out[c][r] = in[r][c] */
int cvcl_add(int dtype, const void* a, const void* b, void* y, long n, int relu, void* stream);  /* y = a + b (ReLU if relu) */
int cvcl_relu_mask(int dtype, const void* y, const void* dy, void* dx, long n, void* stream);    /* dx = dy where y > 0 */
/* max pool 3x3/2 pad 1, NHWC: dy == NULL -> forward (out = pooled); else backward (out = dx, first arg-max wins) */
int cvcl_maxpool3x3s2(int dtype, const void* x, const void* dy, void* out, int B, int H, int W, int C, void* stream);
/* same pooling with the arg-max recorded (idx [B,Ho,Wo,C] u8, window position 0..8, first maximum): dy == NULL ->
 * forward (reads x, writes out = pooled and idx); else backward from idx (x unused, out = dx)               */
int cvcl_maxpool3x3s2_idx(int dtype, const void* x, const void* dy, void* out, uint8_t* idx, int B, int H, int W, int C,
                          void* stream);
int cvcl_avgpool_bwd(int dtype, const float* d_pooled, void* dx, int B, int HW, int C, void* stream);
/* z[b,2oy,2ox,:] = dy[b,oy,ox,:], zeros elsewhere ([B,2Ho,2Wo,C]): data gradient of a stride-2 conv = stride-1 conv of z */
int cvcl_zero_stuff2(int dtype, const void* dy, void* z, int B, int Ho, int Wo, int C, void* stream);
/* dW[co][ci][ky][kx] (f32, reference OIHW layout) of a k x k (grouped) convolution, direct form */
int cvcl_conv_wgrad_direct(int dtype, const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                           int cin_per_group, int k, int stride, int pad, int x_is_nchw_f32, void* stream);

/* Weight-gradient ("TN") GEMM: C[n][k] = sum_m A[m][n] * B[m][k], A [M, lda >= N], B [M, ldb >= K] row-major in the
 * run dtype, C [N, k_keep] f32 (k_keep <= K drops trailing padded columns).  1x1-conv weight gradient with A = dY,
 * B = X (NHWC, no transposed copies); the contraction is split over M with a fixed-order reduction (deterministic).
 * workspace >= cvcl_gemm_tn_workspace_bytes(dtype, M, N, K).                                             */
size_t cvcl_gemm_tn_workspace_bytes(int dtype, long M, int N, int K);
int cvcl_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, long M, int N, int K, float* C, int k_keep,
                 void* workspace, size_t workspace_bytes, void* stream);
/* nn.Linear backward in one pass over dY (bf16): C = A^T B as above and colsum[n] = sum_m A[m][n] (the bias gradient) */
size_t cvcl_gemm_tn_colsum_workspace_bytes(long M, int N, int K);
int cvcl_gemm_tn_colsum(const void* A, int lda, const void* B, int ldb, long M, int N, int K, float* C, int k_keep, float* colsum,
                        void* workspace, size_t workspace_bytes, void* stream);
/* grouped 3x3 (pad 1, stride 1|2) weight gradient, bf16 activations: dW [C][C/groups][3][3] f32 (reference OIHW) */
size_t cvcl_gconv3x3_wgrad_workspace_bytes(int B, int H, int W, int C, int stride);
int cvcl_gconv3x3_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int C, int groups, int stride,
                        void* workspace, size_t workspace_bytes, void* stream);
/* bf16 patch matrix of the 7x7/2 stem: col [B*H/2*W/2][160] (147 = 3*7*7 columns (c, ky, kx), zero-padded to 160);
 * stem weight gradient = cvcl_gemm_tn(A = dY [P,64], B = col, k_keep = 147)                              */
int cvcl_stem_im2col(const float* x_nchw, void* col_bf16, int B, int H, int W, void* stream);

/* embedding_type == "spatial", sim == "max" (multimodal/multimodal.py:770-787).  mm [Bi*HW, Bt*L] f32 is the match map
 * <image location, word> (one cvcl_gemm of the per-location image rows [Bi*HW, E] against the per-word text rows
 * [Bt*L, E]);  logits[i][t] = exp(*neg_log_temp) * sum_l max_p mm[(i,p)][(t,l)] / len[t]  (all L positions, as the
 * reference), arg [Bi, Bt*L] u8 = the winning location (first maximum).  HW <= 256.  The backward writes the dense
 * d_mm (zero except at the winners), from which d image rows = d_mm . text rows and d text rows = d_mm^T . image rows.
 * sim == "mean" needs no kernel of its own: mean over locations / words (cvcl_seq_sum_div) then cvcl_sim_logits.     */
int cvcl_bf16_to_f32(const void* x, float* y, long n, void* stream);      /* n % 8 == 0, 16-byte aligned */
/* its backward under --finetune_cnn (the gradient of the layer-4 map returns to the bf16 trunk, round to nearest even):
 * reference multimodal/multimodal.py:175-185 lets autograd run through nn.Sequential(trunk, Conv2d(2048, E, 1))          */
int cvcl_f32_to_bf16(const float* x, void* y, long n, void* stream);      /* n % 8 == 0, 16-byte aligned */
int cvcl_spatial_max_fwd(const float* mm, const int64_t* len, const float* neg_log_temp, float* logits, uint8_t* arg,
                         int Bi, int HW, int Bt, int L, void* stream);
int cvcl_spatial_max_bwd(const float* d_logits, const uint8_t* arg, const int64_t* len, const float* neg_log_temp,
                         const float* logits, float* d_mm, float* d_neg_log_temp /* nullable */, int Bi, int HW, int Bt, int L,
                         void* stream);

/* Language-model loss (lambda_lm > 0 configs; multimodal/multimodal.py:861-890, multimodal_lit.py:266-300).
 * token-wise F.cross_entropy(logits [R,V], labels [R], ignore_index, reduction "none"): loss[r] (0 for ignored rows), lse[r]
 * saved for the backward d_logits = (softmax - onehot) * d_loss.                                               */
/* BPTT with gradients on the per-step outputs: dh[b] += d_out[b][t] where len[b] > t (before cvcl_lstm_cell_bwd of step t) */
int cvcl_lstm_add_dout(float* dh, const float* d_out, const int64_t* len, int t, int B, int L, int Hd, void* stream);
/* remaining --text_encoder choices (multimodal.py:505-552).  cvcl_seq_reverse: y[b][t] = x[b][len-1-t] (t < len, else 0):
 * input / output permutation of the backward LSTM direction, self-adjoint.  cvcl_scale_add_f32: y = alpha * (a + b), b
 * nullable (mean of the two directions and its backward).  cvcl_cbow: window sum without the centre / (2 crange),
 * self-adjoint. */
int cvcl_seq_reverse(const float* x, const int64_t* len, float* y, int B, int L, int E, void* stream);
int cvcl_scale_add_f32(const float* a, const float* b, float alpha, float* y, long n, void* stream);
int cvcl_cbow(const float* x, float* y, int B, int L, int E, int crange, void* stream);

/* ---- f3: the training-time frame transform on the device ----------------------------------------------------------------
 * Replaces the per-frame PIL pipeline of multimodal_data_module.py:244-256 (RandomResizedCrop((224,224), scale (0.2,1)) ->
 * RandomApply([GaussianBlur([.1,2.])], p .5) (utils.py:94-103) -> RandomHorizontalFlip -> ToTensor -> Normalize (:57)) for a
 * whole batch in one launch, bit-identically to Pillow's integer pixel arithmetic (Resample.c bilinear with the down-scale
 * widened triangle filter, BoxBlur.c three-pass fractional box blur) and torch's fp32 (u8 / 255 - mean) / std.
 *   frames      uint8 [B][H][W][3]   decoded RGB frames (HWC), device memory
 *   crop        int32 [B][4]         top, left, h, w per frame (RandomResizedCrop.get_params order), device memory
 *   blur_sigma  fp32  [B]            <= 0 where RandomApply skipped the blur, device memory
 *   flip        int32 [B]            device memory
 *   mean, std3  3 HOST floats each
 *   out         fp32  [B][3][out_h][out_w]; out_u8 (nullable): uint8 [B][out_h][out_w][3], the image before ToTensor
 *   max_crop_h  upper bound of crop[:, 2] (H always works): sizes the LDS plan; boxes taller than ~500 rows at 224 x 224
 *               output do not fit the single-pass plan and are refused (CVCL_EARG)                                            */
int cvcl_augment_frames(const void* frames, int B, int H, int W, const int32_t* crop, const float* blur_sigma, const int32_t* flip,
                        const float* mean, const float* std3, void* out, int out_h, int out_w, void* out_u8, int max_crop_h,
                        void* stream);
int cvcl_token_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* lse, long R, int V, int ignore_index,
                      void* stream);
int cvcl_token_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* d_loss, float* d_logits,
                      long R, int V, int ignore_index, void* stream);
/* the three masked means of multimodal_lit.py:284-300 and their token counts.  d_loss == NULL: forward (means[3],
 * counts[3] written); else backward: d_loss[r] = sum_k d_means[k] * mask_k[r] / counts[k].                      */
int cvcl_lm_loss_summaries(const float* loss, const int64_t* labels, const float* d_means, float* means, float* counts,
                           float* d_loss, int R, int pad, int sos, int eos, void* stream);

/* ------------------------------------------------------------------------------------------
 * FP8 (OCP e4m3) linears for the ViT encoder -- BASELINE configs[4] "fp8 MFMA weights/activations".
 * cvcl_quant_rows_fp8: q[r][:] = e4m3(y[r][:] / s[r]), s[r] = amax_k |y[r][k]| / 448 (1 for a zero row), where y = x or,
 *   with ln_gamma/ln_beta, nn.LayerNorm(x) (the vit blocks' norm1 / norm2 fused with the quantisation).  src_dtype: CVCL_BF16
 *   activations or CVCL_F32 (weights: one scale per output channel).  K % 8 == 0, K <= 4096.
 * cvcl_gemm_fp8: C[M,N] bf16 = act((A8 . W8^T) * a_scale[m] * w_scale[n] + bias[n]) (+ R), products on
 *   v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales, fp32 accumulation.  K % 128 == 0, N % 128 == 0.     */
int cvcl_quant_rows_fp8(int src_dtype, const void* x, long x_row_stride, const float* ln_gamma, const float* ln_beta,
                        float ln_eps, void* q, float* scale, long rows, int K, void* stream);
/* MX variants: a_block_scales = one e8m0 byte (2^(b-127)) per 32-element block of A instead of the per-row fp32 scale (the
 * scaled MFMA applies it in hardware), tiled [K/128][M][4] (block k/32 of row m at ((k/128)*M + m)*4 + (k/32)%4, so 32 rows'
 * words are contiguous); c8 / c_block_scales = e4m3 output [M][ldc8] + e8m0 scales in the same tiling [N/128][M][4]
 * (2^ceil(log2(amax/448)) per block) instead of bf16 C, ready to be the next GEMM's MX input -- no separate quantisation pass.
 * Large shapes (>= 96 tiles of 256 x 256 with a well-filled last round) run on the 8-wave kernel (csrc/gemm8f_kernel.h), which
 * fetches a tile's a_scale / a_block_scales in 16-byte granules: both must be readable up to the next multiple of 16 bytes
 * (a_scale: M rounded up to 4 floats; a_block_scales: 12 bytes past its end when M % 4 != 0).  Same results either way.            */
int cvcl_gemm_fp8_mx(const void* A8, const float* a_scale, const void* a_block_scales, int lda, const void* W8,
                     const float* w_scale, int ldw, void* C, int ldc, void* c8, void* c_block_scales, int ldc8,
                     const float* bias, int act, const void* R, int ldr, int M, int N, int K, void* stream);
/* The same with every option in one block (round 5), plus nn.LayerNorm FOLDED into the e4m3 linear it feeds (reference
 * vision_transformer_dino_mugs.py:136-149: x + attn(norm1(x)), x + mlp(norm2(x)); the bf16 form is cvcl_gemm_args.ln_stats):
 *   producer (row_part != NULL; proj / fc2: MX input, bias + residual): besides the bf16 rows C = round(acc sw + bias) + R the kernel
 *     writes their MX-quantised copy c8 / c_block_scales -- the RAW operand of the next qkv / fc1, e8m0 per 32 elements, no row-wide
 *     amax needed -- and row_part[m][N / 64][2] = (sum, sum of squares) of the stored row per 64-column strip (cvcl_row_stats_finalize
 *     turns them into ln_stats);
 *   consumer (ln_stats != NULL; qkv / fc1: MX input = those raw rows): W8 = e4m3(W diag(gamma)) with row scales w_scale,
 *     ln_colsum[n] = w_scale[n] * sum_k W8[n][k], bias[n] = b[n] + sum_k W[n][k] beta[k], ln_stats[m] = (rstd, -mean rstd):
 *     y = act(rstd (A8 . W8^T) w_scale - mean rstd ln_colsum + bias), to bf16 C or (c8 != NULL) to MX output.  8-wave kernel only:
 *     cvcl_gemm_fp8_ln_supported(M, N, K).  ln_stats readable for an even number of rows.
 * The 24 LayerNorm + row-quantise passes of a ViT-B (1.0 ms of an 8.2 ms step at B = 256) go; cvcl_quant_rows_mx quantises the
 * assembled tokens once per forward.                                                                                          */
typedef struct {
    const void* A8; const float* a_scale; const void* a_block_scales; int lda;
    const void* W8; const float* w_scale; int ldw;
    void* C; int ldc; void* c8; void* c_block_scales; int ldc8;
    const float* bias; int act; const void* R; int ldr;
    int M, N, K;
    const float* ln_stats; const float* ln_colsum; float* row_part;
} cvcl_gemm_fp8_args;
int cvcl_gemm_fp8_ex(const cvcl_gemm_fp8_args* args, void* stream);
int cvcl_gemm_fp8_ln_supported(int M, int N, int K);
/* bf16 rows [rows][K] (K % 128 == 0) -> e4m3 q [rows][K] + e8m0 block scales tiled [K / 128][rows][4] (one per 32 elements; the
 * quantiser of the MX-output epilogues, bit for bit). */
int cvcl_quant_rows_mx(const void* x, long x_row_stride, void* q, void* block_scales, long rows, int K, void* stream);
int cvcl_gemm_fp8(const void* A8, const float* a_scale, int lda, const void* W8, const float* w_scale, int ldw, void* C, int ldc,
                  const float* bias, int act, const void* R, int ldr, int M, int N, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVCL_HIP_H */
