/*
 * goat_hip.h — C ABI of libgoat_hip.so: hand-written gfx950 (CDNA4) kernels for GOAT's cross-modal
 * transformer forward/backward (VLN-GOAT pre-training / fine-tuning hot path).
 *
 * The reference (CrystalSixone/VLN-GOAT) is pure PyTorch: it has NO native interface for this path
 * (SURVEY.md §2.1).  Each entry point below therefore replaces a *PyTorch eager op sequence* of the
 * reference; the sequence is cited as file:line (P/ = pretrain_src/, M/ = map_nav_src/).  A maintainer
 * binds these with ctypes; `vln-goat_amd/_lib.py` is that binding.
 *
 * Conventions
 *   - plain C, raw device pointers, explicit sizes/strides (in ELEMENTS), no ownership transfer;
 *   - `stream` is a hipStream_t; every call is stream-ordered, never synchronises, never allocates
 *     (hipGraph-capturable);
 *   - dtype: GOAT_F32 = 0 (exact-f32 MFMA path), GOAT_BF16 = 1 (bf16 storage, f32 accumulate);
 *   - parameters (bias, LayerNorm gamma/beta) and statistics are always float32;
 *   - return 0 on success, negative on invalid argument / unsupported shape (GOAT_E_*), or the positive
 *     hipError_t of a failed launch.  No global state except lazily-set kernel attributes.
 */
#ifndef GOAT_HIP_H
#define GOAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOAT_F32 0
#define GOAT_BF16 1

#define GOAT_E_ARG (-1)    /* null pointer / bad enum */
#define GOAT_E_SHAPE (-2)  /* unsupported shape or alignment */

/* epilogues of goat_gemm_nt */
#define GOAT_EPI_NONE 0       /* C = A·Bᵀ (+bias) */
#define GOAT_EPI_GELU 1       /* u = A·Bᵀ+bias ; aux<-u (if aux) ; C = erf-gelu(u)   P/model/Bert_backbone.py:41-47,345-357 */
#define GOAT_EPI_RELU 2       /* same with relu                                   P/model/pretrain_goat.py:32-35 */
#define GOAT_EPI_MUL_DGELU 3  /* C = (A·Bᵀ) * gelu'(aux)   (backward of the GELU epilogue) */
#define GOAT_EPI_MUL_DRELU 4  /* C = (A·Bᵀ) * [aux>0] */
#define GOAT_EPI_ACCUM 5      /* C += A·Bᵀ  (float32 C only, no bias): weight gradients accumulated in place into the
                                flat gradient arena, i.e. autograd's `param.grad += dW` without the temporary */

/* library/version probe: returns 100*major+minor */
int goat_version(void);

/* C[M,N] = epilogue(A[M,K] · B[N,K]ᵀ + bias[N]).  Both operands K-contiguous ("NT"), which is
 * torch.nn.Linear's layout: replaces F.linear / addmm (P/model/Bert_backbone.py:170-172,302,348,362;
 * nn.MultiheadAttention in_proj/out_proj P/model/transformer.py:137; heads P/model/pretrain_goat.py:18-35).
 * Also used for dgrad (B = Wᵀ shadow) and wgrad (A = dYᵀ, B = Xᵀ, split_k>1, out f32).
 *   dtype_in : element type of A, B, aux        dtype_out: element type of C (GOAT_F32 allowed with bf16 in)
 *   K and lda/ldb must be multiples of 16 bytes worth of elements (8 bf16 / 4 f32); bases 16-B aligned.
 *   split_k>1 : K is split over gridDim.y and partial tiles are atomically added into C (requires
 *               dtype_out==GOAT_F32, epilogue NONE, bias NULL; C must hold the value to accumulate onto).
 */
int goat_gemm_nt(void* stream, int dtype_in, int dtype_out,
                 const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                 int M, int N, int K, const float* bias, int epilogue,
                 void* aux, int64_t ldaux, int split_k);

#define GOAT_GEMM_8WAVES 0x100   /* flag in goat_gemm_bf16's nstage argument: run the 128-row tile with eight waves */
#define GOAT_GEMM_PP 0x200       /* flag in nstage: the "ping-pong" main loop (csrc/gemm5_tile.hpp) — two groups of four waves half a
                                    K-tile out of phase, one feeding the matrix pipe while the other loads; tiles 256x256, 192x256 (K-contiguous
                                    A only), 128x256, 256x128, 128x128, nstage 2; same operands, epilogues and (bit-identical) results */
#define GOAT_GEMM_PERSIST 0x400  /* flag in nstage, with GOAT_GEMM_PP: one workgroup per CU walks the tiles of its XCD's chunk and requests the next
                                    tile's first K-tile before the current tile's epilogue (bf16 results, unsplit, K-contiguous A; bit-identical
                                    results; pays when the problem has more tiles than the chip has CUs, e.g. per-rank batch 256) */

/* Pipelined bf16 GEMM with direct-to-LDS (LDS-DMA) operand staging, all operand layouts (csrc/gemm2.hip):
 *   C[M,N] = epilogue( op(A) · op(B)ᵀ + bias ),  contraction length Kc
 *   trans_a=0: A is [M,Kc] (Kc contiguous) ; trans_a=1: A is [Kc,M] (M contiguous)   (same for B with N)
 *   (0,0) forward y = x·Wᵀ  (F.linear) ; (0,1) dgrad dx = dy·W ; (1,1) wgrad dW = dyᵀ·x  — the autograd of
 *   every nn.Linear on the path (P/model/Bert_backbone.py:170-172,302,348,362; P/model/transformer.py:137-140).
 * bf16 inputs; dtype_out GOAT_BF16 or GOAT_F32 (F32 only with GOAT_EPI_NONE); the activation-derivative epilogues take no bias.  K-contiguous operands need
 * Kc % 64 == 0 (transposed operands: any Kc, the tail is zero-filled by the buffer bounds check); lda/ldb
 * multiples of 8, bases 16-B aligned, each operand < 2 GiB.  split_k>1: f32 atomic accumulation into C.
 * bm: 64, 128 (four waves) or 256 (eight waves sharing one B tile: 25 % fewer L2->LDS bytes per flop, nstage <= 3; for
 * M >= 2048) — the M-tile; 64 fills the chip on small-M problems.  bm 128 with nstage | GOAT_GEMM_8WAVES: the 128-row tile
 * on eight waves (32x64 wave patches; twice the waves issue the tile's LDS-DMA).  nstage: 2..4 LDS ring stages (2 = most
 * workgroups per CU, 3-4 = deeper prefetch for long/cold contractions).
 * colsum (trans_a only, may be NULL): colsum[m] += sum_k A[k,m] (float32, atomic; caller zero-fills) — the bias
 * gradient of the Linear, accumulated from the A fragments the wgrad already holds in registers. */
int goat_gemm_bf16(void* stream, int trans_a, int trans_b, int dtype_out,
                   const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   int M, int N, int Kc, const float* bias, int epilogue,
                   void* aux, int64_t ldaux, int split_k, int bm, int nstage, float* colsum);
/* colsum[c] += sum_r x[r,c] (float32, atomic; caller zero-fills): bias gradient of a Linear. */
int goat_colsum(void* stream, int dtype, const void* x, int64_t ld, int R, int C, float* colsum);

/* Weight (and bias) gradient of a Linear with K <= 16 input features — the 7- / 14-wide position Linears
 * (P/model/vilmodel_goat.py:300-303,406,475): dw[n, k] += sum_r dy[r, n] x[r, k], dbias[n] += sum_r dy[r, n] (dbias may be NULL).
 * dy [rows, N] and x [rows, ld_x] in `dtype`, row strides in elements; x rows padded to whole 16-byte chunks and 16-byte aligned; dw float32 [N, K] with row stride ld_dw; both outputs are
 * ADDED to (float atomics: gradient-arena slices cleared at step start, or zero-filled by the caller). */
int goat_wgrad_smallk(void* stream, int dtype, const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int rows, int N, int K,
                      float* dw, int64_t ld_dw, float* dbias);

/* out[c, r] = in[r, c] for r<R, c<C; out columns R..ld_out-1 are zero-filled (so the result can feed
 * goat_gemm_nt as a K-padded operand).  If colsum!=NULL, colsum[c] += sum_r in[r,c] (float32, atomic):
 * that is the bias gradient of a Linear (autograd of P/model/Bert_backbone.py:302 et al.). */
int goat_transpose(void* stream, int dtype, const void* in, int64_t ld_in, void* out, int64_t ld_out,
                   int R, int C, float* colsum);

/* z = residual + dropout_p(x) ; y = LayerNorm(z)*gamma+beta.   BertSelfOutput/BertOutput
 * (P/model/Bert_backbone.py:306-310,366-370), RobertaEmbeddings LN (:115-116), pre-LN norms
 * (P/model/transformer.py:174,178).  residual may be NULL; p may be 0.  z_out may be NULL when
 * residual==NULL and p==0 (then z==x).  mean/rstd: float32[M] saved for backward.
 * Dropout keep-mask for flat element index i is hash(seed + *rng_dev, offset+i) (see csrc/common.hpp);
 * rng_dev (device uint64, may be NULL) lets a captured hipGraph draw fresh masks on every replay. */
int goat_ln_fwd(void* stream, int dtype, const void* x, const void* residual,
                const float* gamma, const float* beta, float eps,
                float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev,
                void* y, void* z_out, float* mean, float* rstd, int M, int H);

/* backward of goat_ln_fwd.  dz = LN-backward(dy + dy2) ; d_res<-dz (if non-NULL) ; dx<-dz*mask/(1-p) (if non-NULL).
 * dy2 (may be NULL): a second upstream gradient of y, summed on load.  In the post-LN blocks y feeds both the next
 * sub-layer's first Linear and the next LayerNorm's residual input; the two gradients arrive separately
 * (hipops.layer_norm(fork=True)) and autograd's add kernel between them is not needed.
 * dgamma/dbeta: float32[H], overwritten (accumulate=0) or added to (accumulate=1: gradient-arena slices).
 * ws == NULL (default): every block adds its column partials to dgamma / dbeta with float atomics — one launch, summation
 *   order not reproducible (accumulate=0 clears the two vectors with a memset node first).
 * ws != NULL: float32 scratch of goat_ln_bwd_ws_floats(H) elements for per-block column partials; a second tiny kernel
 *   reduces them — no atomics, deterministic (round-1 behaviour: 43 extra launches per pre-training step).
 * accumulate == 2 (ws required, goat_ln_bwd_nparts(M) * 2 * H floats): the per-block partials are LEFT in ws and nothing is
 *   written to dgamma / dbeta; the caller sums the partials of many LayerNorm calls into their gradient vectors with ONE
 *   goat_ln_reduce_batched launch when the backward pass ends (the column reduction is 40 % of this kernel's time at the GOAT
 *   row counts; deterministic).
 * dx_add (may be NULL; same shape / dtype as dx): added to dx on store — the gradient that reaches the LayerNorm's INPUT through
 *   its other consumer (the skip connection around a pre-LN sub-layer, P/model/transformer.py:170-182), so autograd launches no
 *   add kernel for that junction (hipops.layer_norm(fork_in=True)).
 *   accumulate | GOAT_LN_ADD_BEFORE: dx_add is instead the gradient that reaches the PRE-NORM SUM z = residual + dropout(x) through
 *   its other consumer and joins before the residual / dropout split: d_res <- dz + dx_add, dx <- (dz + dx_add) * mask / (1 - p).
 *   With it one goat_ln_fwd / goat_ln_bwd pair serves "src = skip + dropout(a); n = LayerNorm(src)" of a pre-LN block, where src
 *   itself continues as the next skip connection (hipops.layer_norm(z_out=True)).
 * Kernels: bf16 rows of H = 768 (every LayerNorm of the model at its hidden size) run ln_bwd768_kernel (round 6: 8-byte chunks, rows in
 *   flight packed as bf16, <= 128 VGPRs); other widths and float32 run the generic ln_bwd_kernel.  Same grid, same partial rows
 *   (goat_ln_bwd_nparts), same semantics; GOAT_LN_BWD_GENERIC=1 in the environment selects the generic kernel everywhere (A/B, tests). */
#define GOAT_LN_ADD_BEFORE 4
typedef struct goat_ln_partial {
  const float* ws;         /* partials written by goat_ln_bwd(..., accumulate = 2): [nparts][2][H] float32 */
  float* dgamma;           /* float32[H], ADDED to */
  float* dbeta;
  int32_t nparts;          /* goat_ln_bwd_nparts(M) of that call */
  int32_t reserved;
} goat_ln_partial;
int goat_ln_bwd_ws_floats(int H);
int goat_ln_bwd_nparts(int M);
/* entries with the same dgamma (a LayerNorm applied several times in one backward pass) must share dbeta; they are summed by one
 * writer in call order. */
int goat_ln_reduce_batched(void* stream, const goat_ln_partial* entries, int n, int H);
int goat_ln_bwd(void* stream, int dtype, const void* dy, const void* dy2, const void* z,
                const float* gamma, const float* mean, const float* rstd,
                float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev,
                void* dx, void* d_res, float* dgamma, float* dbeta, float* ws, int M, int H, int accumulate,
                const void* dx_add);
/* The same pair with dropout on the LayerNorm's OUTPUT as well: y = dropout_{p_out}(LayerNorm(z)) (the embedding blocks,
 * P/model/Bert_backbone.py:108-110, P/model/vilmodel_goat.py:228-237: `dropout(layer_norm(x))`), mask from counter offset_out of the
 * same seed; the backward masks dy (+ dy2) the same way before anything else.  post_add (may be NULL; shape / dtype of y): a second
 * summand behind the norm, y = dropout_{p_out}(LayerNorm(z) + post_add) — the image embedding block's normalised view features plus
 * normalised location features (P/model/vilmodel_goat.py:340-344); its gradient (the masked dy) is written to d_post.
 * p_out == 0 and post_add == NULL: exactly goat_ln_fwd / goat_ln_bwd. */
int goat_ln_fwd_do(void* stream, int dtype, const void* x, const void* residual, const float* gamma, const float* beta, float eps,
                   float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev, void* y, void* z_out, float* mean, float* rstd,
                   int M, int H, float p_out, uint64_t offset_out, const void* post_add);
int goat_ln_bwd_do(void* stream, int dtype, const void* dy, const void* dy2, const void* z, const float* gamma, const float* mean,
                   const float* rstd, float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev, void* dx, void* d_res,
                   float* dgamma, float* dbeta, float* ws, int M, int H, int accumulate, const void* dx_add, float p_out,
                   uint64_t offset_out, void* d_post);

/* y = residual + dropout_p(x)  (residual may be NULL, y may alias x).  nn.Dropout + pre-LN residual adds
 * (P/model/transformer.py:177,181; P/model/vilmodel_goat.py:316). n = element count. */
int goat_dropout_add_fwd(void* stream, int dtype, const void* x, const void* residual, void* y,
                         int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev);
/* dx = dy * mask/(1-p) with the same (seed, offset). */
int goat_dropout_bwd(void* stream, int dtype, const void* dy, void* dx,
                     int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev);

/* dx = dropmask_p(dy) * act'(u), act = GOAT_EPI_GELU | GOAT_EPI_RELU: backward of h = dropout_p(act(u)),
 * the inner activation (+dropout) of the panorama encoder FFN (P/model/transformer.py:179) and of the small heads. */
int goat_act_bwd(void* stream, int dtype, const void* dy, const void* u, void* dx, int64_t n, int act,
                 float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev);

/* Masked multi-head attention, head_dim fixed at 64.
 *   O[b,q,h,:] = dropout_p(softmax_k(scale * Q[b,q,h,:]·K[b,k,h,:] + kmask[b,k] + bias[b,q,k])) · V[b,k,h,:]
 * Replaces BertSelfAttention.forward (P/model/Bert_backbone.py:246-290; additive -10000 masks P/model/ops.py:25-34,
 * graph_sprels bias :690-691) and F.multi_head_attention_forward with key_padding_mask (-inf)
 * (P/model/transformer.py:172-176).  Q/K/V/O are [B, L, nh*64] views with explicit row and batch strides
 * (so q,k,v may be slices of one fused QKV projection, or — round 6 — K|V columns of a projection BANK that holds every layer's
 * cross-attention K|V of one attended sequence: row stride n_layers * 2 * nh * 64; hipops.linear_bank).  kmask: float32 [B,Lk] additive or NULL;
 * bias: float32 [B,Lq,Lk] additive or NULL.  lse: float32 [B,nh,Lq] saved for backward.
 * Lk <= 256.  Rows whose keys are all -inf produce zeros.
 * Dropout bits are a function of (seed + *rng_dev, offset, b, h, q, key) and of the dtype only — never of which kernel family
 * (LDS-staged / streaming) served the call — so goat_attn_bwd regenerates goat_attn_fwd's mask for every shape. */
int goat_attn_fwd(void* stream, int dtype,
                  const void* Q, int64_t q_rs, int64_t q_bs,
                  const void* K, int64_t k_rs, int64_t k_bs,
                  const void* V, int64_t v_rs, int64_t v_bs,
                  void* O, int64_t o_rs, int64_t o_bs,
                  const float* kmask, const float* bias, float* lse,
                  int B, int nh, int Lq, int Lk, float scale,
                  float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev);

/* backward of goat_attn_fwd.  dQ/dK/dV share the stride convention; dbias: float32 [B,Lq,Lk] or NULL,
 * ACCUMULATED (atomic; caller zero-fills) with the sum over heads of dS (the gradient of `bias`; feeds sprel_linear's 1->1 Linear,
 * P/model/vilmodel_goat.py:496-497). */
int goat_attn_bwd(void* stream, int dtype,
                  const void* Q, int64_t q_rs, int64_t q_bs,
                  const void* K, int64_t k_rs, int64_t k_bs,
                  const void* V, int64_t v_rs, int64_t v_bs,
                  const void* O, int64_t o_rs, int64_t o_bs,
                  const void* dO, int64_t do_rs, int64_t do_bs,
                  void* dQ, int64_t dq_rs, int64_t dq_bs,
                  void* dK, int64_t dk_rs, int64_t dk_bs,
                  void* dV, int64_t dv_rs, int64_t dv_bs,
                  const float* kmask, const float* bias, const float* lse, float* dbias,
                  int B, int nh, int Lq, int Lk, float scale,
                  float p, uint64_t seed, uint64_t offset, const uint64_t* rng_dev);

/* Softmax cross-entropy (reduction none) on float32 logits [M, ld] with N valid columns (ld >= N may be padded):
 * loss[m] = logsumexp(logits[m,:N]) - logits[m,target[m]], lse saved.  Replaces F.cross_entropy on the 576 x 50265
 * MLM scores (P/model/pretrain_goat.py:213-215).  Backward writes dlogits (GOAT_BF16 or GOAT_F32) with row stride
 * ld_out, zeros in the padding columns [N, ld_out), ready to be the K-padded operand of the decoder's dgrad/wgrad.
 * A negative target marks an ignored row (loss 0, zero gradient): the padding rows of a shape-bucketed static batch. */
int goat_ce_fwd(void* stream, const float* logits, int64_t ld, int M, int N, const int64_t* targets,
                float* loss, float* lse);
int goat_ce_bwd(void* stream, int dtype_out, const float* logits, int64_t ld, int M, int N,
                const int64_t* targets, const float* lse, const float* dloss, void* dlogits, int64_t ld_out);

/* Adaptive panorama fusion: w = softmax_v(tanh(x[n,v,:]·a + a0)) over ALL V slots (no mask);
 * fused[n,:] = sum_v w_v x[n,v,:]    (P/model/vilmodel_goat.py:354-361).  a: float32[H], a0: float32[1].
 * wsave: float32 [N,V] softmax weights saved for backward. */
int goat_pano_fusion_fwd(void* stream, int dtype, const void* x, const float* a, const float* a0,
                         void* fused, float* wsave, int N, int V, int H);
/* backward: dx (overwritten), da/da0 float32 accumulated atomically. */
int goat_pano_fusion_bwd(void* stream, int dtype, const void* x, const float* a, const float* a0,
                         const float* wsave, const void* dfused, void* dx, float* da, float* da0,
                         int N, int V, int H);

/* Row gather / segment mean:  out[i,:] = scale[i] * sum_{j in [start[i],start[i+1])} src[idx[j],:]
 * (idx = -1 contributes zero).  One launch replaces the host triple loop of
 * GlobalMapEncoder._aggregate_gmap_features (P/model/vilmodel_goat.py:438-460), the [stop]-token
 * concat/pad of vp_input_embedding (:377-391) and pad_tensors_wgrad (P/model/ops.py:46-68).
 * idx/start are int32 device arrays built once per batch on the host from the string ids.
 * tok_w (may be NULL): a weight per index entry, out[i] = scale[i] * sum_j tok_w[j] src[idx[j]].  With the INVERSE index of a
 * gather (graphmap.inverse_index: for every source row the segments that read it, tok_w = their scales) this same entry point is
 * the gather's backward pass — one writer per row, results in the activation dtype, no zero fill / atomics / cast. */
int goat_gather_segmean_fwd(void* stream, int dtype, const void* src, int64_t src_rows,
                            const int32_t* idx, const int32_t* start, const float* scale,
                            void* out, int n_out, int H, const float* tok_w);
/* backward: dsrc[idx[j],:] += scale[i]*dout[i,:]  (float atomics into float32 dsrc32 [src_rows,H]). */
int goat_gather_segmean_bwd(void* stream, int dtype, const void* dout,
                            const int32_t* idx, const int32_t* start, const float* scale,
                            float* dsrc32, int n_out, int H);

/* Grouped weight gradients: n <= 48 independent problems dW_i[n_out,n_in] (float32) = dY_i[rows,n_out]^T · X_i[rows,n_in]
 * (bf16 operands, any row count — the contraction tail is zero-filled) in ONE launch of the goat_gemm_bf16 tile
 * kernel, unsplit.  The autograd of the several nn.Linear of a transformer block (P/model/Bert_backbone.py:170-172,302,
 * 348,362) produces weight gradients of 36-144 tiles each; together they fill the 256 CUs without the split-K atomics
 * and zero fills a single small problem needs.  accumulate != 0: dW += ...; dbias (may be NULL): float32 [n_out],
 * += column sums of dY (atomic; the caller clears it).  ld_* in elements; ld_dy, ld_x multiples of 8, bases 16-B aligned.
 * bm 64 | 128, nstage 2..4 (| GOAT_GEMM_8WAVES with bm 128) as in goat_gemm_bf16. */
typedef struct goat_wgrad_problem {
  const void* dy; int64_t ld_dy;
  const void* x; int64_t ld_x;
  float* dw; int64_t ld_dw;
  float* dbias;
  int rows, n_out, n_in;
  int accumulate;
} goat_wgrad_problem;
int goat_wgrad_grouped(void* stream, const goat_wgrad_problem* probs, int n, int bm, int nstage);
/* Contraction-balanced form of the same launch (round 5).  goat_wgrad_grouped gives every output tile a workgroup, so a group is
 * whole rounds of tiles over the CUs plus a tail, and a group that mixes contraction lengths (panorama rows beside text rows) ends
 * with most CUs idle.  Here ONE workgroup per CU takes an equal, contiguous share of the group's K-tile iterations; a tile cut by a
 * share boundary is summed through `workspace` in a fixed order (deterministic; differs from goat_wgrad_grouped by float32
 * summation order only).  Ping-pong tiles only: bm = 256 | 256 << 16, 128 | 256 << 16, 256 | 128 << 16 (as goat_wgrad_grouped's bm).
 * workspace: goat_wgrad_balanced_ws_bytes(bm) bytes, 256-byte aligned, ZEROED once by the caller before the first launch (the kernel
 * leaves its flags zero), and not shared by launches that may run concurrently (one per stream).  GOAT_E_SHAPE when the group has
 * fewer K-tile iterations than the device has CUs (use goat_wgrad_grouped). */
int goat_wgrad_grouped_balanced(void* stream, const goat_wgrad_problem* probs, int n, int bm, void* workspace,
                                int64_t workspace_bytes);
int goat_wgrad_balanced_ws_bytes(int bm);   /* < 0: bm is not one of the tiles above */

/* Embedding tables (float32 masters):  out[r,:] = word[ids[r],:] + type[type_ids ? type_ids[r] : 0,:] + pos[r % L,:]
 * cast to `dtype` — the three nn.Embedding lookups and two adds of BertEmbeddings.forward
 * (P/model/Bert_backbone.py:98-113; the reference's position ids are arange(L) for every sample), and with
 * type_tab = pos_tab = NULL the single lookup of gmap_step_embeddings (P/model/vilmodel_goat.py:474) /
 * nav_type_embedding.  ids/type_ids are int64 [rows] (torch LongTensor).  An id outside [0,vocab) sets bit 0 of
 * *err_flag (if given) and reads row 0 instead of faulting. */
int goat_embed_fwd(void* stream, int dtype, const float* word, const int64_t* ids, const float* type_tab,
                   const int64_t* type_ids, const float* pos_tab, int L, void* out, int rows, int H, int vocab,
                   int* err_flag);
/* backward: scatter-add of dout[rows,H] into the PRE-ZEROED float32 table gradients (float atomics).  Any of
 * dword / dtype_tab / dpos may be NULL.  Rows with ids[r] == word_pad and positions l == pos_pad contribute
 * nothing (nn.Embedding(padding_idx) semantics, Bert_backbone.py:85-87; pass -1 for "no padding row").
 * dtype_tab is only written when type_ids != NULL: the all-zero-type case is a column sum (goat_colsum). */
int goat_embed_bwd(void* stream, int dtype, const void* dout, const int64_t* ids, const int64_t* type_ids, int L,
                   float* dword, float* dtype_tab, float* dpos, int rows, int H, int vocab, int word_pad, int pos_pad);

/* ---- causal-learning heads (csrc/causal.hip) -------------------------------------------------------------------
 * tanh-attention pooling of the CFP heads: a = softmax_l(tanh(x[b,l,:])·w) over ALL L <= 256 slots (no padding mask,
 * as the reference), out[b,:] = tanh(sum_l a_l x[b,l,:])   (P/model/pretrain_goat.py:502-515,
 * M/models/vilmodel_GOAT.py:909-922).  x [B,L,H] in `dtype`; w float32 [H]; out float32 [B,H]; attn float32 [B,L] (saved).
 * slot_mask (float32 [B,L], may be NULL): added to the scores before the softmax — 0 / -inf.  NULL = the reference: every slot
 * of the batch's padded width takes part.  A shape-bucketed static batch is padded BEYOND that width; the mask removes exactly
 * those extra slots, so the pooled vector is what the reference computes on the batch's own padding. */
int goat_attn_pool_fwd(void* stream, int dtype, const void* x, const float* w, float* out, float* attn, float* ws, int B,
                       int L, int H, const float* slot_mask);   /* ws: float32 scratch of B*L elements */
/* backward: dx [B,L,H] (dtype) overwritten; dw float32 [H] accumulated (atomics; caller zero-fills); ws: float32 scratch
 * of B*L elements.  H % 4 == 0. */
int goat_attn_pool_bwd(void* stream, int dtype, const void* x, const float* w, const float* attn, const float* out,
                       const float* dout, void* dx, float* dw, float* ws, int B, int L, int H);

/* "door" gate of BACL type_2 / FACL: s = sigmoid(aug·wa + ba + ori·wo + bo) per row, out = s*aug + (1-s)*ori
 * (P/model/vilmodel_goat.py:137-143; M/models/vilmodel_GOAT.py:147-153, 548-552: two nn.Linear(H,1) + nn.Sigmoid).
 * aug/ori/out [rows,H] in `dtype`; wa/wo float32 [H]; ba/bo float32 device scalars; gate float32 [rows] (saved). */
int goat_door_gate_fwd(void* stream, int dtype, const void* aug, const void* ori, const float* wa, const float* wo,
                       const float* ba, const float* bo, void* out, float* gate, int rows, int H);
/* backward: daug/dori overwritten; dwa/dwo float32 [H] and dbias float32 [1] accumulated (caller zero-fills);
 * dbias is the gradient of BOTH biases; dbias2 (may be NULL) receives the same sum — the second bias' own gradient slice.  H <= 1024. */
int goat_door_gate_bwd(void* stream, int dtype, const void* aug, const void* ori, const float* wa, const float* wo,
                       const float* gate, const void* dout, void* daug, void* dori, float* dwa, float* dwo, float* dbias,
                       int rows, int H, float* dbias2);

/* probability-weighted dictionary sum of BACL type_1: out[b,:] = sum_k p[b,k] z[b,k,:]
 * (P/model/vilmodel_goat.py:115-118; M/models/vilmodel_GOAT.py:246-249).  z float32 [B,K,H], p float32 [B,K];
 * out in dtype_out.  backward: dz = p (x) dout, dp = z·dout; either may be NULL. */
int goat_dict_wsum_fwd(void* stream, int dtype_out, const float* z, const float* p, void* out, int B, int K, int H);
int goat_dict_wsum_bwd(void* stream, int dtype_dout, const void* dout, const float* z, const float* p, float* dz, float* dp,
                       int B, int K, int H);

/* ---- fused optimizer step on the gradient arena ------------------------------------------------------------
 * The reference's update (P/train_r2r_goat.py:349-366): grad-norm clip 5.0 over every parameter that has a gradient
 * (torch.nn.utils.clip_grad_norm_), then HF-style AdamW (P/optim/adamw.py:53-110): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= step_size * m / (sqrt(v) + eps) with step_size = lr * sqrt(1-b2^t) / (1-b1^t) (t = that PARAMETER's own step count);
 * then the decoupled decay applied AFTER the update, p -= lr * weight_decay * p (weight_decay 0 for names containing
 * "bias" / "LayerNorm.weight", P/optim/misc.py:13-23); parameters without a gradient in this step are skipped entirely.
 * Here: gradients live in ONE float32 arena (dp.GradArena), the moments in two arenas of the same layout, the parameters
 * stay the model's own float32 tensors.
 *   goat_grad_sqnorm : *out_sq += sum of g^2 over the element ranges [begin, end) of `ranges` (int64 pairs, device memory)
 *   goat_adamw_step  : one launch over `nchunks` chunks (<= 65536 elements each; chunk c = {tensor index, first element}
 *                      int32 pairs in device memory) of the tensors described by `tensors` (device memory).  The clip
 *                      coefficient min(1, max_norm / (sqrt(*sq_norm) + 1e-6)) is computed in the kernel from the device
 *                      scalar goat_grad_sqnorm produced (max_norm <= 0: no clipping), so the step needs no host
 *                      synchronisation.  shadow0 / shadow1 / shadow_f32 (optional): the copies of the parameter the GEMMs read
 *                      (bf16 operand "shadows", the float32 image inside a concatenated QKV bias) refreshed in the same
 *                      pass AT THEIR ADDRESSES — a captured hipGraph that reads them sees the updated weights. */
typedef struct goat_adamw_tensor {
  float* param;            /* float32 master weights */
  void* shadow0;           /* bf16 copy, same element order, or NULL */
  void* shadow1;           /* second bf16 copy (e.g. inside a row-concatenated QKV shadow), or NULL */
  int64_t arena_off;       /* first element of this tensor in the gradient / moment arenas */
  int64_t numel;
  float step_size;         /* lr * sqrt(1 - b2^t) / (1 - b1^t) for this tensor's step count t (or lr without bias correction) */
  float decay;             /* lr * weight_decay (0: no decay) */
  float* shadow_f32;       /* float32 copy, same element order (a member of a concatenated bias), or NULL */
  int32_t cols;            /* > 0: shadow0 is a row-padded image — element i goes to (i / cols) * ld0 + i % cols */
  int32_t ld0;             /*      (the K-padded bf16 copies of the 7- / 14-wide position Linears) */
} goat_adamw_tensor;
int goat_grad_sqnorm(void* stream, const float* arena, const int64_t* ranges, int n_ranges, float* out_sq);
int goat_adamw_step(void* stream, const float* grad_arena, float* exp_avg, float* exp_avg_sq, const goat_adamw_tensor* tensors,
                    const int32_t* chunks, int nchunks, float beta1, float beta2, float eps, float max_norm,
                    const float* sq_norm);

/* Tail of the single-action-prediction head, one launch per direction (csrc/causal.hip).  gs [B,G] / ls [B,W]: raw scores of the
 * global / local heads (dtype GOAT_BF16 | GOAT_F32); fwl [B]: fusion logit (fw = sigmoid(fwl) when fw_sigmoid, fwl itself otherwise;
 * NULL: fw = 0.5).  Masks (bytes, any may be NULL): gvis 1 = visited, gvalid 0 = beyond the map, glens map lengths, lmask 1 = masked
 * (1 = valid when lmask_is_valid).  M [B,G,W] float32 logit-fusion matrix (NULL: none).  Outputs float32:
 *   gl = mask(gs * fw), ll = mask(ls * (1 - fw)), fused = gl + M · zero-filled(ll) (+ ll[0] on column 0 when add_stop),
 *   loss[b] = CE(gl, ga) + CE(ll, la) + CE(fused, ga) when loss != NULL (negative label: 0), lse [B,3] saved for the backward.
 * Replaces the reference's elementwise / masked_fill / bmm / log_softmax chain: P/model/pretrain_goat.py:375-413 (pre-training),
 * M/models/vilmodel_GOAT.py:803-839 (fine-tuning: add_stop = 1, lmask = vp_nav_masks with lmask_is_valid = 1).
 * Backward: upstream dloss [B] (with the labels) and / or dgl, dll, dfused (NULL = zero) -> dgs, dls (dtype), dfwl (when fwl). */
int goat_sap_fuse_fwd(void* stream, int dtype, const void* gs, const void* ls, const void* fwl, int fw_sigmoid,
                      const uint8_t* gvis, const uint8_t* gvalid, const int64_t* glens, const uint8_t* lmask, int lmask_is_valid,
                      const float* M, int add_stop, const int64_t* ga, const int64_t* la, float* gl, float* ll, float* fused,
                      float* loss, float* lse, int B, int G, int W);
int goat_sap_fuse_bwd(void* stream, int dtype, const void* gs, const void* ls, const void* fwl, int fw_sigmoid,
                      const uint8_t* gvis, const uint8_t* gvalid, const int64_t* glens, const uint8_t* lmask, int lmask_is_valid,
                      const float* M, int add_stop, const int64_t* ga, const int64_t* la, const float* gl, const float* ll,
                      const float* fused, const float* lse, const float* dloss, const float* dgl, const float* dll,
                      const float* dfused, void* dgs, void* dls, void* dfwl, int B, int G, int W);
/* CFP contrastive losses (P/model/pretrain_goat.py:519-534): loss[i] = sum over x in {gmap, vp, fused} of
 * 1/2 [CE(x_loc[i]·txt_allᵀ/τ, t_i) + CE(txt_loc[i]·x_allᵀ/τ, t_i)], t_i = target0 + i.  x_loc / x_all: arrays of 3 device
 * pointers ([Bl,H] / [Ba,H] float32; all = loc on one rank, the all-gathered rows under data parallelism); loss [Bl] is
 * ADDED to (one writer per sample: deterministic); prob: [6,Bl,Ba] float32 scratch the backward reads.
 * Backward: gradients are ADDED to dx_loc[k] / dx_all[k] / dtxt_loc / dtxt_all (any may be NULL; loc and all pointers may
 * alias on one rank).  Every output element has one writer that sums its contributions in a fixed order — no atomics, so a
 * captured or phased step reproduces the eager one bit for bit.  Replaces ~70 ATen launches of the torch formulation. */
int goat_infonce_fwd(void* stream, const float* const* x_loc, const float* const* x_all, const float* txt_loc,
                     const float* txt_all, float* loss, float* prob, int Bl, int Ba, int H, int target0, float temperature);
int goat_infonce_bwd(void* stream, const float* const* x_loc, const float* const* x_all, const float* txt_loc,
                     const float* txt_all, const float* dloss, const float* prob, float* const* dx_loc, float* const* dx_all,
                     float* dtxt_loc, float* dtxt_all, int Bl, int Ba, int H, int target0, float temperature);

/* Debug/probe helper used by tests: fills out[64*4] with the element indices returned by
 * ds_read_b64_tr_b16 when lane l points at elements 4l..4l+3 of an LDS array holding 0,1,2,... */
int goat_probe_tr16(void* stream, uint16_t* out);

/* Linear(H, 1) — the last layer of ClsPrediction (P/model/pretrain_goat.py:27-38: the global / local action scores, the fusion logit, the
 * object scores): y[m] = x[m,:] . w + b (w: float32 [H], rounded to the activation dtype as the GEMM path's shadow weight; b: float32 [1] or
 * NULL; float32 accumulation; y in `dtype`).  H a multiple of 8, <= 1024. */
int goat_rowdot_fwd(void* stream, int dtype, const void* x, const float* w, const float* b, void* y, int M, int H);
/* backward: dx[m,:] = dy[m] w (NULL: skipped); dw[H] += sum_m dy[m] x[m,:], db[1] += sum_m dy[m] (float32 atomics: the caller clears or
 * accumulates; NULL dw: both skipped). */
int goat_rowdot_bwd(void* stream, int dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db, int M, int H);

/* CFP fused vector (P/model/pretrain_goat.py:486-499 with the glocal fusion weight of :393-399): w = sigmoid(fwl[b]) (fwl: the
 * output of sap_fuse_linear, [B] in dtype_fwl); fo[b,:] = go[b,:] * w + vo[b,:] * (1 - w) (float32 [B,H]); fw[b] = w saved for backward.
 * Replaces sigmoid, two muls, rsub, add (and their ~10 backward launches) on [B,768] tensors. */
int goat_cfp_mix_fwd(void* stream, int dtype_fwl, const float* go, const float* vo, const void* fwl, float* fo, float* fw, int B, int H);
/* backward: dgo = dfo * w, dvo = dfo * (1 - w) (accumulate != 0: ADDED to what dgo / dvo hold — the InfoNCE gradients of the same vectors),
 * dfwl[b] = w (1 - w) sum_h dfo (go - vo) in dtype_fwl. */
int goat_cfp_mix_bwd(void* stream, int dtype_fwl, const float* go, const float* vo, const float* fw, const float* dfo, float* dgo,
                     float* dvo, void* dfwl, int B, int H, int accumulate);

/* ---- glue of the captured steps (csrc/glue.hip) -------------------------------------------------------------------------
 * out[numel] = sum_i srcs[i][numel] (n <= 8 tensors of `dtype`, float32 accumulation, ONE rounding): the gradient of a tensor with
 * several consumers — replaces the k - 1 pairwise `add` launches of torch's autograd engine (AccumulateGrad-free fan-in: the text
 * states read by every cross-modal layer, P/model/vilmodel_goat.py:163-199; LayerNorm outputs with two consumers,
 * P/model/Bert_backbone.py:304-310).  srcs is a HOST array of device pointers (16-B aligned); out may alias srcs[0]. */
int goat_add_n(void* stream, int dtype, const void* const* srcs, int n, void* out, int64_t numel);
/* n <= 16 byte ranges (16-B aligned, multiples of 16 bytes) cleared by one launch — optimizer.zero_grad() of the reference
 * (P/train_r2r_goat.py:301-363) over the gradient arena's per-task ranges.  ptrs / nbytes are HOST arrays. */
int goat_zero_ranges(void* stream, void* const* ptrs, const int64_t* nbytes, int n);

#ifdef __cplusplus
}
#endif
#endif /* GOAT_HIP_H */
