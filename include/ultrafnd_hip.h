/* ultrafnd_hip.h -- C ABI of libultrafnd_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the text+vision fusion train-step hot path of
 * Nuralamsiddik16/Ultrafnd_git.  The reference has no FFI of its
 * own (it is pure Python on torch CPU/MPS); each entry point below names the reference
 * Python interface it replaces (file:line under the reference root).  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host;
 *   - row-major, fp32 unless a name says bf16 (bf16 = raw uint16 storage);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     allocates, frees or synchronises (safe inside hipGraph capture);
 *   - return 0 on success, a UFND_ERR_* code otherwise; ufnd_last_error() gives the text;
 *   - buffers are caller-owned; nothing is retained across calls.
 */
#ifndef ULTRAFND_HIP_H
#define ULTRAFND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UFND_OK 0
#define UFND_ERR_INVALID 1 /* bad argument: shape, alignment, null pointer */
#define UFND_ERR_LAUNCH 2  /* HIP launch error */

#define UFND_ABI_VERSION 5

const char* ufnd_last_error(void);
int ufnd_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Device-resident step state (graph-replay safe: hyper-parameters and counters are read
 * from memory, never baked into launches).  Host initialises it with hipMemcpy.
 * ---------------------------------------------------------------------------------- */
typedef struct ufnd_step_state {
  uint64_t step;       /* optimizer steps taken so far (AdamW's t-1); ++ by ufnd_adamw_step */
  uint64_t seed;       /* dropout key */
  float lr;            /* current learning rate (StepLR writes it between epochs) */
  float weight_decay;
  float beta1, beta2, eps;
  float max_norm;      /* clip_grad_norm_ max_norm; <= 0 disables clipping */
  float grad_scale;    /* multiplied into every gradient before use (1/world for summed DP grads) */
  float loss;          /* out: mean CE loss of the last ufnd_softmax_ce */
  float grad_norm;     /* out: global L2 norm (after grad_scale, before clipping) */
  float clip_coef;     /* out: min(1, max_norm / (grad_norm + 1e-6)) */
  float bc1, bc2_sqrt; /* out: 1-beta1^t, sqrt(1-beta2^t) for the step being applied */
  float reserved[3];
} ufnd_step_state;

/* ------------------------------------------------------------------------------------
 * Tier A geometry (configs/model_configs/fusion.yaml:2-7, classifier.yaml:2-17)
 * ---------------------------------------------------------------------------------- */
typedef struct ufnd_dims {
  int hidden;       /* 512; must be a multiple of 256 */
  int text_dim;     /* 768 */
  int audio_dim;    /* 128 */
  int visual_dim;   /* 512 */
  int temporal_dim; /* 256 */
  int gnn_dim;      /* 128 */
  int aux_dim;      /* 2 (0 = classifier without aux) */
  int trees;        /* 6 */
  int depth;        /* 4 (<= 6) */
  int classes;      /* 2 (only 2 is supported) */
  float fusion_dropout; /* 0.1 */
  float clf_dropout;    /* 0.1 */
  float node_dropout;   /* 0.3 */
} ufnd_dims;

/* Parameter (or gradient) pointer table.  The same struct type carries the gradients.
 * Names follow the reference's state_dict.  Three groups must be
 * CONTIGUOUS in memory because the kernels treat them as one stacked matrix:
 *   qkv_w : rows [attn_tv.q, attn_ta.q, attn_tv.k, attn_tv.v, attn_vu.q,
 *                 attn_ta.k, attn_ta.v, attn_vu.k, attn_vu.v] each (hidden x hidden)
 *   qkv_b : the nine biases in the same order
 *   gates : node.trees.{t}.gates.{k} for t-major, k-minor, each (hidden)
 *   thresh: node.trees.{t}.thresh.{k}, same order, one float each
 *   leaf  : node.trees.{t}.leaf_logits, t-major, each (2^depth x classes)
 *   tau   : node.trees.{t}.tau, one float each
 */
typedef struct ufnd_fusion_params {
  float *text_w, *text_b;         /* text_proj      (hidden x text_dim)   cross_modal_transformer.py:96 */
  float *audio_w, *audio_b;       /* audio_proj                                                    :97 */
  float *visual_w, *visual_b;     /* visual_proj                                                   :98 */
  float *temporal_w, *temporal_b; /* temporal_proj                                                 :99 */
  float *gnn_w, *gnn_b;           /* gnn_proj                                                      :102 */
  float *qkv_w, *qkv_b;           /* stacked co-attention projections                              :31-33 */
  float *ev0_w[3], *ev0_b[3];     /* attn_{tv,ta,vu}.evidence_proj.0 (hidden x 3), (hidden)        :34-38 */
  float *ev2_w[3], *ev2_b[3];     /* attn_{tv,ta,vu}.evidence_proj.2 (1 x hidden), (1)                    */
  float *fuse0_w, *fuse0_b;       /* fuse_mlp.0 (2*hidden x 16*hidden)                             :122 */
  float *fuse3_w, *fuse3_b;       /* fuse_mlp.3 (hidden x 2*hidden)                                :125 */
  float *cls_w, *cls_b;           /* classifier (2 x hidden) -- aux head, no grad under the trainer :130 */
} ufnd_fusion_params;

typedef struct ufnd_clf_params {
  float *pre0_w, *pre0_b;     /* pre.0 (hidden x (hidden+aux_dim))   deep_truth_classifier.py:122 */
  float *pre3_w, *pre3_b;     /* pre.3 (hidden x hidden)                                     :125 */
  float *gates, *thresh;      /* NODE gates / thresholds                                     :44-45 */
  float *leaf, *tau;          /* leaf logits, temperature tau per tree                       :41,49 */
  float *bypass_w, *bypass_b; /* bypass (classes x hidden)                                   :137 */
  float *temperature;         /* scalar                                                      :115 */
} ufnd_clf_params;

/* ------------------------------------------------------------------------------------
 * Workspaces.  Sizes depend on the batch size only; the caller allocates `floats`
 * fp32 elements (256-byte aligned) and passes the base pointer to every call of the
 * same batch size.  Forward leaves its saved activations there for backward.
 * ---------------------------------------------------------------------------------- */
size_t ufnd_fusion_workspace_floats(const ufnd_dims* d, int B);
size_t ufnd_clf_workspace_floats(const ufnd_dims* d, int B);

/* ------------------------------------------------------------------------------------
 * CrossModalTransformer.forward                 src/models/fusion/cross_modal_transformer.py:134-210
 *   in : text (B,text_dim) audio (B,audio_dim) visual (B,visual_dim) temporal (B,temporal_dim)
 *        gnn (B,gnn_dim); contiguous rows.  gnn must be given (the reference's fuse_mlp is
 *        built for the 16*hidden concat and rejects a missing gnn_feat, :184-195).
 *   out: fused (B,hidden) with row stride ld_fused (multiple of 4), logits (B,2) or NULL to
 *        skip the aux head, forensic (3,B) = emotion_intensity, semantic_conflict, temporal_delay.
 *   train != 0 applies dropout with the mask keyed by state->{seed,step}.
 * ---------------------------------------------------------------------------------- */
int ufnd_fusion_forward(const ufnd_dims* d, const ufnd_fusion_params* p, const float* text, const float* audio,
                        const float* visual, const float* temporal, const float* gnn, int B, int train,
                        float* workspace, float* fused, int ld_fused, float* logits, float* forensic,
                        const ufnd_step_state* state, void* stream);

/* The two backward entry points take an optional `side_stream` (NULL = everything on `stream`): the
 * dX chain -- the critical path -- stays on `stream`, every dW / parameter-gradient kernel goes to
 * `side_stream`, forked with events as soon as its inputs exist.  `join` != 0 makes `stream` wait for
 * the side work before the call returns control of the stream (pass 0 to keep the overlap going into
 * the next call and join there).  Event record/wait pairs are capturable into a hipGraph.
 *
 * autograd backward of the above (what loss.backward() runs, forensic_trainer.py:291).
 *   d_fused (B,hidden) stride ld_dfused; d_logits (B,2) or NULL.  Writes EVERY gradient in
 *   `g` that can receive one (cls_w/cls_b only when d_logits != NULL; overwritten, not
 *   accumulated).  Inputs get no gradient (they are data). */
int ufnd_fusion_backward(const ufnd_dims* d, const ufnd_fusion_params* p, const ufnd_fusion_params* g,
                         const float* text, const float* audio, const float* visual, const float* temporal,
                         const float* gnn, int B, int train, float* workspace, const float* d_fused,
                         int ld_dfused, const float* d_logits, const ufnd_step_state* state, void* stream,
                         void* side_stream, int join);

/* The same backward in two phases, for a gradient exchange that overlaps backward (data parallel; the reference
 * step is single-process, forensic_trainer.py:285-298).  The flat gradient arena is laid out in gradient-ready
 * order [classifier | fuse_mlp | co-attention | projections]:
 *   UFND_BWD_FUSE_MLP  d_fused -> fuse_mlp.3 -> fuse_mlp.0: their dW / db (incl. the 33.5 MB fuse_mlp.0.weight
 *                      gradient, 2/3 of all gradient bytes) are complete when this call's work has run -- the
 *                      caller starts all-reducing [classifier | fuse_mlp] while
 *   UFND_BWD_REST      computes everything else (co-attention, stacked q/k/v, projections).
 * FUSE_MLP followed by REST writes bit-identical gradients to UFND_BWD_ALL (= ufnd_fusion_backward). */
#define UFND_BWD_ALL 0
#define UFND_BWD_FUSE_MLP 1
#define UFND_BWD_REST 2
/* OR-ed into `phase` (and the `flags` of ufnd_classifier_backward_ex): run the backward WITHOUT the Linear layers' dW / db
 * products -- every other gradient and the whole dX chain as usual.  The caller forms those products from gathered
 * factors (ufnd_head_linear_grads_from_factors below); until then the Linear entries of `g` hold stale values. */
#define UFND_BWD_NO_LINEAR_GRADS 16
int ufnd_fusion_backward_phase(const ufnd_dims* d, const ufnd_fusion_params* p, const ufnd_fusion_params* g,
                               const float* text, const float* audio, const float* visual, const float* temporal,
                               const float* gnn, int B, int train, float* workspace, const float* d_fused,
                               int ld_dfused, const float* d_logits, const ufnd_step_state* state, void* stream,
                               void* side_stream, int join, int phase);

/* ------------------------------------------------------------------------------------
 * DeepTruthClassifier.forward                   src/models/fusion/deep_truth_classifier.py:148-171
 *   fused (B,hidden) stride ld_fused; aux (B,aux_dim) or NULL when aux_dim == 0.
 *   out: logits (B,2), probs (B,2) = softmax(logits / clamp(temperature,0.5,5)).
 *   If `fused` already points at the workspace's input panel (ufnd_clf_input_panel) the
 *   copy is skipped -- the fused train step lets the fusion write there directly.
 * ---------------------------------------------------------------------------------- */
float* ufnd_clf_input_panel(const ufnd_dims* d, float* clf_workspace, int B, int* ld);
int ufnd_classifier_forward(const ufnd_dims* d, const ufnd_clf_params* p, const float* fused, int ld_fused,
                            const float* aux, int B, int train, float* workspace, float* logits, float* probs,
                            const ufnd_step_state* state, void* stream);
/* backward for a gradient arriving at `logits` (the trainer's CE-on-logits loss; temperature
 * and tau get no gradient, deep_truth_classifier.py:41,164-170).  d_fused (B,hidden) stride
 * ld_dfused is written. */
int ufnd_classifier_backward(const ufnd_dims* d, const ufnd_clf_params* p, const ufnd_clf_params* g, int B,
                             int train, float* workspace, const float* d_logits, float* d_fused, int ld_dfused,
                             const ufnd_step_state* state, void* stream, void* side_stream, int join);

int ufnd_classifier_backward_ex(const ufnd_dims* d, const ufnd_clf_params* p, const ufnd_clf_params* g, int B,
                                int train, float* workspace, const float* d_logits, float* d_fused, int ld_dfused,
                                const ufnd_step_state* state, void* stream, void* side_stream, int join, int flags);

/* ------------------------------------------------------------------------------------
 * Factor form of the head's Linear gradients, for a data-parallel gradient exchange that moves FACTORS instead of
 * gradients (the reference step is single-process: forensic_trainer.py:285-298; loss.backward() at :291 is what is
 * being distributed).  All 13 Linear layers of the head (fuse_mlp.0/.3, the nine stacked q/k/v, the five input
 * projections, pre.0/.3: 50.9 of the 51.0 MB of gradient) have dW = dY^T X and db = column sums of dY over the batch
 * rows, so the sum over ranks is the same product over ALL ranks' rows:
 *   1. backward with UFND_BWD_NO_LINEAR_GRADS (both modules);
 *   2. ufnd_head_pack_factors: the rank's dY / X panels out of the two workspaces and the step's inputs into ONE
 *      contiguous pack of ufnd_head_factor_floats(d, B) floats (one launch; 2.5 MB at B = 32, hidden 512);
 *   3. all-gather the packs (rank r's pack at packs + r * rank_stride);
 *   4. ufnd_head_linear_grads_from_factors: the SUMMED dW / db of every Linear over ranks * B rows, written into the
 *      gradient tables in one grouped launch -- two from 128 gathered rows on when (hidden + aux_dim) % 4 != 0 -- (fp32 MFMA; rows in rank order, so every rank computes the same bits).
 * The remaining gradients (gates, thresholds, leaves, bypass, evidence_proj: 21 k floats) are summed by an ordinary
 * all-reduce.  ranks == 1 reproduces the plain backward's dW / db bit for bit.
 * ---------------------------------------------------------------------------------- */
size_t ufnd_head_factor_floats(const ufnd_dims* d, int B);
int ufnd_head_pack_factors(const ufnd_dims* d, const float* text, const float* audio, const float* visual, const float* temporal,
                           const float* gnn, int B, float* fusion_workspace, float* clf_workspace, float* pack, void* stream);
int ufnd_head_linear_grads_from_factors(const ufnd_dims* d, const ufnd_fusion_params* fusion_grads, const ufnd_clf_params* clf_grads,
                                        const float* packs, size_t rank_stride, int ranks, int B, void* stream);

/* ------------------------------------------------------------------------------------
 * The head's train step over BOTH modules in two calls (round 4): forensic_trainer.py:285-291's
 *   logits, _ = model(...); loss = F.cross_entropy(logits, y); loss.backward()
 * with the same kernels and arithmetic as ufnd_fusion_forward -> ufnd_classifier_forward -> ufnd_softmax_ce ->
 * ufnd_classifier_backward -> ufnd_fusion_backward_phase, minus the launches that exist only because those are five calls
 * (classifier input preparation, the CE kernel, the activation backward between the modules, one of the two parameter-gradient
 * launches, the classifier's own weight-gradient launch): 26 -> 21 launches per step at B = 32.  Logits, probabilities, forensic scalars, state->loss, d_logits and every
 * gradient are bit-identical to the five-call sequence.  The fusion writes `fused` straight into the classifier's input panel;
 * the aux head (fusion logits) is not evaluated.  Plain mean CE only (the weighted / label-smoothed criterion keeps the
 * five-call sequence).  `phase` as in ufnd_fusion_backward_phase (UFND_BWD_FUSE_MLP includes the classifier's backward).
 * ---------------------------------------------------------------------------------- */
typedef struct ufnd_head_io {
  const float *text, *audio, *visual, *temporal, *gnn, *aux;   /* the step's inputs (gnn NULL iff dims.gnn_dim == 0, aux NULL iff aux_dim == 0) */
  const int64_t* labels;                                        /* (B) */
  float *fusion_workspace, *clf_workspace;                      /* ufnd_fusion_workspace_floats / ufnd_clf_workspace_floats */
  float *logits, *probs, *forensic;                             /* (B,2), (B,2), (3,B) */
  float* d_logits;                                              /* (B,2): written by the forward, read by the backward */
} ufnd_head_io;
int ufnd_head_forward_loss(const ufnd_dims* d, const ufnd_fusion_params* fusion, const ufnd_clf_params* clf, const ufnd_head_io* io, int B,
                           int train, ufnd_step_state* state, void* stream);
int ufnd_head_backward(const ufnd_dims* d, const ufnd_fusion_params* fusion, const ufnd_fusion_params* fusion_grads, const ufnd_clf_params* clf,
                       const ufnd_clf_params* clf_grads, const ufnd_head_io* io, int B, int train, ufnd_step_state* state, void* stream,
                       void* side_stream, int join, int phase);

/* F.cross_entropy(logits, y), mean reduction, + its gradient (forensic_trainer.py:287).
 * labels int64 (B).  loss_rows (B) or NULL; d_logits (B,2) = (softmax - onehot)/B or NULL.
 * state->loss receives the mean (summed in a fixed order: bit-reproducible). */
int ufnd_softmax_ce(const float* logits, const int64_t* labels, int B, float* loss_rows, float* d_logits,
                    ufnd_step_state* state, void* stream);

/* nn.CrossEntropyLoss(weight=(w0, w1), label_smoothing=eps), mean reduction (normalised by the sum of the
 * target-class weights), + its gradient: the criterion of the integrated trainer variant
 * (src/training/forensic_trainer_integrated.py:154-166).  eps = 0, w = (1, 1) is ufnd_softmax_ce. */
int ufnd_softmax_ce_weighted(const float* logits, const int64_t* labels, int B, float w0, float w1, float label_smoothing,
                             float* loss_rows, float* d_logits, ufnd_step_state* state, void* stream);

/* ------------------------------------------------------------------------------------
 * clip_grad_norm_ + AdamW.step over a flat fp32 arena    forensic_trainer.py:292-298,176
 *   grad/param/exp_avg/exp_avg_sq: n floats each (n % 4 == 0, 16-byte aligned).
 *   ufnd_grad_norm : state->grad_norm = ||grad * grad_scale||_2, clip_coef, bias corrections
 *                    for step t = state->step + 1.  partials: >= 1024 floats scratch.
 *   ufnd_adamw_step: p *= 1 - lr*wd; m,v update with g = grad*grad_scale*clip_coef; p -= ...
 *   ufnd_step_advance: state->step += 1 (once per optimizer step, after every arena is done).
 * ---------------------------------------------------------------------------------- */
int ufnd_grad_norm(const float* grad, size_t n, float* partials, ufnd_step_state* state, void* stream);
int ufnd_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    const ufnd_step_state* state, void* stream);
int ufnd_step_advance(ufnd_step_state* state, void* stream);
/* The three calls above (four launches) as ONE call of two launches, for a single arena: the sum of squares (its first
 * block also advances state->step), then AdamW, in which every block re-derives the norm, the clip coefficient and the
 * bias corrections from the partials.  Same arithmetic in the same order: parameters, moments and the published scalars
 * are bit-identical to the three-call form. */
int ufnd_clip_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* partials,
                         ufnd_step_state* state, void* stream);

/* ====================================================================================
 * Tier B -- the frozen, forward-only encoders that produce `text` and `visual`
 * (SURVEY.md 8 rows a10/a11).  bf16 operands on v_mfma_f32_16x16x32_bf16, fp32
 * accumulate, fp32 residual stream, fp32 LayerNorm/softmax statistics.
 * The arithmetic these replace is third-party: transformers' BertModel (called at
 * src/core_blocks/text_blocks.py:79) and CLIPVisionModel (geometry pointer
 * configs/model_configs/semantic.yaml:2); the pooling is the reference's own
 * (text_blocks.py:82-101,126-128).
 * ================================================================================== */
#define UFND_ACT_NONE 0
#define UFND_ACT_GELU 1       /* exact erf GELU (BERT intermediate) */
#define UFND_ACT_QUICK_GELU 2 /* x * sigmoid(1.702 x) (CLIP MLP) */
#define UFND_ACT_GELU_BWD 3       /* ufnd_gemm_bf16_dgrad only: out = acc * GELU'(aux) */
#define UFND_ACT_QUICK_GELU_BWD 4 /* ufnd_gemm_bf16_dgrad only: out = acc * quick_GELU'(aux) */

/* fp32 -> bf16 (round to nearest even); used once per weight tensor. */
int ufnd_cast_bf16(const float* src, void* dst_bf16, size_t n, void* stream);

/* out = act(A W^T + bias) + residual -- every nn.Linear of both encoders.
 *   A (M,K) bf16 row stride lda; W (N,K) bf16 row stride ldw (nn.Linear layout);
 *   bias fp32 (N) or NULL; residual fp32 (M,N) row stride ldr or NULL (added after act);
 *   out_bf16 (M,N) row stride ldo and/or out_f32 (M,N) row stride ldf (either may be NULL).
 *   K % 64 == 0, N % 64 == 0, strides multiples of 8, pointers 16-byte aligned. */
int ufnd_gemm_bf16(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                   float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                   void* stream);

/* Same, with an explicit tile configuration (the encoders pick tiles per compute-unit partition): tile_cfg < 0 =
 * automatic; otherwise the id of a tile built into this library (ufnd_gemm_bf16_tile_info; table in
 * csrc/gemm_bf16_kernel.hpp: block tile, wave grid, LDS ring slots of the A and W operands).  N must be a multiple
 * of the tile width; any other id is rejected with UFND_ERR_INVALID.
 * UFND_GEMM_TILE_PERSISTENT (round 4) names the persistent, software-pipelined form (csrc/gemm_bf16_pp.hpp): one workgroup per
 * compute unit walks 256 x 128 tiles with two accumulator sets -- the epilogue of a tile rides through the K loop of the
 * next one, the LDS-DMA operand stream never drains -- for K = 768, M a multiple of 256, N a multiple of 128, bf16 output
 * only (ufnd_gemm_bf16_ln: folded LayerNorm [+ activation], or the bf16 residual stream with out_stats; no fp32 residual /
 * output); any other call that names it is refused.  Same bits as the table's tiles (tests/test_gpu_gemm_pp.py).  The
 * automatic choice takes it for folded-LayerNorm calls with an activation once a workgroup gets two tiles (the FFN1 Linears
 * of both encoders at encoder lookahead >= 2), and for every supported call with at least three tiles per workgroup whose
 * last round of tiles is at least 93 % full (the configs[3] geometry, 65,536+ rows per launch). */
#define UFND_GEMM_TILE_PERSISTENT 64
int ufnd_gemm_bf16_ex(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                      float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                      int tile_cfg, void* stream);

/* LayerNorm-aware form: the LayerNorms around a Linear are folded into the GEMM epilogues, so that no
 * LayerNorm kernel (and no normalised copy of the activations) sits between two Linears.
 *   producer side  out_stats (M, N/32, 2): the partial {sum, sum of squares} of every aligned group of 32
 *                  columns of the fp32 output row (N/32 = ufnd_gemm_bf16_stat_parts(M,N,K), <= 24).
 *   consumer side  a_stats (M, a_parts, 2) + colsum (N): A is the bf16 rounding of the UN-normalised rows;
 *                  with W' = W * gamma (rows scaled, rounded to bf16), colsum[n] = sum_k W'[n,k] and
 *                  bias' = b + W beta:  LayerNorm(x) W^T + b = rstd (x W'^T - mean colsum) + bias'.
 *   residual side  r_stats (M, r_parts, 2) + r_gamma, r_beta (N): the residual added is
 *                  LayerNorm(residual) * r_gamma + r_beta (BERT's post-LN stream), computed on the fly.
 * a_stats and r_stats are mutually exclusive per call; partial counts are even, 2..24; statistics are
 * over `width` elements; partial layout and summation order are canonical (independent of the tile the
 * shape selects, no atomics): results are run-to-run identical and a row's outputs do not depend on the
 * batch it is computed in.
 * Accuracy of the folded form.  The consumer multiplies the bf16 rounding of the UN-normalised row, so a common-mode
 * offset of the row is rounded before it is subtracted: relative to the normalised output the error is about
 * (1 + |mean| / std) * 2^-9 (measured: tests/test_gpu_tier_b.py::test_gemm_ln_fold_error_grows_with_the_row_offset).
 * Rows of trained BERT / CLIP streams have |mean| / std well below 1; ufnd_ln_fold_guard (below) reports the largest
 * ratio in a statistics buffer so that a caller can fall back to a materialised LayerNorm (ufnd_layernorm +
 * ufnd_gemm_bf16) when it is not; the folding GEMM itself reports the same ratio through ufnd_gemm_ln.guard (below: the reduction
 * in its prologue, ONE atomicMax per workgroup at the kernel's very end -- an atomic in the prologue sits in front of the K loop's
 * counted vmcnt waits and cost 10 us per launch).  A non-finite statistic reports +inf.
 * Replaces nn.LayerNorm + nn.Linear pairs of the third-party encoders (transformers modeling_bert.py
 * BertSelfOutput / BertOutput, modeling_clip.py CLIPEncoderLayer) behind text_blocks.py:79. */
typedef struct ufnd_gemm_ln {
  const float* a_stats;
  const float* colsum;
  const float* r_stats;
  const float* r_gamma;
  const float* r_beta;
  float* out_stats;
  int a_parts, r_parts;
  float a_eps, r_eps;
  int width;
  int tile_cfg; /* < 0: automatic; otherwise a LayerNorm-aware tile id (ufnd_gemm_bf16_tile_info) or UFND_GEMM_TILE_PERSISTENT */
  /* bf16 residual stream (ABI v3): the residual operand as (M, ldrb) bf16 rows -- the rounding the producing GEMM already
   * wrote for its consumer -- instead of the fp32 `residual` argument (exclusive).  With it and out_f32 = NULL a residual
   * GEMM moves 2 + 2 B per element of the stream instead of 4 + 4 + 2; the stream then carries 8 significant bits per
   * layer (statistics are still taken from the fp32 sums before rounding). */
  const void* residual_bf16;
  int ldrb;
  /* fold guard inside the GEMM (ABI v4; optional, used with a_stats): UFND_FOLD_GUARD_SLOTS floats.  Every workgroup of the
   * launch leaves max(slot, largest |mean| * rstd among the rows it folds) in slot (workgroup id % UFND_FOLD_GUARD_SLOTS) -- the
   * statistics are in its registers at that point anyway, so the guard costs one atomicMax per workgroup at its very end instead
   * of a kernel that re-reads every statistics buffer of the pass (ufnd_ln_fold_guard_multi: 55-75 MB per pass).  The largest
   * ratio any folded row has had since the slots were last zeroed = the maximum over the slots (the caller reduces 4 KB). */
  float* guard;
} ufnd_gemm_ln;
#define UFND_FOLD_GUARD_SLOTS 1024
int ufnd_gemm_bf16_ln(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                      float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                      const ufnd_gemm_ln* ln, void* stream);
/* Number of {sum, sumsq} partials per row that ufnd_gemm_bf16_ln writes to out_stats for this shape
 * (0 = the shape's tile has no statistics epilogue). */
int ufnd_gemm_bf16_stat_parts(int M, int N, int K);

/* *guard = max(*guard, max over rows of |mean| * rstd) for the M rows of a statistics buffer (M, parts, 2) as
 * ufnd_gemm_bf16_ln writes and reads them (partial {sum, sumsq}; statistics over `width` elements). */
int ufnd_ln_fold_guard(const float* stats, int M, int parts, int width, float eps, float* guard, void* stream);
/* The same over `nbuf` statistics buffers of M rows each, `buf_stride` floats apart, in ONE launch: an encoder keeps one
 * statistics buffer per folded LayerNorm of a pass (2 x layers of them, back to back) and looks at all of them with a single
 * kernel at the end of the pass -- every row of every batch is guarded, at the price of re-reading the statistics once
 * (75 MB per 16,384-row text pass). */
int ufnd_ln_fold_guard_multi(const float* stats, int M, int parts, int nbuf, size_t buf_stride, int width, float eps, float* guard,
                             void* stream);

/* The tile table of this library: ids 0 .. ufnd_gemm_bf16_tile_count()-1; ufnd_gemm_bf16_tile_info returns 1 and the
 * block tile (rows x columns) of a tile that is built into the library (ln_aware: usable by ufnd_gemm_bf16_ln), 0 for
 * an id that exists only in the diagnostics library.  Timing ablations, in-kernel stamps and experimental tiles are
 * NOT reachable through this ABI (they live in libultrafnd_hip_diag.so, csrc/diag/). */
int ufnd_gemm_bf16_tile_count(void);
int ufnd_gemm_bf16_tile_info(int tile_cfg, int* bm, int* bn, int* ln_aware);

/* y = LayerNorm(x) * gamma + beta over the last dim (H % 256 == 0, H <= 1024).
 *   x fp32 rows with stride ldx; outputs (M,H) contiguous: bf16 (next GEMM's operand) and/or
 *   fp32 (the residual stream). */
int ufnd_layernorm(const float* x, int ldx, const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                   int M, int H, float eps, void* stream);

/* softmax(Q K^T / sqrt(d) + mask) V for every (batch, head); d = 64.
 *   qkv (B*L, 3*heads*64) bf16 = [Q | K | V] per token (the fused QKV projection's output);
 *   key_mask (B,L) int32, 1 = attend, 0 = masked (HF's additive finfo.min), or NULL;
 *   ctx (B*L, heads*64) bf16. */
int ufnd_attention_bf16(const void* qkv, const int32_t* key_mask, void* ctx, int B, int L, int heads,
                        void* stream);

/* BertSelfAttention of one layer in ONE launch, for 128-token samples: the fused Q/K/V projection of X (B*128, H) bf16
 * with the stacked weight Wqkv (3H, H) / bias (3H) -- optionally of LayerNorm(X) folded exactly as ufnd_gemm_bf16_ln
 * does (ln != NULL with a_stats / colsum / a_parts / a_eps / width; the other fields are ignored) -- followed by
 * ufnd_attention_bf16 on the result, without the (tokens, 3H) round trip through HBM: one workgroup computes one
 * sample's Q | K | V for two heads (a 128 x 384 x H tile), rounds them to bf16 into LDS and runs their attention.
 * Same operations in the same order as the two-launch form: ctx (B*128, H) bf16 is bit-identical to it.
 * L must be 128, heads even; other shapes use the two-launch form.
 * Replaces modeling_bert.py BertSelfAttention (query / key / value Linears + eager attention) behind text_blocks.py:79. */
int ufnd_qkv_attention_bf16(const void* X, const void* Wqkv, const float* bqkv, const int32_t* key_mask, void* ctx, int B, int L,
                            int heads, int ldx, int ldw, const ufnd_gemm_ln* ln, void* stream);

/* Packed (un-padded) forms: the tokens a padding mask keeps, concatenated over the batch (T rows; sequence b owns
 * rows cu_seqlens[b] .. cu_seqlens[b+1]; pos_ids (T) = each token's position in its sequence).  HF BertModel computes
 * the padded positions too (text_blocks.py:71-79 pads every string to max_length) and the pooling then ignores them
 * (:82-86); skipping them changes no kept value -- with prefix masks the features are bit-identical.
 *   ufnd_bert_embed_packed     LayerNorm(word[ids] + position[pos_ids] + token_type[0]) for T packed tokens
 *   ufnd_attention_bf16_varlen per-sequence attention over the packed fused-QKV rows (every key valid)
 *   ufnd_meanpool_l2_packed    mean over each sequence's rows, then v / (||v|| + 1e-9) */
int ufnd_bert_embed_packed(const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos, const float* type0,
                           const float* gamma, const float* beta, void* x_bf16, float* x_f32, int T, int max_pos, int H,
                           int vocab, float eps, void* stream);
int ufnd_attention_bf16_varlen(const void* qkv, const int32_t* cu_seqlens, void* ctx, int B, int max_len, int heads, void* stream);
int ufnd_meanpool_l2_packed(const float* hidden, const int32_t* cu_seqlens, const int32_t* pos_ids, float* out, int B, int H,
                            void* stream);

/* BertEmbeddings: LayerNorm(word[ids] + position[0..L) + token_type[0]).  Tables fp32.
 *   ids (B,L) int64 in [0,vocab).  Outputs (B*L,H): bf16 and fp32. */
int ufnd_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                    const float* beta, void* x_bf16, float* x_f32, int B, int L, int H, int vocab, float eps,
                    void* stream);

/* BERTContextEncoder.encode pooling (text_blocks.py:82-86,100): masked mean over tokens
 * (denominator clamp_min 1e-6) then v / (||v|| + 1e-9).  hidden (B,L,H) fp32, mask (B,L) int32. */
int ufnd_masked_meanpool_l2(const float* hidden, const int32_t* mask, float* out, int B, int L, int H,
                            void* stream);

/* BERTContextEncoder.encode_fields (text_blocks.py:108-128): parts (N, M, D) fp32 are the encoded
 * title / OCR / comment vectors of N records, valid (N, M) int32 marks the parts that exist
 * (empty strings are skipped by the reference); out (N, D) = mean of the valid parts, then
 * v / (||v|| + 1e-9); zero vector when a record has no part. */
int ufnd_field_mean_l2(const float* parts, const int32_t* valid, float* out, int N, int M, int D, void* stream);

/* CLIP patch embedding as an im2col-free GEMM operand: frames (N,3,S,S) fp32 ->
 * patches (N*(S/P)^2, 3*P*P) bf16 in the conv weight's (c,ky,kx) order. */
int ufnd_vit_patchify(const float* frames, void* patches_bf16, int N, int image, int patch, void* stream);

/* tokens = pre_layrnorm([class_embedding ; patch_emb] + position_embedding) -> x_f32 (N, P+1, H) and / or its
 * bf16 rounding x_bf16 (at least one of the two; with the bf16 residual stream only x_bf16), and optionally the rows' {sum, sumsq} as two partials per row
 * (stats (N*(P+1), 2, 2): the a_stats operand of ufnd_gemm_bf16_ln for the first layer). */
int ufnd_vit_assemble(const float* patch_emb, const float* cls, const float* pos, const float* gamma,
                      const float* beta, float* x_f32, void* x_bf16, float* stats, int N, int P, int H, float eps,
                      void* stream);

/* per-frame e / (||e|| + 1e-9); then, for F > 1, mean over the F frames of a sample and
 * L2-normalise again (text_blocks.py:126-128 idiom).  e (B*F, D) fp32 -> out (B, D). */
int ufnd_l2norm_frames(const float* e, float* out, int B, int F, int D, void* stream);

/* ------------------------------------------------------------------------------------
 * Batch assembly from the device-resident cache: CachedTensorDataset.__getitem__ +
 * default collate, src/training/forensic_trainer.py:60-83,232-234, and the gnn_Z[global_idx] gather of
 * _forward_batch :240-252.  For every item, dst row r = src row idx[r] (r < B); rows are row_bytes long
 * (a multiple of 8, both sides 8-B aligned and densely packed).  One launch for up to 8 tensors.
 * Indices come from the loader's permutation; an index outside [0, src_rows) is clamped into it.
 * ---------------------------------------------------------------------------------- */
#define UFND_GATHER_MAX_ITEMS 8
typedef struct ufnd_gather_item {
  const void* src;
  void* dst;
  int row_bytes;
  int64_t src_rows;
} ufnd_gather_item;
int ufnd_gather_rows(const int64_t* idx, int B, const ufnd_gather_item* items, int n_items, void* stream);

/* ------------------------------------------------------------------------------------
 * TemporalSyncNet.align, batched          src/core_blocks/temporal_blocks.py:102-140 (+ _cosine :10-13)
 *   text (B,D), visual (B,Dv) fp32 -> out (B,out_dim): W3 GELU(W0 [t, v^, t-v^, t*v^, cos] + b0) + b3
 *   with v^ = visual zero-padded / truncated to D.  w0 is (hidden, 4D+1) stored with row stride
 *   ufnd_temporal_weight_ld(D) (4D+1 rounded up to a multiple of 4, pad zero); w3 (out_dim, hidden).
 *   workspace: ufnd_temporal_workspace_floats(B, D, hidden) floats.
 *   dropout_p > 0 = the module's train mode: nn.Dropout(p) after the GELU, mask keyed by state->{seed, step} (the
 *   reference's align() is under torch.inference_mode, which does not switch dropout off, and its cache builder never
 *   calls .eval()); dropout_p = 0 (state may be NULL) = eval mode.
 * ---------------------------------------------------------------------------------- */
int ufnd_temporal_weight_ld(int in_dim);
size_t ufnd_temporal_workspace_floats(int B, int in_dim, int hidden);
int ufnd_temporal_align(const float* text, const float* visual, const float* w0, const float* b0, const float* w3,
                        const float* b3, float* workspace, float* out, int B, int in_dim, int vis_dim, int hidden,
                        int out_dim, float dropout_p, const ufnd_step_state* state, void* stream);

/* ------------------------------------------------------------------------------------
 * TemporalSyncNet.forward, the sequence path   src/core_blocks/temporal_blocks.py:141-157, _TinyTCN :16-43
 *   text_seq (B,T,text_dim), vis_seq (B,T,vis_dim) fp32, channel-last; in_ch = text_dim + vis_dim.
 *   Per block i: Conv1d(ch -> hid, kernel, padding 'same', dilation 2^i) -> BatchNorm1d -> GELU -> dropout, added to
 *   its input when the widths match; then out (B,out_dim) = head([mean_t h, max_t h]).
 *   layer.w is the Conv1d weight (hid, ch, kernel) re-packed tap-major: w[h][j*ch + c] = weight[h][c][j], row stride
 *   ufnd_tcn_weight_ld(ch, kernel) (ch*kernel rounded up to a multiple of 4, pad zero).  head_w (out_dim, 2*hid).
 *   train != 0: batch statistics (and the running_mean / running_var update, `momentum`), dropout keyed by `state`;
 *   train == 0: running statistics, no dropout.  Forward only (nothing in the reference trains this module).
 *   workspace: ufnd_tcn_workspace_floats(B, T, in_ch, hid, kernel) floats.
 * ---------------------------------------------------------------------------------- */
typedef struct ufnd_tcn_layer {
  const float *w, *b;                 /* packed conv weight, bias (hid) */
  const float *gamma, *beta;          /* BatchNorm1d weight, bias (hid) */
  float *running_mean, *running_var;  /* (hid); written when train != 0 */
} ufnd_tcn_layer;
int ufnd_tcn_weight_ld(int in_ch, int kernel);
size_t ufnd_tcn_workspace_floats(int B, int T, int in_ch, int hid, int kernel);
int ufnd_tcn_forward(const float* text_seq, int text_dim, const float* vis_seq, int vis_dim, int B, int T,
                     const ufnd_tcn_layer* layers, int n_layers, int kernel, int hid, const float* head_w,
                     const float* head_b, int out_dim, int train, float dropout_p, float momentum, float eps,
                     const ufnd_step_state* state, float* workspace, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Graph side of the trainer's construction: src/training/forensic_trainer.py
 *   build_adj_from_ocr :114-132, SimpleGCN :25-53, ForensicTrainer._pretrain_gnn :214-224.
 * Init-time work in the reference (an O(N^2) Python loop and dense (N,N) products on the CPU).
 * ---------------------------------------------------------------------------------- */

/* adj (N, N) row stride ld: 1 where i == j or Jaccard(set_i, set_j) >= thresh, else 0 (both sets empty ->
 * Jaccard 0).  Phrase sets as CSR: offsets (N+1) int32, tokens = every set's phrase ids, sorted ascending and
 * duplicate-free (any injective phrase -> id map; Jaccard only tests equality).  The comparison is
 * inter / (union + 1e-9) >= thresh in double, as the reference's Python floats do. */
int ufnd_ocr_adjacency(const int32_t* offsets, const int32_t* tokens, int N, double thresh, float* adj, int ld, void* stream);

typedef struct ufnd_gcn_params {
  const float* w1; /* lin1.weight (hid, in_dim) */
  const float* b1; /* lin1.bias   (hid)         */
  const float* w2; /* lin2.weight (out, hid)    */
  const float* b2; /* lin2.bias   (out)         */
} ufnd_gcn_params;

size_t ufnd_gcn_workspace_floats(int N, int in_dim, int hid, int out_dim, int train);

/* SimpleGCN.forward: z = lin2(A_norm @ dropout(gelu(lin1(A_norm @ x)))), A_norm = D^-1/2 (adj + I) D^-1/2 with
 * D = rowsum(adj + I) + 1e-9.  x (N, in_dim), adj (N, N) row stride ld_adj, z (N, out_dim).  dropout_p > 0 =
 * train mode (counter-based mask keyed by state->seed / state->step); in_dim % 4 == 0, hid and out_dim % 32 == 0. */
int ufnd_gcn_forward(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* z, float* workspace, int N,
                     int in_dim, int hid, int out_dim, float dropout_p, const ufnd_step_state* state, void* stream);

/* One step of _pretrain_gnn: forward (train-mode dropout), loss = mse(sigmoid(z head_w^T + head_b), rowsum(adj) /
 * max(1, N)) written to *loss, backward through the GCN, torch.optim.Adam(lr, weight_decay as L2 on the gradient,
 * betas (0.9, 0.999), eps 1e-8) on the GCN parameters only (the head is not in that optimizer).  The four
 * parameter tensors must be ONE flat buffer [w1 | b1 | w2 | b2]; exp_avg / exp_avg_sq have that layout; `step`
 * is 1-based; z receives the forward output of this step. */
int ufnd_gcn_pretrain_step(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* exp_avg,
                           float* exp_avg_sq, const float* head_w, const float* head_b, float* z, float* workspace, int N,
                           int in_dim, int hid, int out_dim, float dropout_p, float lr, float weight_decay, int step,
                           const ufnd_step_state* state, float* loss, void* stream);

/* ------------------------------------------------------------------------------------
 * The integrated trainer variant's in-graph GNN: src/training/forensic_trainer_integrated.py
 *   build_adj_from_ocr_sets :77-98 (weighted Jaccard adjacency of the mini-batch), _pack_batch / _forward :203-224,
 *   src/models/gnn/gnn_model.py GNNModel :7-41 (lin1 -> A_norm -> ReLU -> dropout -> A_norm -> lin2), trained WITH the head.
 *   (The reference's `torch.stack([T, A, V, U]).mean(0)` at :205 stacks tensors of widths 768/128/512/256 and raises; the
 *   node feature built here is the 416-wide one that line's own comment and `gnn_in_dim = 416` describe, i.e. the main
 *   trainer's compact slice concat, forensic_trainer.py:193-195.)
 * ---------------------------------------------------------------------------------- */

/* adj (N, N) row stride ld: Jaccard(set_i, set_j) where it is >= thresh, i != j and both sets are non-empty, else 0
 * (zero diagonal).  Sets as for ufnd_ocr_adjacency. */
int ufnd_ocr_adjacency_weighted(const int32_t* offsets, const int32_t* tokens, int N, double thresh, float* adj, int ld, void* stream);

/* out (B, n_text + n_audio + n_visual + n_temporal) = [text[:, :n_text] | audio[:, :n_audio] | visual[:, :n_visual] |
 * temporal[:, :n_temporal]] with every row divided by (its L2 norm + 1e-9)     forensic_trainer.py:193-195 */
int ufnd_node_features(const float* text, int ld_text, const float* audio, int ld_audio, const float* visual, int ld_visual,
                       const float* temporal, int ld_temporal, int n_text, int n_audio, int n_visual, int n_temporal, int B,
                       float* out, void* stream);

size_t ufnd_gnn_workspace_floats(int N, int in_dim, int hid, int out_dim);
/* GNNModel.forward: z (N, out_dim).  dropout_p > 0 = train mode (mask keyed by state->{seed, step}).  The workspace keeps
 * what ufnd_gnn_backward needs. */
int ufnd_gnn_forward(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* z, float* workspace, int N,
                     int in_dim, int hid, int out_dim, float dropout_p, const ufnd_step_state* state, void* stream);
/* its autograd backward for a gradient d_z (N, out_dim) at z: the four parameter gradients (overwritten). */
int ufnd_gnn_backward(const float* x, const ufnd_gcn_params* p, float* g_w1, float* g_b1, float* g_w2, float* g_b2, const float* d_z,
                      float* workspace, int N, int in_dim, int hid, int out_dim, float dropout_p, const ufnd_step_state* state,
                      void* stream);
/* d loss / d text_features (B, text_dim) and d loss / d visual_features (B, visual_dim) out of the workspace a fusion backward
 * has just filled (either pointer may be NULL).  The reference's trainer treats both as cached data
 * (src/training/forensic_trainer.py:60-83); with trainable encoders they are where the encoders' backward starts. */
int ufnd_fusion_feature_grads(const ufnd_dims* d, const ufnd_fusion_params* p, float* workspace, int B, float* d_text, float* d_visual,
                              const ufnd_step_state* state, void* stream);

/* d loss / d gnn_feat (B, gnn_dim) after ufnd_fusion_backward[_phase] has run on `workspace`. */
int ufnd_fusion_gnn_input_grad(const ufnd_dims* d, const ufnd_fusion_params* p, float* workspace, int B, float* d_gnn,
                               const ufnd_step_state* state, void* stream);

/* ------------------------------------------------------------------------------------
 * Compute-unit partitions.  The reference runs its step in program order on one queue
 * (src/training/forensic_trainer.py:285-298); here the text encoder, the visual encoder and the
 * head -> exchange -> optimizer chain are concurrent chains on their own HIP streams, each confined to
 * its share of the 256 CUs so that whole-CU GEMM workgroups of different chains do not displace each other.
 *   ufnd_stream_create_cu_mask: *stream_out = a hipStream_t restricted to the CUs whose bits are set in
 *     mask_words (n_words x 32 bits; bit b = logical CU b in the driver's numbering).
 *   ufnd_device_cu_count: CUs of the current device (256 on MI355X), negative on error.
 * ---------------------------------------------------------------------------------- */
int ufnd_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out);
int ufnd_stream_destroy(void* stream);
int ufnd_device_cu_count(void);

/* ====================================================================================
 * Tier-B backward (ABI v3; SURVEY.md 8b's *_bwd list).  The reference never trains its encoders
 * (src/core_blocks/text_blocks.py:52 `.eval()`, :63 `inference_mode`), so no reference interface is replaced here: these
 * are the backward forms of the forward entry points above, for TrainConfig.train_encoders.  Parity: torch autograd over
 * oracle/encoders_ref.py, itself checked against the installed third-party classes ("parity unpinned by the reference").
 * bf16 operands, fp32 accumulation; parameter gradients are written in fp32; nothing uses atomics (results are
 * run-to-run identical).
 * ================================================================================== */

/* out (M, N) = dY (M, K) x Wt (N, K)^T [x act'(aux)] [+ residual]: the data gradient of y = x W^T through the TRANSPOSED
 * weight copy Wt = W^T (N = in_features rows of K = out_features).  act = UFND_ACT_GELU_BWD / UFND_ACT_QUICK_GELU_BWD
 * multiplies by the activation's derivative at the pre-activations aux (M, ldaux) bf16 (fused dgrad through FFN1's
 * activation); residual (M, ldr) fp32 adds the gradient arriving over the residual branch.  aux and residual are exclusive. */
int ufnd_gemm_bf16_dgrad(const void* dY, const void* Wt, const float* residual, const void* aux, void* out_bf16, float* out_f32,
                         int M, int N, int K, int lda, int ldw, int ldr, int ldaux, int ldo, int ldf, int act, void* stream);

/* dW (n_out, k_in) fp32 [+]= dYt (n_out, tokens) x Xt (k_in, tokens)^T: the weight gradient of y = x W^T from the transposed
 * activations (ufnd_transpose_bf16; `tokens` padded to a multiple of 64 with zero columns).  The token range is cut into
 * slices (grid = tiles x slices, fp32 partial slabs in `workspace`), a reduce pass adds the slabs in slice order. */
size_t ufnd_gemm_bf16_wgrad_workspace_floats(int n_out, int k_in, int tokens);
int ufnd_gemm_bf16_wgrad(const void* dYt, const void* Xt, float* dW, int n_out, int k_in, int tokens, int lda, int ldb, int ldw,
                         float* workspace, int accumulate, void* stream);

/* A LayerNorm's deferred parameter-gradient finish (ufnd_layernorm_bwd with accumulate = UFND_PARTIALS_DEFER leaves the block
 * partials part[blk][2][H] in its workspace): out0 (H) = dgamma, out1 (H) = dbeta, the nblk blocks added in ascending order. */
typedef struct ufnd_partials_job {
  const float* part;
  int nblk, H;
  float *out0, *out1;
} ufnd_partials_job;
#define UFND_PARTIALS_DEFER 2

/* One Linear's weight and bias gradient from the row-major activations, in THREE launches (round 4; the five-launch sequence
 * ufnd_transpose_bf16 x 2 + column-sum finish + ufnd_gemm_bf16_wgrad + slab reduce stays available):
 *   dW (N, K) fp32 = dY (M, N)^T x X (M, K), db (N) = column sums of dY (NULL: skipped); both OVERWRITTEN.
 *   1. both operand transposes (dYt (N, ldt), Xt (K, ldt), ldt >= M rounded up to 64; zero-padded) + dY's column-sum partials;
 *   2. the NT kernel over the token dimension in slices (slab_workspace: ufnd_gemm_bf16_wgrad_workspace_floats(N, K, Mpad));
 *   3. ONE finish launch: slab reduce of dW, column-sum finish of db (colsum_workspace: ufnd_transpose_colsum_workspace_floats)
 *      and, if `extra` != NULL, a deferred LayerNorm dgamma / dbeta finish.
 * Same arithmetic in the same order as the five-launch sequence: bit-identical gradients.
 * `part`: UFND_WGRAD_ALL, or the two halves for a caller that runs the product beside its data-gradient chain on a second
 * stream: UFND_WGRAD_TRANSPOSE (launch 1: dY and X are free to be overwritten once it has run), then UFND_WGRAD_PRODUCT
 * (launches 2 and 3, same arguments). */
#define UFND_WGRAD_ALL 0
#define UFND_WGRAD_TRANSPOSE 1
#define UFND_WGRAD_PRODUCT 2
int ufnd_linear_wgrad(const void* dY, int lddy, const void* X, int ldx, int M, int N, int K, float* dW, float* db, void* dYt, void* Xt, int ldt,
                      float* slab_workspace, float* colsum_workspace, const ufnd_partials_job* extra, int part, void* stream);

/* bf16 W (rows, ld_w) and W^T (cols, ld_wt) of MANY Linears from their fp32 masters (rows, ld_master) in ONE launch (after an
 * optimizer step).  rows and cols multiples of 64, every pointer 16-B aligned, ld_master % 4 == 0, ld_w % 8 == 0, ld_wt % 8 == 0;
 * tile0 = the item's first 64 x 64 tile in the launch (items sorted by it: tile0 of item i + (rows/64)(cols/64) = tile0 of item i+1);
 * `items_device` is a DEVICE-resident table (it is read by the kernel), total_tiles the sum over the items. */
typedef struct ufnd_refresh_item {
  const void* master;
  void* w;
  void* wt;
  int rows, cols, ld_master, ld_w, ld_wt, tile0;
} ufnd_refresh_item;
int ufnd_refresh_operands(const ufnd_refresh_item* items_device, int n_items, int total_tiles, void* stream);

/* dst (cols, ldd) bf16 = src (rows, lds)^T, columns rows..rows_pad-1 zero; src bf16, or fp32 (src_is_f32: cast on the way --
 * weight masters to transposed operand copies).  colsum != NULL (bf16 sources): colsum (cols) [+]= the column sums of src
 * (bias gradients: db = sum over tokens of dy), two-stage through colsum_ws (ufnd_transpose_colsum_workspace_floats). */
size_t ufnd_transpose_colsum_workspace_floats(int rows_pad, int cols);
int ufnd_transpose_bf16(const void* src, int src_is_f32, int rows, int cols, int lds, void* dst, int ldd, int rows_pad, float* colsum,
                        float* colsum_ws, int colsum_accumulate, void* stream);

/* ufnd_attention_bf16 that also keeps lse (B L, heads) fp32: each query's log-sum-exp of its scaled, masked scores in the
 * log2 domain -- what ufnd_attention_bf16_bwd recomputes P from. */
int ufnd_attention_bf16_lse(const void* qkv, const int32_t* key_mask, void* ctx, float* lse, int B, int L, int heads, void* stream);
/* dqkv (B L, 3H) bf16 = the gradients of the fused q | k | v rows, from dctx (B L, H) bf16, the forward's qkv, ctx and lse.
 * workspace: ufnd_attention_bwd_workspace_floats floats (delta = rowsum(dO o O)).  Three launches (delta; dQ; dK and dV). */
size_t ufnd_attention_bwd_workspace_floats(int B, int L, int heads);
int ufnd_attention_bf16_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, const int32_t* key_mask, void* dqkv,
                            float* workspace, int B, int L, int heads, void* stream);

/* LayerNorm backward: dx = rstd (g - mean(g) - xh mean(g xh)) [+ add], g = dy gamma, xh = (x - mean) rstd, from the LayerNorm's
 * INPUT x (M rows of stride ldx); dx as fp32 and / or bf16 (stride lddx).  dgamma / dbeta (H) [+]= their row sums (two-stage,
 * through `workspace`: ufnd_layernorm_bwd_workspace_floats); both may be NULL (then workspace may be too).
 * accumulate = UFND_PARTIALS_DEFER: only the block partials are written (ufnd_layernorm_bwd_blocks(M) blocks); the caller hands
 * them to ufnd_linear_wgrad (`extra`) or ufnd_row_partials_finish before the workspace is reused -- dgamma / dbeta are not touched. */
size_t ufnd_layernorm_bwd_workspace_floats(int M, int H);
int ufnd_layernorm_bwd_blocks(int M);
int ufnd_row_partials_finish(const ufnd_partials_job* job, int accumulate, void* stream);
int ufnd_layernorm_bwd(const float* x, int ldx, const float* gamma, const float* dy, int lddy, const float* add, int ldadd, float* dx_f32,
                       void* dx_bf16, int lddx, float* dgamma, float* dbeta, float* workspace, int accumulate, int M, int H, float eps,
                       void* stream);

/* out = act(x) over n bf16 elements (act = UFND_ACT_GELU / UFND_ACT_QUICK_GELU; n a multiple of 8): the training forward keeps
 * FFN1's pre-activations (for the fused activation backward of ufnd_gemm_bf16_dgrad) next to their activation. */
int ufnd_act_bf16(const void* x, void* out, size_t n, int act, void* stream);

/* ufnd_masked_meanpool_l2 backward: dhidden (B L, H) from dfeat (B, H) and the forward's inputs. */
int ufnd_masked_meanpool_l2_bwd(const float* hidden, const int32_t* mask, const float* dfeat, float* dhidden, int B, int L, int H, void* stream);
/* ufnd_l2norm_frames backward: de (B F, D) from dfeat (B, D) and the forward's input e. */
int ufnd_l2norm_frames_bwd(const float* e, const float* dfeat, float* de, int B, int F, int D, void* stream);
/* BERT embeddings backward from ds (B L, H) = the gradient of the summed embeddings (ufnd_bert_embed with gamma = beta = NULL
 * returns those sums; their LayerNorm is ufnd_layernorm / ufnd_layernorm_bwd): dword (vocab, H), dpos (max_pos, H),
 * dtype (type_vocab, H) are overwritten. */
int ufnd_bert_embed_bwd(const int64_t* ids, const float* ds, float* dword, float* dpos, float* dtype, int B, int L, int H, int vocab,
                        int max_pos, int type_vocab, void* stream);
/* ViT token assembly backward from ds (N (P + 1), H): dcls (H), dpos (P + 1, H) overwritten; dpe (N P, H) bf16 = the patch rows
 * (the `patch_embed_bwd` of SURVEY 8b is ufnd_gemm_bf16_wgrad on dpe and the patch matrix). */
int ufnd_vit_assemble_bwd(const float* ds, float* dcls, float* dpos, void* dpe_bf16, int N, int P, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ULTRAFND_HIP_H */
