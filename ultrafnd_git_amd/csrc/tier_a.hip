// Tier-A head: CrossModalTransformer + DeepTruthClassifier forward / backward, cross-entropy.
// Reference arithmetic: src/models/fusion/cross_modal_transformer.py:39-55,134-210 and
// src/models/fusion/deep_truth_classifier.py:54-74,88-90,148-171 (fp32 throughout).
//
// Data layout (per call, B rows, H = hidden):
//   CAT  (B, 16H)  the fuse_mlp input, written in place by its producers -- slot order
//        [t a v u | t+a t*a |t-a| t+v t*v |t-v| t+u v+u | tv* ta* vu* | g]   (:172-195)
//   QKV  (B, 9H)   stacked co-attention projections, column blocks
//        [q_tv q_ta | k_tv v_tv q_vu | k_ta v_ta | k_vu v_vu]  so that each source vector
//        (t, v, a, u) feeds ONE stacked GEMM forward and ONE stacked GEMM backward.
// Row kernels use one 64-lane wave per sample; lane l owns columns 4l..4l+3 (+256 i).
#include "gemm_f32.hpp"

namespace {

constexpr uint32_t LAYER_FUSE0 = 1, LAYER_FUSE3 = 2, LAYER_PRE0 = 3, LAYER_PRE3 = 4, LAYER_TREE = 5;
#ifndef UFND_NSPLIT_FUSE0
#define UFND_NSPLIT_FUSE0 4
#endif
// fuse_mlp.0 forward: K = 16H split over the grid (x4 waves in-block), then one epilogue pass over the partials.  Measured (round 4, one
// box, A/B/A; 16 ways before): B = 32: 8 ways 0.2309 ms per head step against 0.2347-0.2357 (16), 0.2354 (4), 0.2364 (12); B = 256: 0.5146
// (8) against 0.5176-0.5188 (16) and 0.5016 (4).  One value for every batch size: a row's arithmetic does not depend on its batch.
constexpr int KSPLIT_FUSE0 = 8;
constexpr int NSPLIT_FUSE0 = UFND_NSPLIT_FUSE0;  // fuse_mlp.0 dX: contraction 2H split 4 ways (x4 waves in-block)

inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

// ---------------------------------------------------------------- workspaces
struct FusionWs {
  float *cat, *qkv, *evid, *gate, *s, *part1, *z1, *h1, *z2;
  float *dz2, *dz1, *dcatp, *dtavu, *dg, *dqkv, *dout, *gpart;
  size_t total;
};
// Row-sliced parameter reductions (gate_param / node_param): up to PARAM_SLICES_MAX slices of PARAM_SLICE_ROWS rows each, used
// when the batch has more than PARAM_SLICE_MIN_B rows (at B = 32 the one-pass kernels stay: same bits, no extra launch)
constexpr int PARAM_SLICE_ROWS = 8, PARAM_SLICES_MAX = 32, PARAM_SLICE_MIN_B = 64;
inline int param_slices(int B) {
  if (B <= PARAM_SLICE_MIN_B) return 1;
  const int s = (B + PARAM_SLICE_ROWS - 1) / PARAM_SLICE_ROWS;
  return s > PARAM_SLICES_MAX ? PARAM_SLICES_MAX : s;
}
FusionWs carve_fusion(const ufnd_dims& d, int B, float* base) {
  const size_t H = d.hidden;
  FusionWs w;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += al64(n); return p; };
  w.cat = take((size_t)B * 16 * H);
  w.qkv = take((size_t)B * 9 * H);
  w.evid = take((size_t)B * 4);
  w.gate = take((size_t)B * 4);
  w.s = take((size_t)B * 4);
  w.part1 = take((size_t)KSPLIT_FUSE0 * B * 2 * H);
  w.z1 = take((size_t)B * 2 * H);
  w.h1 = take((size_t)B * 2 * H);
  w.z2 = take((size_t)B * H);
  w.dz2 = take((size_t)B * H);
  w.dz1 = take((size_t)B * 2 * H);
  w.dcatp = take((size_t)NSPLIT_FUSE0 * B * 16 * H);
  w.dtavu = take((size_t)4 * B * H);
  w.dg = take((size_t)B * H);
  w.dqkv = take((size_t)B * 9 * H);
  w.dout = take((size_t)B * 4);
  w.gpart = take(param_slices(B) > 1 ? (size_t)param_slices(B) * 3 * (5 * H + 64) : 0);     // gate_param row-slice partials
  w.total = o;
  return w;
}

struct ClfWs {
  float *xin, *z3, *h3, *z4, *hh, *alpha, *fs, *df, *dz4, *dz3, *dl, *lrow, *npart;
  int ldx;
  size_t total;
};
ClfWs carve_clf(const ufnd_dims& d, int B, float* base) {
  const size_t H = d.hidden;
  ClfWs w;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += al64(n); return p; };
  w.ldx = d.hidden + 4;  // [fused | aux (<= 4) | zero pad]; row stride stays a multiple of 4
  w.xin = take((size_t)B * w.ldx);
  w.z3 = take((size_t)B * H);
  w.h3 = take((size_t)B * H);
  w.z4 = take((size_t)B * H);
  w.hh = take((size_t)B * H);
  w.alpha = take((size_t)d.trees * d.depth * H);
  w.fs = take((size_t)B * 64);
  w.df = take((size_t)B * 64);
  w.dz4 = take((size_t)B * H);
  w.dz3 = take((size_t)B * H);
  w.dl = take((size_t)B * 4);
  w.lrow = take((size_t)B);                      // CE rows of the fused head step (node_head writes them, node_bwd's block 0 averages them)
  // node_param row-slice partials: per slice [gates TK x H | thresh 64 | bypass 2 x H | bypass bias 64 | leaves trees x 2^depth x 2]
  w.npart = take(param_slices(B) > 1 ? (size_t)param_slices(B) * ((size_t)(d.trees * d.depth + 2) * H + 128 + (size_t)d.trees * (1 << d.depth) * 2) : 0);
  w.total = o;
  return w;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// ---------------------------------------------------------------- split-K epilogue
// Y = drop(act(sum_s PART[s] + bias)), Z = pre-activation.   (M*N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* part, int nsplit, int M, int N,
                                                              const float* bias, float* Z, float* Y, int act,
                                                              float drop_p, uint32_t layer,
                                                              const ufnd_step_state* st) {
  const size_t total4 = (size_t)M * N / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int n = (int)(e % N);
    f32x4 v = ld4(part + e);
    for (int s = 1; s < nsplit; ++s) v += ld4(part + (size_t)s * M * N + e);
    if (bias) v += ld4(bias + n);
    if (Z) st4(Z + e, v);
    float dm[4];
    dropout_mul4(st, drop_p, layer, (uint32_t)e, dm);      // (e is a multiple of 4: one Philox evaluation for the four elements)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = v[q];
      if (act == 1) x = gelu_f(x);
      x *= dm[q];
      v[q] = x;
    }
    st4(Y + e, v);
  }
}

// ---------------------------------------------------------------- evidence scalars + gates
// cross_modal_transformer.py:153-164 (no_grad scalars) and :48 (evidence_proj -> sigmoid)
struct EvPtrs {
  const float *w0[3], *b0[3], *w2[3], *b2[3];
};
struct EvGrads {
  float *w0[3], *b0[3], *w2[3], *b2[3];
};
// ---------------------------------------------------------------- the classifier's input preparation
// Blocks [0, B): the input panel [fused | aux | 0] (the fused columns are copied only when the fusion did not write them in
// place).  Blocks [B, B + trees * depth): alpha[tk] = softmax(gates[tk]) (deep_truth_classifier.py:64).  Its own launch ahead of the
// classifier's GEMMs (clf_prep_kernel), or -- the fused head step -- extra blocks of the co-attention row kernel: nothing here
// depends on the fusion's outputs when the fused columns are written in place.
struct ClfPrep {
  const float *fused, *aux, *gates;
  float *xin, *alpha;
  int ldf, aux_dim, ldx, copy_fused, blocks;      // blocks = B + trees * depth (0: no such role in this launch)
};
__device__ __forceinline__ void clf_prep_block(const ClfPrep& a, int B, int H, int block) {
  __shared__ float sh[4];
  if (block < B) {
    const int row = block;
    if (a.copy_fused)
      for (int c = threadIdx.x; c < H; c += blockDim.x) a.xin[(size_t)row * a.ldx + c] = a.fused[(size_t)row * a.ldf + c];
    if (threadIdx.x < 4)
      a.xin[(size_t)row * a.ldx + H + threadIdx.x] = (a.aux && (int)threadIdx.x < a.aux_dim) ? a.aux[row * a.aux_dim + threadIdx.x] : 0.0f;
    return;
  }
  const int tk = block - B;
  const float* gp = a.gates + (size_t)tk * H;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < H; c += 256) mx = fmaxf(mx, gp[c]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float sm = 0;
  for (int c = threadIdx.x; c < H; c += 256) sm += __expf(gp[c] - mx);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sm;
  __syncthreads();
  sm = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  for (int c = threadIdx.x; c < H; c += 256) a.alpha[(size_t)tk * H + c] = __expf(gp[c] - mx) / sm;
}
__global__ __launch_bounds__(256) void clf_prep_kernel(ClfPrep a, int B, int H) { clf_prep_block(a, B, H, blockIdx.x); }

// ---------------------------------------------------------------- co-attention + pairwise -> CAT
// cross_modal_transformer.py:153-164 (evidence scalars), :48 (evidence gates), :44-54 and :172-178 (co-attention combine, pairwise features).
// ONE WORKGROUP per row (round 3; one wave per row before): wave 0 the evidence scalars while waves 1-3 take one co-attention score each; then
// waves 0-2 one evidence gate each (their parameter loads requested before the barrier); then the 11 pairwise / co-attention slots
// are dealt over the waves by column chunk and slot group.  Every scalar is still one wave's reduction over the same lanes and
// every element the same expression: the same bits as the one-wave form it replaced, a third of its chain.
template <int NI>
__global__ __launch_bounds__(256) void coattn_pairs_wg_kernel(float* cat, const float* qkv, float* gate_io, int B, int H, float* s_out,
                                                              EvPtrs ev, float* evid, float* forensic, ClfPrep prep) {
  if ((int)blockIdx.x >= B) {                 // (fused head step: the classifier's input preparation rides in this launch)
    clf_prep_block(prep, B, H, blockIdx.x - B);
    return;
  }
  __shared__ float e_lds[4], g_lds[4], s_lds[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  float* c = cat + (size_t)row * 16 * H;
  const float* q = qkv + (size_t)row * 9 * H;
  const int QI[3] = {0, 1, 4}, KI[3] = {2, 5, 7}, VI[3] = {3, 6, 8};
  const int XS[3] = {0, 0, 2}, YS[3] = {2, 1, 3};
  // gate parameters of block w (waves 0-2), requested now: 4 NI hidden units per lane
  float pw0[4 * NI][3], pb0[4 * NI], pw2[4 * NI];
  if (w < 3) {
#pragma unroll
    for (int q4 = 0; q4 < 4 * NI; ++q4) {
      const int jj = lane + 64 * q4;
      pw0[q4][0] = ev.w0[w][jj * 3 + 0]; pw0[q4][1] = ev.w0[w][jj * 3 + 1]; pw0[q4][2] = ev.w0[w][jj * 3 + 2];
      pb0[q4] = ev.b0[w][jj];
      pw2[q4] = ev.w2[w][jj];
    }
  }
  if (w == 0) {          // evidence scalars (cross_modal_transformer.py:153-164)
    float tt = 0, vv = 0, uu = 0, tv = 0, tu = 0, ta = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * lane + 256 * i;
      const f32x4 t = ld4(c + col), v = ld4(c + 2 * H + col), u = ld4(c + 3 * H + col);
      tt += dot4(t, t); vv += dot4(v, v); uu += dot4(u, u); tv += dot4(t, v); tu += dot4(t, u);
      ta += fabsf(t[0]) + fabsf(t[1]) + fabsf(t[2]) + fabsf(t[3]);
    }
    tt = wave_sum(tt); vv = wave_sum(vv); uu = wave_sum(uu); tv = wave_sum(tv); tu = wave_sum(tu); ta = wave_sum(ta);
    const float nt = fmaxf(sqrtf(tt), 1e-12f), nv = fmaxf(sqrtf(vv), 1e-12f), nu = fmaxf(sqrtf(uu), 1e-12f);
    const float conf = 1.0f - 0.5f * (fminf(fmaxf(tv / (nt * nv), -1.0f), 1.0f) + 1.0f);
    const float delay = 1.0f - 0.5f * (fminf(fmaxf(tu / (nt * nu), -1.0f), 1.0f) + 1.0f);
    const float emo = tanhf(ta / (float)H);
    if (lane == 0) {
      e_lds[0] = conf; e_lds[1] = emo; e_lds[2] = delay;
      st4(evid + (size_t)row * 4, f32x4{conf, emo, delay, 0.f});
      forensic[row] = emo;                // emotion_intensity
      forensic[B + row] = conf;           // semantic_conflict
      forensic[2 * B + row] = delay;      // temporal_delay
    }
  } else {               // co-attention score of block w - 1
    const int b = w - 1;
    float dots = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * lane + 256 * i;
      dots += dot4(ld4(q + QI[b] * H + col), ld4(q + KI[b] * H + col));
    }
    const float inv = 1.0f / sqrtf((float)H);
    const float sb = sigmoid_f(wave_sum(dots) * inv);
    if (lane == 0) s_lds[b] = sb;
  }
  __syncthreads();
  if (w < 3) {           // evidence gate of block w (cross_modal_transformer.py:48: evidence_proj -> sigmoid)
    const float conf = e_lds[0], emo = e_lds[1], delay = e_lds[2];
    const float e0 = w == 0 ? conf : (w == 1 ? emo : delay), e1 = w == 0 ? emo : 0.f, e2 = 0.f;
    float o = 0.0f;
#pragma unroll
    for (int q4 = 0; q4 < 4 * NI; ++q4) {
      const float pre = pw0[q4][0] * e0 + pw0[q4][1] * e1 + pw0[q4][2] * e2 + pb0[q4];
      o += pw2[q4] * gelu_f(pre);
    }
    const float gb = sigmoid_f(wave_sum(o) + ev.b2[w][0]);
    if (lane == 0) g_lds[w] = gb;
  }
  __syncthreads();
  const float g3[3] = {g_lds[0], g_lds[1], g_lds[2]}, s3[3] = {s_lds[0], s_lds[1], s_lds[2]};
  if (threadIdx.x == 0) {
    st4(gate_io + (size_t)row * 4, f32x4{g3[0], g3[1], g3[2], 0.f});
    st4(s_out + (size_t)row * 4, f32x4{s3[0], s3[1], s3[2], 0.f});
  }
  // slots 4..14: wave w takes column chunk w % NI and slot group w / NI of SG = 4 / NI groups
  constexpr int SG = 4 / NI;
  const int i = w % NI, sg = w / NI;
  auto mine = [&](int slot) { return ((slot - 4) * SG) / 11 == sg; };
  {
    const int col = 4 * lane + 256 * i;
    const f32x4 t = ld4(c + col), a = ld4(c + H + col), v = ld4(c + 2 * H + col), u = ld4(c + 3 * H + col);
    f32x4 ab;
    if (mine(4)) st4(c + 4 * H + col, t + a);
    if (mine(5)) st4(c + 5 * H + col, t * a);
#pragma unroll
    for (int k = 0; k < 4; ++k) ab[k] = fabsf(t[k] - a[k]);
    if (mine(6)) st4(c + 6 * H + col, ab);
#pragma unroll
    for (int k = 0; k < 4; ++k) ab[k] = fabsf(t[k] - v[k]);
    if (mine(7)) st4(c + 7 * H + col, t + v);
    if (mine(8)) st4(c + 8 * H + col, t * v);
    if (mine(9)) st4(c + 9 * H + col, ab);
    if (mine(10)) st4(c + 10 * H + col, t + u);
    if (mine(11)) st4(c + 11 * H + col, v + u);
    const f32x4 xs[4] = {t, a, v, u};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (mine(12 + b)) {
        const f32x4 val = ld4(q + VI[b] * H + col);
        const f32x4 base = 0.5f * (xs[XS[b]] + xs[YS[b]]);
        st4(c + (12 + b) * H + col, g3[b] * (s3[b] * val) + (1.0f - g3[b]) * base);
      }
    }
  }
}

// backward of the above.  dcatp: [NSPLIT][B][16H] partial sums of dCAT.
// ONE WORKGROUP of 2 NI waves per row (round 3; one wave per 256-column slice before).  Waves [0, NI) ("A", one per 256-column chunk): the co-attention
// outputs' gradients, the two row reductions (combined over the chunks in LDS, in chunk order), d q / k / v and the gate gradient.
// Waves [NI, 2 NI) ("B"): the 12 x nsplit partial-sum loads of the pairwise slots and their part of d t / a / v / u, handed to the A
// wave of the chunk through LDS, which adds the co-attention halves in the one-wave form's order: the same bits, half the loads per
// wave and twice the waves.
template <int NI, int NSPLIT>
__global__ __launch_bounds__(2 * NI * 64) void coattn_pairs_bwd_wg_kernel(const float* dcatp, const float* cat, const float* qkv,
                                                                          const float* gate, const float* s_in, int B, int H, float* dtavu,
                                                                          float* dqkv, float* dg, float* dout) {
  __shared__ float red[NI][8];
  __shared__ f32x4 dpart[NI][4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = w / NI, part = w % NI;
  const int row = blockIdx.x;
  const float* c = cat + (size_t)row * 16 * H;
  const float* q = qkv + (size_t)row * 9 * H;
  const int QI[3] = {0, 1, 4}, KI[3] = {2, 5, 7}, VI[3] = {3, 6, 8};
  const int XS[3] = {0, 0, 2}, YS[3] = {2, 1, 3};
  auto G = [&](int slot, int col) {
    f32x4 v = ld4(dcatp + (size_t)row * 16 * H + slot * H + col);      // (NSPLIT is a compile-time constant: every partial of every slot in flight together)
#pragma unroll
    for (int p = 1; p < NSPLIT; ++p) v += ld4(dcatp + ((size_t)p * B + row) * 16 * H + slot * H + col);
    return v;
  };
  const int col = 4 * lane + 256 * part;
  const f32x4 gt = ld4(gate + (size_t)row * 4), sv = ld4(s_in + (size_t)row * 4);
  f32x4 dob[3] = {};
  float r_dg[3] = {0, 0, 0}, r_ds[3] = {0, 0, 0};
  if (role == 1) {
    const f32x4 t = ld4(c + col), a = ld4(c + H + col), v = ld4(c + 2 * H + col), u = ld4(c + 3 * H + col);
    const f32x4 g4 = G(4, col), g5 = G(5, col), g6 = G(6, col), g7 = G(7, col), g8 = G(8, col), g9 = G(9, col),
                g10 = G(10, col), g11 = G(11, col);
    f32x4 sta, stv;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sta[k] = sgn(t[k] - a[k]); stv[k] = sgn(t[k] - v[k]); }
    dpart[part][0][lane] = G(0, col) + g4 + g5 * a + g6 * sta + g7 + g8 * v + g9 * stv + g10;
    dpart[part][1][lane] = G(1, col) + g4 + g5 * t - g6 * sta;
    dpart[part][2][lane] = G(2, col) + g7 + g8 * t - g9 * stv + g11;
    dpart[part][3][lane] = G(3, col) + g10 + g11;
  } else {
    const f32x4 xs[4] = {ld4(c + col), ld4(c + H + col), ld4(c + 2 * H + col), ld4(c + 3 * H + col)};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      dob[b] = G(12 + b, col);
      const f32x4 val = ld4(q + VI[b] * H + col);
      r_dg[b] = dot4(dob[b], sv[b] * val - 0.5f * (xs[XS[b]] + xs[YS[b]]));
      r_ds[b] = dot4(dob[b], val);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) { r_dg[b] = wave_sum(r_dg[b]); r_ds[b] = wave_sum(r_ds[b]); }
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < 3; ++b) { red[part][b] = r_dg[b]; red[part][3 + b] = r_ds[b]; }
    }
  }
  __syncthreads();
  if (role == 1) return;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    float sg = red[0][b], ss = red[0][3 + b];
#pragma unroll
    for (int pp = 1; pp < NI; ++pp) { sg += red[pp][b]; ss += red[pp][3 + b]; }
    r_dg[b] = sg;
    r_ds[b] = ss;
  }
  const float inv = 1.0f / sqrtf((float)H);
  float dscore[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float dgate = r_dg[b];
    const float ds = gt[b] * r_ds[b];
    dscore[b] = ds * sv[b] * (1.0f - sv[b]) * inv;
    r_dg[b] = dgate * gt[b] * (1.0f - gt[b]);  // gradient at the gate's pre-sigmoid output
  }
  if (lane == 0 && part == 0) st4(dout + (size_t)row * 4, f32x4{r_dg[0], r_dg[1], r_dg[2], 0.f});
  float* dq = dqkv + (size_t)row * 9 * H;
  f32x4 d[4] = {dpart[part][0][lane], dpart[part][1][lane], dpart[part][2][lane], dpart[part][3][lane]};
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const f32x4 half = (0.5f * (1.0f - gt[b])) * dob[b];
    d[XS[b]] += half;
    d[YS[b]] += half;
    st4(dq + VI[b] * H + col, (gt[b] * sv[b]) * dob[b]);
    st4(dq + QI[b] * H + col, dscore[b] * ld4(q + KI[b] * H + col));
    st4(dq + KI[b] * H + col, dscore[b] * ld4(q + QI[b] * H + col));
  }
#pragma unroll
  for (int x = 0; x < 4; ++x) st4(dtavu + ((size_t)x * B + row) * H + col, d[x]);
  st4(dg + (size_t)row * H + col, G(15, col));
}

// evidence_proj parameter gradients: one block per co-attention block, thread j = hidden unit.  part == NULL: the whole batch in one
// pass (gridDim.y = 1).  part != NULL (larger batches): block (b, s) sums rows [s R, (s + 1) R) into part[s][b][5 H + 64] =
// [dw0 (H x 3) | db0 (H) | dw2 (H) | db2]; gate_param_finish_kernel adds the slices in ascending order.
__global__ void gate_param_kernel(const float* evid, const float* dout, int B, int H, EvPtrs p, EvGrads g, float* part, int R) {
  const int b = blockIdx.x;
  const int r0 = part ? blockIdx.y * R : 0, r1 = part ? (r0 + R < B ? r0 + R : B) : B;
  float* pb = part ? part + ((size_t)blockIdx.y * 3 + b) * (5 * H + 64) : nullptr;
  float db2 = 0.0f;
  for (int jj = threadIdx.x; jj < H; jj += blockDim.x) {
    const float w00 = p.w0[b][jj * 3], w01 = p.w0[b][jj * 3 + 1], w02 = p.w0[b][jj * 3 + 2], b0 = p.b0[b][jj],
                w2 = p.w2[b][jj];
    float dw0[3] = {0, 0, 0}, db0 = 0, dw2 = 0;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
      const f32x4 ev = ld4(evid + (size_t)r * 4);
      float e[3];
      if (b == 0) { e[0] = ev[0]; e[1] = ev[1]; e[2] = 0.f; }
      else if (b == 1) { e[0] = ev[1]; e[1] = 0.f; e[2] = 0.f; }
      else { e[0] = ev[2]; e[1] = 0.f; e[2] = 0.f; }
      const float pre = w00 * e[0] + w01 * e[1] + w02 * e[2] + b0;
      const float d_o = dout[(size_t)r * 4 + b];
      dw2 += d_o * gelu_f(pre);
      const float dpre = d_o * w2 * gelu_grad_f(pre);
      dw0[0] += dpre * e[0]; dw0[1] += dpre * e[1]; dw0[2] += dpre * e[2];
      db0 += dpre;
    }
    if (pb) {
      pb[jj * 3] = dw0[0]; pb[jj * 3 + 1] = dw0[1]; pb[jj * 3 + 2] = dw0[2];
      pb[3 * H + jj] = db0;
      pb[4 * H + jj] = dw2;
    } else {
      g.w0[b][jj * 3] = dw0[0]; g.w0[b][jj * 3 + 1] = dw0[1]; g.w0[b][jj * 3 + 2] = dw0[2];
      g.b0[b][jj] = db0;
      g.w2[b][jj] = dw2;
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll 8
    for (int r = r0; r < r1; ++r) db2 += dout[(size_t)r * 4 + b];
    if (pb) pb[5 * H] = db2;
    else g.b2[b][0] = db2;
  }
}
// Small batches (B <= 64, the one-pass form): ONE WAVE per (co-attention block b, hidden unit jj), lane r = batch row r -- the per-row
// erf / exp work of a unit runs side by side on the lanes and five DPP wave sums add the rows, instead of one thread per unit walking
// the rows one after the other on three workgroups (11.9 us at B = 32).  Rows beyond B contribute exact zeros.
__device__ __forceinline__ void gate_param_rows_block(const float* evid, const float* dout, int B, int H, const EvPtrs& p, const EvGrads& g, int b, int by) {
  const int jj = by * 4 + (threadIdx.x >> 6), r = threadIdx.x & 63;
  if (jj >= H) return;
  const float w00 = p.w0[b][jj * 3], w01 = p.w0[b][jj * 3 + 1], w02 = p.w0[b][jj * 3 + 2], b0 = p.b0[b][jj], w2 = p.w2[b][jj];
  float dw0[3] = {0, 0, 0}, db0 = 0, dw2 = 0, db2 = 0;
  if (r < B) {
    const f32x4 ev = ld4(evid + (size_t)r * 4);
    // selects, not an if / else-if / else chain on b: hipcc 7.2 left e[0] undefined on the b == 2 path of that chain in this kernel
    // (the register was only written on the other two paths; found by the golden gradient test, confirmed in the ISA)
    const float e[3] = {b == 0 ? ev[0] : (b == 1 ? ev[1] : ev[2]), b == 0 ? ev[1] : 0.f, 0.f};
    const float pre = w00 * e[0] + w01 * e[1] + w02 * e[2] + b0;
    const float d_o = dout[(size_t)r * 4 + b];
    dw2 = d_o * gelu_f(pre);
    const float dpre = d_o * w2 * gelu_grad_f(pre);
    dw0[0] = dpre * e[0]; dw0[1] = dpre * e[1]; dw0[2] = dpre * e[2];
    db0 = dpre;
    db2 = d_o;
  }
  dw0[0] = wave_sum(dw0[0]); dw0[1] = wave_sum(dw0[1]); dw0[2] = wave_sum(dw0[2]);
  db0 = wave_sum(db0); dw2 = wave_sum(dw2);
  if (jj == 0) db2 = wave_sum(db2);       // (wave-uniform branch)
  if (r == 0) {
    g.w0[b][jj * 3] = dw0[0]; g.w0[b][jj * 3 + 1] = dw0[1]; g.w0[b][jj * 3 + 2] = dw0[2];
    g.b0[b][jj] = db0;
    g.w2[b][jj] = dw2;
    if (jj == 0) g.b2[b][0] = db2;
  }
}
__global__ __launch_bounds__(256) void gate_param_rows_kernel(const float* evid, const float* dout, int B, int H, EvPtrs p, EvGrads g) {
  gate_param_rows_block(evid, dout, B, H, p, g, blockIdx.x, blockIdx.y);
}
// (fused head step: these blocks ride behind node_param_kernel's own, block e -> (b, by) = (e % 3, e / 3))
struct GateRows {
  const float *evid, *dout;
  EvPtrs p;
  EvGrads g;
  int first;      // the launch's first gate block; 0: no such role
};
// grid (3, ceil((5 H + 1) / 256)): one element per thread, its S slice loads all in flight (3 workgroups walking 5 H elements x S
// slices one dependent load at a time took 53 us at B = 256)
__global__ __launch_bounds__(256) void gate_param_finish_kernel(const float* part, int S, int H, EvGrads g) {
  const int b = blockIdx.x, n = 5 * H + 1, i = blockIdx.y * 256 + threadIdx.x;
  if (i >= n) return;
  const float* src = part + (size_t)b * (5 * H + 64) + i;
  const size_t stride = (size_t)3 * (5 * H + 64);
  float s = 0.0f;
  int k = 0;
  for (; k + 8 <= S; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];      // ascending slice order, as before
  }
  for (; k < S; ++k) s += src[(size_t)k * stride];
  if (i < 3 * H) g.w0[b][i] = s;
  else if (i < 4 * H) g.b0[b][i - 3 * H] = s;
  else if (i < 5 * H) g.w2[b][i - 4 * H] = s;
  else g.b2[b][0] = s;
}

// dZ = (dY + dlog @ Wc) * gelu'(Z) * dropmask        (entry of the fusion backward)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dY, int ldd, const float* dlog, const float* wc,
                                                      const float* Z, float* dZ, int M, int N, float drop_p,
                                                      uint32_t layer, const ufnd_step_state* st) {
  const size_t total = (size_t)M * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    float d = dY ? dY[(size_t)m * ldd + n] : 0.0f;
    if (dlog) d += dlog[m * 2] * wc[n] + dlog[m * 2 + 1] * wc[N + n];
    dZ[i] = d * gelu_grad_f(Z[i]) * dropout_mul(st, drop_p, layer, (uint32_t)i);
  }
}

// aux head: logits = fused Wc^T + bc (classes = 2); one wave per row
__global__ __launch_bounds__(256) void head2_fwd_kernel(const float* x, int ldx, const float* w, const float* b,
                                                        float* out, int B, int H) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wv;
  if (row >= B) return;
  float a0 = 0, a1 = 0;
  for (int c = lane; c < H; c += 64) {
    const float xv = x[(size_t)row * ldx + c];
    a0 += xv * w[c];
    a1 += xv * w[H + c];
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if (lane == 0) { out[row * 2] = a0 + b[0]; out[row * 2 + 1] = a1 + b[1]; }
}
// its parameter gradients; fused is recomputed from Z2 (gelu + dropout mask)
__global__ void head2_bwd_kernel(const float* dlog, const float* Z, int B, int H, float drop_p, uint32_t layer,
                                 const ufnd_step_state* st, float* dw, float* db) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < H; c += gridDim.x * blockDim.x) {
    float s0 = 0, s1 = 0;
#pragma unroll 8
    for (int r = 0; r < B; ++r) {
      const size_t i = (size_t)r * H + c;
      const float f = gelu_f(Z[i]) * dropout_mul(st, drop_p, layer, (uint32_t)i);
      s0 += dlog[r * 2] * f;
      s1 += dlog[r * 2 + 1] * f;
    }
    dw[c] = s0;
    dw[H + c] = s1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    float s = 0;
#pragma unroll 8
    for (int r = 0; r < B; ++r) s += dlog[r * 2 + threadIdx.x];
    db[threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------- classifier pieces
// NODE ensemble + bypass + temperature softmax        deep_truth_classifier.py:54-74,88-90,164-170
// One WORKGROUP per row (round 3; one wave per row before): its four waves take a quarter of the gates each (one batch of alpha
// rows in flight, then that wave's reductions), the bypass rows go to waves 0 and 1, the trees are dealt round-robin; the pieces
// meet in LDS.  Every gate, tree and class is still one wave's arithmetic in the same order, and the trees are added in
// ascending order at the end: the same bits as the one-wave form, a quarter of its chain.
template <int NI>
__global__ __launch_bounds__(256) void node_head_kernel(const float* hh, const float* alpha, const float* thresh,
                                                        const float* leaf, const float* tau, const float* bw,
                                                        const float* bb, const float* temperature, int B, int H,
                                                        int trees, int depth, float node_p,
                                                        const ufnd_step_state* st, float* fs, float* logits,
                                                        float* probs, const int64_t* labels, float* loss_rows, float* dlog) {
  __shared__ float s_lds[32], byp_lds[2], lg_lds[16][2];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;                 // (grid = B)
  f32x4 h[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) h[i] = ld4(hh + (size_t)row * H + 4 * lane + 256 * i);
  const int TK = trees * depth;
  // gates tk = w, w + 4, ... (at most 8 per wave: TK <= 32): their 8 * NI loads are independent and all in flight before the first
  // wave reduction
  {
    float p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int tk = (w + 4 * u < TK) ? w + 4 * u : TK - 1;
      float acc = 0;
#pragma unroll
      for (int i = 0; i < NI; ++i) acc += dot4(h[i], ld4(alpha + (size_t)tk * H + 4 * lane + 256 * i));
      p[u] = acc;
    }
    float bypass = 0;
    if (w < 2) {                              // (wave-uniform) bypass row w
#pragma unroll
      for (int i = 0; i < NI; ++i) bypass += dot4(h[i], ld4(bw + (size_t)w * H + 4 * lane + 256 * i));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int tk = w + 4 * u;
      const float f = wave_sum(p[u]);
      if (tk < TK && lane == 0) {
        const float sv = sigmoid_f(tau[tk / depth] * (f - thresh[tk]));
        s_lds[tk] = sv;
        fs[(size_t)row * 64 + tk] = sv;
      }
    }
    if (w < 2) {
      const float bsum = wave_sum(bypass) + bb[w];
      if (lane == 0) byp_lds[w] = bsum;
    }
  }
  __syncthreads();
  // the row's 2 * trees dropout multipliers: lane j evaluates element row * 2 * trees + j (every wave the same values)
  const float my_dm = lane < 2 * trees ? dropout_mul(st, node_p, LAYER_TREE, (uint32_t)(row * 2 * trees + lane)) : 1.0f;
  const int leaves = 1 << depth;
  for (int t = w; t < trees; t += 4) {
    float prob = 1.0f;
    for (int k = 0; k < depth; ++k) {
      const float sk = s_lds[t * depth + k];
      prob *= ((lane >> k) & 1) ? sk : (1.0f - sk);
    }
    if (lane >= leaves) prob = 0.0f;
    const float* lf = leaf + ((size_t)t * leaves + (lane < leaves ? lane : 0)) * 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float tl = wave_sum(prob * lf[c]);
      const float dm = __shfl(my_dm, t * 2 + c, 64);
      if (lane == 0) {
        lg_lds[t][c] = tl * dm;
        // backward reads the multiplier back instead of re-running Philox per (row, tree, class)
        fs[(size_t)row * 64 + 32 + t * 2 + c] = dm;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float lg[2] = {0.f, 0.f};
    for (int t = 0; t < trees; ++t) { lg[0] += lg_lds[t][0]; lg[1] += lg_lds[t][1]; }      // ascending tree order, as the one-wave form added them
    const float l0 = lg[0] / (float)trees + byp_lds[0], l1 = lg[1] / (float)trees + byp_lds[1];
    logits[row * 2] = l0;
    logits[row * 2 + 1] = l1;
    const float T = fminf(fmaxf(temperature[0], 0.5f), 5.0f);
    const float a0 = l0 / T, a1 = l1 / T, mx = fmaxf(a0, a1);
    const float e0 = __expf(a0 - mx), e1 = __expf(a1 - mx);
    probs[row * 2] = e0 / (e0 + e1);
    probs[row * 2 + 1] = e1 / (e0 + e1);
    if (labels) {      // fused head step: this row of F.cross_entropy (forensic_trainer.py:287), the expressions of softmax_ce_kernel
      const float cm = fmaxf(l0, l1);
      const float lse = cm + logf(__expf(l0 - cm) + __expf(l1 - cm));
      const int y = (int)labels[row];
      loss_rows[row] = lse - (y ? l1 : l0);
      dlog[row * 2] = (__expf(l0 - lse) - (y == 0 ? 1.f : 0.f)) / (float)B;
      dlog[row * 2 + 1] = (__expf(l1 - lse) - (y == 1 ? 1.f : 0.f)) / (float)B;
    }
  }
}

// backward through logits = mean_t(drop(probs_t @ leaf_t)) + bypass(h), down to dZ4
// One WORKGROUP per row (round 3; one wave per row before).  Wave 0: the gates' score gradients df (lanes < trees * depth, each
// walking its tree's leaves) and the bypass term of dh; waves 1-3: a third of the gates each -- their alpha rows are requested
// BEFORE the barrier that publishes df, so the loads fly under wave 0's leaf loops -- then d * alpha in ascending gate order.
// The four partial rows meet in LDS as (P0 + P1) + (P2 + P3); the GELU derivative, the dropout mask and the store follow on
// H / 4 threads.  (A fixed order, but not the one-wave form's: gradients agree to rounding, not bit for bit.)
// (fused head step: block 0 also writes the batch-mean loss from the CE rows node_head left -- the reduction of softmax_ce_kernel,
//  same order, same bits)
__device__ __forceinline__ void ce_mean_block(const float* loss_rows, int B, ufnd_step_state* st) {
  __shared__ float ce_sh[4];
  float part = 0;
  for (int r = threadIdx.x; r < B; r += 256) part += loss_rows[r];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) ce_sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) st->loss = ((ce_sh[0] + ce_sh[1]) + (ce_sh[2] + ce_sh[3])) / (float)B;
}
template <int NI>
__global__ __launch_bounds__(256) void node_bwd_kernel(const float* dlog, const float* fs, const float* alpha,
                                                       const float* leaf, const float* tau, const float* bw,
                                                       const float* z4, int B, int H, int trees, int depth,
                                                       float node_p, float clf_p, const ufnd_step_state* st,
                                                       float* df, float* dz4, const float* loss_rows, ufnd_step_state* st_loss) {
  __shared__ float df_lds[32];
  __shared__ f32x4 part[4][NI][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;                 // (grid = B)
  const float dl0 = dlog[row * 2], dl1 = dlog[row * 2 + 1];
  const int TK = trees * depth, leaves = 1 << depth;
  constexpr int GW = 11;                      // gates per wave of waves 1-3: ceil(32 / 3)
  f32x4 av[GW][NI];
  f32x4 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (w == 0) {
    {
      // lanes tk and tk + 32 share gate tk: the lower half of the wave walks the first half of the tree's leaves, the upper half the
      // second; the tree's gate scores are loaded once (depth <= 6), not per leaf
      const int tk = lane & 31, half = lane >> 5;
      const int tc = tk < TK ? tk : TK - 1;
      const int t = tc / depth, k = tc % depth;
      const float d0 = dl0 / (float)trees * fs[(size_t)row * 64 + 32 + t * 2];          // cached dropout multipliers (node_head)
      const float d1 = dl1 / (float)trees * fs[(size_t)row * 64 + 32 + t * 2 + 1];
      float sj[6];
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) sj[jj] = jj < depth ? fs[(size_t)row * 64 + t * depth + jj] : 0.0f;
      float ds = 0.0f;
      const int l_lo = half * (leaves >> 1), l_hi = l_lo + (leaves >> 1);      // (depth >= 1: leaves is even)
      for (int l = l_lo; l < l_hi; ++l) {
        float prod = 1.0f;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
          if (jj < depth && jj != k) prod *= ((l >> jj) & 1) ? sj[jj] : (1.0f - sj[jj]);
        }
        const f32x2 lf = *reinterpret_cast<const f32x2*>(leaf + ((size_t)t * leaves + l) * 2);
        const float dprob = lf[0] * d0 + lf[1] * d1;
        ds += (((l >> k) & 1) ? dprob : -dprob) * prod;
      }
      ds += __shfl_xor(ds, 32, 64);
      const float sk = fs[(size_t)row * 64 + tc];
      const float my_df = ds * tau[t] * sk * (1.0f - sk);
      if (lane < TK) {
        df[(size_t)row * 64 + lane] = my_df;
        df_lds[lane] = my_df;
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * lane + 256 * i;
      acc[i] = dl0 * ld4(bw + col) + dl1 * ld4(bw + H + col);
    }
  } else {
#pragma unroll
    for (int u = 0; u < GW; ++u) {
      const int tk = (w - 1) + 3 * u;
      const int tc = tk < TK ? tk : TK - 1;
#pragma unroll
      for (int i = 0; i < NI; ++i) av[u][i] = ld4(alpha + (size_t)tc * H + 4 * lane + 256 * i);
    }
  }
  __syncthreads();
  if (w != 0) {
#pragma unroll
    for (int u = 0; u < GW; ++u) {
      const int tk = (w - 1) + 3 * u;
      if (tk < TK) {                          // (wave-uniform)
        const float d = df_lds[tk];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += d * av[u][i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) part[w][i][lane] = acc[i];
  __syncthreads();
  for (int c4 = threadIdx.x; c4 < H / 4; c4 += 256) {
    const int i = c4 >> 6, ln = c4 & 63, col = 4 * c4;
    const f32x4 a = (part[0][i][ln] + part[1][i][ln]) + (part[2][i][ln] + part[3][i][ln]);
    const f32x4 z = ld4(z4 + (size_t)row * H + col);
    f32x4 o;
    float dm[4];
    dropout_mul4(st, clf_p, LAYER_PRE3, (uint32_t)(row * H + col), dm);      // (col and H are multiples of 4)
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = a[q] * gelu_grad_f(z[q]) * dm[q];
    st4(dz4 + (size_t)row * H + col, o);
  }
  if (loss_rows && row == 0) ce_mean_block(loss_rows, B, st_loss);      // (block-uniform)
}

// NODE / bypass parameter gradients.  blocks [0,TK): gate tk (+ its threshold); [TK,TK+2): bypass
// row c (+ bias); block TK+2: leaf tables.  Deterministic loops over rows.
__global__ __launch_bounds__(256) void node_param_kernel(const float* df, const float* dlog, const float* hh,
                                                         const float* alpha, const float* fs, int B, int H, int trees,
                                                         int depth, float node_p, const ufnd_step_state* st,
                                                         float* g_gates, float* g_thresh, float* g_leaf, float* g_bw,
                                                         float* g_bb, float* part, int R, GateRows gr) {
  if (gr.first > 0 && (int)blockIdx.x >= gr.first) {
    const int e = blockIdx.x - gr.first;
    gate_param_rows_block(gr.evid, gr.dout, B, H, gr.p, gr.g, e % 3, e / 3);
    return;
  }
  __shared__ float sh[4];
  const int TK = trees * depth, blk = blockIdx.x;
  if (part) {
    // row-sliced form (larger batches): block (blk, s) sums rows [s R, (s + 1) R) of the same products into
    // part[s] = [da TK x H | dthresh 64 | bypass 2 x H | bypass bias 64 | leaves]; node_param_finish_kernel adds the slices in order
    const int leaves = 1 << depth, r0 = blockIdx.y * R, r1 = r0 + R < B ? r0 + R : B;
    float* ps = part + (size_t)blockIdx.y * ((size_t)(TK + 2) * H + 128 + (size_t)trees * leaves * 2);
    if (blk < TK) {
      for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) s += df[(size_t)r * 64 + blk] * hh[(size_t)r * H + c];
        ps[(size_t)blk * H + c] = s;
      }
      if (threadIdx.x == 0) {
        float s = 0;
        for (int r = r0; r < r1; ++r) s += df[(size_t)r * 64 + blk];
        ps[(size_t)TK * H + blk] = s;
      }
    } else if (blk < TK + 2) {
      const int c2 = blk - TK;
      float* pw = ps + (size_t)TK * H + 64 + (size_t)c2 * H;
      for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) s += dlog[r * 2 + c2] * hh[(size_t)r * H + c];
        pw[c] = s;
      }
      if (threadIdx.x == 0) {
        float s = 0;
        for (int r = r0; r < r1; ++r) s += dlog[r * 2 + c2];
        ps[(size_t)(TK + 2) * H + 64 + c2] = s;
      }
    } else {
      float* pl = ps + (size_t)(TK + 2) * H + 128;
      for (int idx = threadIdx.x; idx < trees * leaves * 2; idx += 256) {
        const int c = idx & 1, l = (idx >> 1) % leaves, t = (idx >> 1) / leaves;
        float s = 0;
        for (int r = r0; r < r1; ++r) {
          float prob = 1.0f;
          for (int k = 0; k < depth; ++k) {
            const float sk = fs[(size_t)r * 64 + t * depth + k];
            prob *= ((l >> k) & 1) ? sk : (1.0f - sk);
          }
          s += prob * dlog[r * 2 + c] / (float)trees * fs[(size_t)r * 64 + 32 + t * 2 + c];
        }
        pl[idx] = s;
      }
    }
    return;
  }
  if (blk < TK) {
    float da[4] = {0, 0, 0, 0};
    float part = 0;
#pragma unroll
    for (int cnt = 0; cnt < 4; ++cnt) {          // H <= 1024: at most four columns per thread (static indices: registers, not scratch)
      const int c = threadIdx.x + 256 * cnt;
      if (c < H) {
        float s = 0;
#pragma unroll 8
        for (int r = 0; r < B; ++r) s += df[(size_t)r * 64 + blk] * hh[(size_t)r * H + c];
        da[cnt] = s;
        part += alpha[(size_t)blk * H + c] * s;
      }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
    __syncthreads();
    const float dotv = (sh[0] + sh[1]) + (sh[2] + sh[3]);
#pragma unroll
    for (int cnt = 0; cnt < 4; ++cnt) {
      const int c = threadIdx.x + 256 * cnt;
      if (c < H) g_gates[(size_t)blk * H + c] = alpha[(size_t)blk * H + c] * (da[cnt] - dotv);
    }
    if (threadIdx.x == 0) {
      float s = 0;
#pragma unroll 8
      for (int r = 0; r < B; ++r) s += df[(size_t)r * 64 + blk];
      g_thresh[blk] = -s;
    }
  } else if (blk < TK + 2) {
    const int c2 = blk - TK;
    for (int c = threadIdx.x; c < H; c += 256) {
      float s = 0;
#pragma unroll 8
      for (int r = 0; r < B; ++r) s += dlog[r * 2 + c2] * hh[(size_t)r * H + c];
      g_bw[(size_t)c2 * H + c] = s;
    }
    if (threadIdx.x == 0) {
      float s = 0;
#pragma unroll 8
      for (int r = 0; r < B; ++r) s += dlog[r * 2 + c2];
      g_bb[c2] = s;
    }
  } else {
    // leaf tables: workgroup (blk - TK - 2) owns eight table entries, 32 lanes each -- lane r adds rows r, r + 32, ... and the 32 lanes
    // meet in a fixed butterfly.  (One workgroup walking the rows of all 2 * trees * leaves entries, six dependent loads per row and
    // entry, was the longest chain of the whole launch: 14.8 us at B = 32.)
    const int leaves = 1 << depth, n_idx = trees * leaves * 2;
    const int idx = (blk - TK - 2) * 8 + ((int)threadIdx.x >> 5), r0 = threadIdx.x & 31;
    const bool live = idx < n_idx;
    const int ii = live ? idx : 0;
    const int c = ii & 1, l = (ii >> 1) % leaves, t = (ii >> 1) / leaves;
    float s = 0;
    for (int r = r0; r < B; r += 32) {
      float prob = 1.0f;
      for (int k = 0; k < depth; ++k) {
        const float sk = fs[(size_t)r * 64 + t * depth + k];
        prob *= ((l >> k) & 1) ? sk : (1.0f - sk);
      }
      s += prob * dlog[r * 2 + c] / (float)trees * fs[(size_t)r * 64 + 32 + t * 2 + c];
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);      // (stays inside each aligned group of 32 lanes)
    if (live && r0 == 0) g_leaf[idx] = s;
  }
}

// adds the row slices of node_param_kernel's partials in ascending order and applies the gate epilogue (alpha (da - alpha . da))
__global__ __launch_bounds__(256) void node_param_finish_kernel(const float* part, int S, const float* alpha, int H, int trees, int depth,
                                                                float* g_gates, float* g_thresh, float* g_leaf, float* g_bw, float* g_bb) {
  __shared__ float sh[4];
  const int TK = trees * depth, blk = blockIdx.x, leaves = 1 << depth;
  const size_t PS = (size_t)(TK + 2) * H + 128 + (size_t)trees * leaves * 2;
  auto total = [&](size_t off) {      // ascending slice order; eight slice loads in flight at a time
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= S; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * PS + off];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < S; ++k) s += part[(size_t)k * PS + off];
    return s;
  };
  if (blk < TK) {
    float da[4] = {0, 0, 0, 0}, dotp = 0;
#pragma unroll
    for (int cnt = 0; cnt < 4; ++cnt) {
      const int c = threadIdx.x + 256 * cnt;
      if (c < H) {
        da[cnt] = total((size_t)blk * H + c);
        dotp += alpha[(size_t)blk * H + c] * da[cnt];
      }
    }
    dotp = wave_sum(dotp);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dotp;
    __syncthreads();
    const float dotv = (sh[0] + sh[1]) + (sh[2] + sh[3]);
#pragma unroll
    for (int cnt = 0; cnt < 4; ++cnt) {
      const int c = threadIdx.x + 256 * cnt;
      if (c < H) g_gates[(size_t)blk * H + c] = alpha[(size_t)blk * H + c] * (da[cnt] - dotv);
    }
    if (threadIdx.x == 0) g_thresh[blk] = -total((size_t)TK * H + blk);
  } else if (blk < TK + 2) {
    const int c2 = blk - TK;
    for (int c = threadIdx.x; c < H; c += 256) g_bw[(size_t)c2 * H + c] = total((size_t)TK * H + 64 + (size_t)c2 * H + c);
    if (threadIdx.x == 0) g_bb[c2] = total((size_t)(TK + 2) * H + 64 + c2);
  } else {
    for (int idx = threadIdx.x; idx < trees * leaves * 2; idx += 256) g_leaf[idx] = total((size_t)(TK + 2) * H + 128 + idx);
  }
}

// ---------------------------------------------------------------- cross-entropy
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const int64_t* labels, int B,
                                                         float* loss_rows, float* dlog, ufnd_step_state* st) {
  __shared__ float sh[4];
  float part = 0;
  for (int r = threadIdx.x; r < B; r += 256) {
    const float l0 = logits[r * 2], l1 = logits[r * 2 + 1], mx = fmaxf(l0, l1);
    const float lse = mx + logf(__expf(l0 - mx) + __expf(l1 - mx));
    const int y = (int)labels[r];
    const float lr = lse - (y ? l1 : l0);
    if (loss_rows) loss_rows[r] = lr;
    part += lr;
    if (dlog) {
      dlog[r * 2] = (__expf(l0 - lse) - (y == 0 ? 1.f : 0.f)) / (float)B;
      dlog[r * 2 + 1] = (__expf(l1 - lse) - (y == 1 ? 1.f : 0.f)) / (float)B;
    }
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) st->loss = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)B;
}

// nn.CrossEntropyLoss(weight=w, label_smoothing=eps), mean reduction (forensic_trainer_integrated.py:166):
//   loss = sum_i [(1-eps) w[y_i] nll_i(y_i) + eps/2 sum_c w[c] nll_i(c)] / sum_i w[y_i]
//   dlogits[i][c] = [p_c ((1-eps) w[y_i] + eps/2 (w0 + w1)) - (1-eps) w[y_i] [c == y_i] - eps/2 w[c]] / sum_i w[y_i]
__global__ __launch_bounds__(256) void softmax_ce_ws_kernel(const float* logits, const int64_t* labels, int B, float w0, float w1,
                                                            float eps, float* loss_rows, float* dlog, ufnd_step_state* st) {
  __shared__ float sh[4], shw[4];
  float part = 0, wsum = 0;
  for (int r = threadIdx.x; r < B; r += 256) wsum += labels[r] ? w1 : w0;
  wsum = wave_sum(wsum);
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = wsum;
  __syncthreads();
  const float W = (shw[0] + shw[1]) + (shw[2] + shw[3]);
  for (int r = threadIdx.x; r < B; r += 256) {
    const float l0 = logits[r * 2], l1 = logits[r * 2 + 1], mx = fmaxf(l0, l1);
    const float lse = mx + logf(__expf(l0 - mx) + __expf(l1 - mx));
    const int y = (int)labels[r];
    const float wy = y ? w1 : w0;
    const float n0 = lse - l0, n1 = lse - l1;                       // -log p_c
    const float lr = (1.0f - eps) * wy * (y ? n1 : n0) + 0.5f * eps * (w0 * n0 + w1 * n1);
    if (loss_rows) loss_rows[r] = lr / W;
    part += lr;
    if (dlog) {
      const float k = (1.0f - eps) * wy + 0.5f * eps * (w0 + w1);
      dlog[r * 2] = (__expf(l0 - lse) * k - (y == 0 ? (1.0f - eps) * wy : 0.f) - 0.5f * eps * w0) / W;
      dlog[r * 2 + 1] = (__expf(l1 - lse) * k - (y == 1 ? (1.0f - eps) * wy : 0.f) - 0.5f * eps * w1) / W;
    }
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) st->loss = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / W;
}

int check_dims(const ufnd_dims* d, int B) {
  UFND_REQUIRE(d, "dims is null");
  UFND_REQUIRE(d->hidden == 256 || d->hidden == 512 || d->hidden == 1024, "hidden=%d: supported 256/512/1024", d->hidden);
  UFND_REQUIRE(d->classes == 2, "classes=%d: only 2 supported", d->classes);
  UFND_REQUIRE(d->depth >= 1 && d->depth <= 6 && d->trees >= 1 && d->trees * d->depth <= 32 && d->trees <= 16, "trees=%d depth=%d unsupported",
               d->trees, d->depth);   // (a row's 64-float slot: 32 gate activations + 2 dropout multipliers per tree)
  UFND_REQUIRE(d->aux_dim >= 0 && d->aux_dim <= 4 && d->aux_dim % 2 == 0, "aux_dim=%d: supported 0, 2, 4", d->aux_dim);
  UFND_REQUIRE(d->text_dim % 4 == 0 && d->audio_dim % 4 == 0 && d->visual_dim % 4 == 0 && d->temporal_dim % 4 == 0 &&
                   d->gnn_dim % 4 == 0, "input feature dims must be multiples of 4");
  UFND_REQUIRE(B >= 1 && B <= 65536, "B=%d out of range", B);
  return UFND_OK;
}

#define NI_DISPATCH(H, KERNEL, GRID, BLOCK, STREAM, ...)                                    \
  do {                                                                                      \
    if ((H) == 256) hipLaunchKernelGGL((KERNEL<1>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);   \
    else if ((H) == 512) hipLaunchKernelGGL((KERNEL<2>), GRID, BLOCK, 0, STREAM, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<4>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);              \
  } while (0)

// fork/join between the caller's stream (the dX chain: the critical path of backward) and an optional
// side stream that takes every dW / parameter-gradient kernel.  Events are created once per thread.
struct ForkJoin {
  hipStream_t main, side;
  bool on() const { return side != nullptr; }
  static hipEvent_t event(int i) {
    static thread_local hipEvent_t ev[8];
    static thread_local bool init = false;
    if (!init) {
      for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
      init = true;
    }
    return ev[i & 7];
  }
  hipStream_t dw() const { return side ? side : main; }
  void fork(int i) const {   // side stream may use everything enqueued on main so far
    if (!side) return;
    hipEventRecord(event(i), main);
    hipStreamWaitEvent(side, event(i), 0);
  }
  void join(int i) const {   // main continues after everything enqueued on side so far
    if (!side) return;
    hipEventRecord(event(i), side);
    hipStreamWaitEvent(main, event(i), 0);
  }
};

#define TRY(x)                 \
  do {                         \
    int rc_ = (x);             \
    if (rc_ != UFND_OK) return rc_; \
  } while (0)

}  // namespace

// =============================================================================================
extern "C" size_t ufnd_fusion_workspace_floats(const ufnd_dims* d, int B) {
  if (!d || B < 1) return 0;
  return carve_fusion(*d, B, nullptr).total;
}
extern "C" size_t ufnd_clf_workspace_floats(const ufnd_dims* d, int B) {
  if (!d || B < 1) return 0;
  return carve_clf(*d, B, nullptr).total;
}
extern "C" float* ufnd_clf_input_panel(const ufnd_dims* d, float* ws, int B, int* ld) {
  if (!d || !ws) return nullptr;
  ClfWs w = carve_clf(*d, B, ws);
  if (ld) *ld = w.ldx;
  return w.xin;
}

namespace {
int fusion_forward_impl(const ufnd_dims* d, const ufnd_fusion_params* p, const float* text,
                        const float* audio, const float* visual, const float* temporal, const float* gnn,
                        int B, int train, float* workspace, float* fused, int ld_fused, float* logits,
                        float* forensic, const ufnd_step_state* state, void* stream_, const ClfPrep* prep_in) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(p && text && audio && visual && temporal && workspace && fused && forensic && state, "fusion_forward: null argument");
  UFND_REQUIRE(gnn || d->gnn_dim == 0, "fusion_forward: gnn_feat is required when dims.gnn_dim > 0 (fuse_mlp then expects the 16*hidden concat, "
               "cross_modal_transformer.py:184-195); gnn_dim == 0 is fusion.yaml's `use_gnn: false` (15*hidden, no gnn_proj)");
  UFND_REQUIRE(ld_fused % 4 == 0 && ld_fused >= d->hidden && ufnd_aligned(fused, 16) && ufnd_aligned(workspace, 256),
               "fusion_forward: fused/workspace alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const int H = d->hidden;
  FusionWs w = carve_fusion(*d, B, workspace);
  const float drop = train ? d->fusion_dropout : 0.0f;
  const dim3 rows(ufnd_cdiv(B, 4)), blk(256);

  // fusion.yaml `use_gnn: false` (dims.gnn_dim == 0): no gnn_proj, the concat is the first 15 slots (:88,101-102,119-120)
  const int NPROJ = d->gnn_dim > 0 ? 5 : 4, CW = (d->gnn_dim > 0 ? 16 : 15) * H;
  // 1. the projections straight into their CAT slots                          (:147-150,184-187)
  {
    const float* xs[5] = {text, audio, visual, temporal, gnn};
    const float* ws_[5] = {p->text_w, p->audio_w, p->visual_w, p->temporal_w, p->gnn_w};
    const float* bs[5] = {p->text_b, p->audio_b, p->visual_b, p->temporal_b, p->gnn_b};
    const int ks[5] = {d->text_dim, d->audio_dim, d->visual_dim, d->temporal_dim, d->gnn_dim};
    const int slot[5] = {0, 1, 2, 3, 15};
    NtProb pr[5];
    for (int i = 0; i < NPROJ; ++i)
      pr[i] = NtProb{xs[i], ws_[i], bs[i], w.cat + (size_t)slot[i] * H, nullptr, B, H, ks[i], ks[i], ks[i], 16 * H, 0,
                     0, 0.0f, 0, 1};
    TRY(launch_nt(pr, NPROJ, state, stream));
  }
  // 2. stacked q/k/v projections: t -> [q_tv q_ta], v -> [k_tv v_tv q_vu], a -> [k_ta v_ta], u -> [k_vu v_vu]
  {
    const int src_slot[4] = {0, 2, 1, 3}, row0[4] = {0, 2, 5, 7}, nrows[4] = {2, 3, 2, 2};
    NtProb pr[4];
    for (int i = 0; i < 4; ++i)
      pr[i] = NtProb{w.cat + (size_t)src_slot[i] * H, p->qkv_w + (size_t)row0[i] * H * H, p->qkv_b + (size_t)row0[i] * H,
                     w.qkv + (size_t)row0[i] * H, nullptr, B, nrows[i] * H, H, 16 * H, H, 9 * H, 0, 0, 0.0f, 0, 1};
    TRY(launch_nt(pr, 4, state, stream));
  }
  // 3. evidence scalars + the three evidence gates (:153-164,48), co-attention combine + pairwise features -> CAT slots
  //    4..14 (:44-54,172-178): one row kernel
  EvPtrs ev;
  for (int b = 0; b < 3; ++b) { ev.w0[b] = p->ev0_w[b]; ev.b0[b] = p->ev0_b[b]; ev.w2[b] = p->ev2_w[b]; ev.b2[b] = p->ev2_b[b]; }
  ClfPrep prep{};
  if (prep_in) prep = *prep_in;               // (fused head step: + B + trees * depth blocks preparing the classifier's input)
  const dim3 cgrid(B + prep.blocks);
  if (H == 256) hipLaunchKernelGGL((coattn_pairs_wg_kernel<1>), cgrid, blk, 0, stream, w.cat, (const float*)w.qkv, w.gate, B, H, w.s, ev, w.evid, forensic, prep);
  else if (H == 512) hipLaunchKernelGGL((coattn_pairs_wg_kernel<2>), cgrid, blk, 0, stream, w.cat, (const float*)w.qkv, w.gate, B, H, w.s, ev, w.evid, forensic, prep);
  else hipLaunchKernelGGL((coattn_pairs_wg_kernel<4>), cgrid, blk, 0, stream, w.cat, (const float*)w.qkv, w.gate, B, H, w.s, ev, w.evid, forensic, prep);
  UFND_CHECK_LAUNCH();
  // 4. fuse_mlp.0: (B,16H) x (2H,16H)^T, split-K then bias+GELU(+dropout)             (:122-124)
  {
    NtProb pr{w.cat, p->fuse0_w, nullptr, w.part1, nullptr, B, 2 * H, CW, 16 * H, CW, 2 * H, 0, 0, 0.0f, 0,
              KSPLIT_FUSE0};
    TRY(launch_nt(&pr, 1, state, stream));
    const int n4 = B * 2 * H / 4;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(ufnd_cdiv(n4, 256)), blk, 0, stream, (const float*)w.part1,
                       KSPLIT_FUSE0, B, 2 * H, (const float*)p->fuse0_b, w.z1, w.h1, 1, drop, LAYER_FUSE0, state);
    UFND_CHECK_LAUNCH();
  }
  // 5. fuse_mlp.3 + GELU(+dropout) -> fused                                           (:125-127)
  {
    NtProb pr{w.h1, p->fuse3_w, p->fuse3_b, fused, w.z2, B, H, 2 * H, 2 * H, 2 * H, ld_fused, H, 1, drop, LAYER_FUSE3, 1};
    TRY(launch_nt(&pr, 1, state, stream));
  }
  // 6. aux head (unused by the trainer's loss, :198)
  if (logits) {
    hipLaunchKernelGGL(head2_fwd_kernel, rows, blk, 0, stream, (const float*)fused, ld_fused, (const float*)p->cls_w,
                       (const float*)p->cls_b, logits, B, H);
    UFND_CHECK_LAUNCH();
  }
  return UFND_OK;
}
}  // namespace

extern "C" int ufnd_fusion_forward(const ufnd_dims* d, const ufnd_fusion_params* p, const float* text,
                                   const float* audio, const float* visual, const float* temporal, const float* gnn,
                                   int B, int train, float* workspace, float* fused, int ld_fused, float* logits,
                                   float* forensic, const ufnd_step_state* state, void* stream_) {
  return fusion_forward_impl(d, p, text, audio, visual, temporal, gnn, B, train, workspace, fused, ld_fused, logits, forensic, state, stream_, nullptr);
}

// ------------------------------------------------------------------------------------------------
// The Linear gradients of the head as dW = dY^T X problems over PANELS.  The backward passes its own workspace panels (one
// contiguous panel of B rows each); the factor form of the data-parallel exchange passes every rank's packed panels
// (ufnd_head_pack_factors) as row segments and forms the sum over ALL ranks' rows in one pass
// (ufnd_head_linear_grads_from_factors): seg = {rows per rank, floats between two ranks' packs}.
// ------------------------------------------------------------------------------------------------
namespace {
struct Seg { int rows, stride; };
struct FusionFactors {
  const float *dz2, *h1, *dz1, *cat, *dqkv;
  const float *dproj[5], *x[5];      // gradients at the [text audio visual temporal gnn] projection outputs; the projections' inputs
};
FusionFactors fusion_factors(const FusionWs& w, size_t B, size_t H, const float* text, const float* audio, const float* visual,
                             const float* temporal, const float* gnn) {
  // (dtavu order is [t a v u] = [text audio visual temporal])
  return FusionFactors{w.dz2, w.h1, w.dz1, w.cat, w.dqkv, {w.dtavu, w.dtavu + B * H, w.dtavu + 2 * B * H, w.dtavu + 3 * B * H, w.dg},
                       {text, audio, visual, temporal, gnn}};
}
inline TnProb tn_prob(const float* dY, const float* X, float* dW, float* db, int M, int N, int K, int lddy, int ldx, int ldw, Seg sg) {
  return TnProb{dY, X, dW, db, M, N, K, lddy, ldx, ldw, sg.rows, sg.stride, sg.stride};
}
// fuse_mlp.3 and fuse_mlp.0 (the 33.5 MB gradient): 2 problems
int fusion_tn_fuse(const ufnd_dims& d, const FusionFactors& f, const ufnd_fusion_params* g, int M, TnProb* tn, Seg sg = Seg{0, 0}) {
  const int H = d.hidden, CW = (d.gnn_dim > 0 ? 16 : 15) * H;      // (use_gnn: false: 15 slots; CAT keeps its 16H row stride)
  tn[0] = tn_prob(f.dz2, f.h1, g->fuse3_w, g->fuse3_b, M, H, 2 * H, H, 2 * H, 2 * H, sg);
  tn[1] = tn_prob(f.dz1, f.cat, g->fuse0_w, g->fuse0_b, M, 2 * H, CW, 2 * H, 16 * H, CW, sg);
  return 2;
}
// stacked q/k/v (t -> [q_tv q_ta], v -> [k_tv v_tv q_vu], a -> [k_ta v_ta], u -> [k_vu v_vu]) and the projections: 4 + (5 | 4) problems
int fusion_tn_rest(const ufnd_dims& d, const FusionFactors& f, const ufnd_fusion_params* g, int M, TnProb* tn, Seg sg = Seg{0, 0}) {
  const int H = d.hidden;
  const int src_slot[4] = {0, 2, 1, 3}, row0[4] = {0, 2, 5, 7}, nrows[4] = {2, 3, 2, 2};
  int n = 0;
  for (int i = 0; i < 4; ++i)
    tn[n++] = tn_prob(f.dqkv + (size_t)row0[i] * H, f.cat + (size_t)src_slot[i] * H, g->qkv_w + (size_t)row0[i] * H * H,
                      g->qkv_b + (size_t)row0[i] * H, M, nrows[i] * H, H, 9 * H, 16 * H, H, sg);
  float* gw[5] = {g->text_w, g->audio_w, g->visual_w, g->temporal_w, g->gnn_w};
  float* gb[5] = {g->text_b, g->audio_b, g->visual_b, g->temporal_b, g->gnn_b};
  const int ks[5] = {d.text_dim, d.audio_dim, d.visual_dim, d.temporal_dim, d.gnn_dim};
  for (int i = 0; i < (d.gnn_dim > 0 ? 5 : 4); ++i) tn[n++] = tn_prob(f.dproj[i], f.x[i], gw[i], gb[i], M, H, ks[i], H, ks[i], ks[i], sg);
  return n;
}
// classifier pre.3 and pre.0 (over the full (hidden + aux) width): 2 problems
struct ClfFactors { const float *dz4, *h3, *dz3, *xin; int ldx; };
int clf_tn(const ufnd_dims& d, const ClfFactors& f, const ufnd_clf_params* g, int M, TnProb* tn, Seg sg = Seg{0, 0}) {
  const int H = d.hidden;
  tn[0] = tn_prob(f.dz4, f.h3, g->pre3_w, g->pre3_b, M, H, H, H, H, H, sg);
  tn[1] = tn_prob(f.dz3, f.xin, g->pre0_w, g->pre0_b, M, H, H + d.aux_dim, H, f.ldx, H + d.aux_dim, sg);
  return 2;
}
}  // namespace

namespace {
// The classifier's parameter-gradient launch, handed to the fusion backward by the fused head step: it is issued there together with
// the evidence-gate parameter gradients, as ONE launch (node_param_kernel's blocks, then gate_param_rows' blocks).
struct NodeParamJob {
  const float *df, *dlog, *hh, *alpha, *fs;
  int trees, depth;
  float node_p;
  float *g_gates, *g_thresh, *g_leaf, *g_bw, *g_bb;
};
struct FusedBwd {
  bool dz2_ready;             // the classifier's last dX product has written dZ2 (activation derivative and mask applied): no act_bwd launch
  const NodeParamJob* np;     // non-null: see above (B <= 64 only)
  const TnProb* extra_tn;     // the classifier's two dW / db problems, to join this module's grouped launch (whole backward only)
  int n_extra;
};
int fusion_backward_impl(const ufnd_dims* d, const ufnd_fusion_params* p, const ufnd_fusion_params* g,
                         const float* text, const float* audio, const float* visual, const float* temporal,
                         const float* gnn, int B, int train, float* workspace, const float* d_fused,
                         int ld_dfused, const float* d_logits, const ufnd_step_state* state, void* stream_,
                         void* side_stream_, int join, int phase, const FusedBwd* fo) {
  TRY(check_dims(d, B));
  // (factor form of the data-parallel exchange: the Linear dW / db products are left to ufnd_head_linear_grads_from_factors)
  const bool linear = !(phase & UFND_BWD_NO_LINEAR_GRADS);
  phase &= ~UFND_BWD_NO_LINEAR_GRADS;
  UFND_REQUIRE(phase == UFND_BWD_ALL || phase == UFND_BWD_FUSE_MLP || phase == UFND_BWD_REST, "fusion_backward: phase=%d", phase);
  const bool do_head = phase != UFND_BWD_REST, do_rest = phase != UFND_BWD_FUSE_MLP;
  UFND_REQUIRE(p && g && text && audio && visual && temporal && (gnn || (d && d->gnn_dim == 0)) && workspace && state, "fusion_backward: null argument");
  const bool dz2_ready = fo && fo->dz2_ready;
  UFND_REQUIRE(!do_head || d_fused || d_logits || dz2_ready, "fusion_backward: no incoming gradient");
  UFND_REQUIRE(!dz2_ready || !d_logits, "fusion_backward: a fused dZ2 and an aux-head gradient are exclusive");
  hipStream_t stream = (hipStream_t)stream_;
  const ForkJoin fj{stream, (hipStream_t)side_stream_};
  const int H = d->hidden;
  FusionWs w = carve_fusion(*d, B, workspace);
  const float drop = train ? d->fusion_dropout : 0.0f;
  const dim3 rows(ufnd_cdiv(B, 4)), blk(256);
  // Every dW = dY^T X product of this module is independent of the others once the dX chain has
  // produced the dY's: they are collected and issued as ONE grouped launch at the end (11 problems,
  // ~50 MB of gradient written by ~2.9k wave tiles) instead of five small launches.
  TnProb tn[UFND_GEMM_MAX_PROB];
  int ntn = 0;

  if (do_head) {
  // aux-head parameter grads (only when a gradient arrives at the aux logits)
  if (d_logits) {
    hipLaunchKernelGGL(head2_bwd_kernel, dim3(ufnd_cdiv(H, 256)), blk, 0, stream, d_logits, (const float*)w.z2, B, H,
                       drop, LAYER_FUSE3, state, g->cls_w, g->cls_b);
    UFND_CHECK_LAUNCH();
  }
  // dZ2 = (d_fused + d_logits Wc) * gelu'(Z2) * mask
  if (!dz2_ready)
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ufnd_cdiv(B * H, 256)), blk, 0, stream, d_fused, ld_dfused, d_logits,
                     (const float*)p->cls_w, (const float*)w.z2, w.dz2, B, H, drop, LAYER_FUSE3, state);
  UFND_CHECK_LAUNCH();
  // fuse_mlp.3: dW, db; dH1 -> dZ1 (epilogue applies gelu'(Z1) * mask)
  {
    NnProb n{w.dz2, p->fuse3_w, w.dz1, w.z1, nullptr, B, H, 2 * H, H, 2 * H, 2 * H, 2 * H, 0, drop, LAYER_FUSE0, 2 * H, 1};
    TRY(launch_nn(&n, 1, state, stream));
  }
  // fuse_mlp.0: dW (the 33.5 MB gradient), db; dCAT partials
  {
    const int CW = (d->gnn_dim > 0 ? 16 : 15) * H;      // (use_gnn: false: 15 slots; CAT / dCAT keep their 16H row stride)
    ntn += fusion_tn_fuse(*d, fusion_factors(w, B, H, text, audio, visual, temporal, gnn), g, B, tn + ntn);
    NnProb n{w.dz1, p->fuse0_w, w.dcatp, nullptr, nullptr, B, 2 * H, CW, 2 * H, CW, 16 * H, 0, 0, 0.0f, 0, 0,
             NSPLIT_FUSE0};
    if (phase == UFND_BWD_FUSE_MLP && linear) {
      // bucketed gradient exchange: the two fuse_mlp weight gradients (67 % of all gradient bytes) are written NOW,
      // ahead of the dCAT product, so that the caller can start reducing them while the rest of backward runs
      // (the grouped launch computes every problem independently: the same bits as the one-launch form)
      fj.fork(1);
      TRY(launch_tn(tn, ntn, fj.dw()));
      ntn = 0;
    }
    TRY(launch_nn(&n, 1, state, stream));
  }
  }  // do_head
  if (do_rest) {
  // concat / pairwise / co-attention backward (row-wise)
  // one workgroup of 2 * (H / 256) waves per row
  if (H == 256) hipLaunchKernelGGL((coattn_pairs_bwd_wg_kernel<1, NSPLIT_FUSE0>), dim3(B), dim3(128), 0, stream, (const float*)w.dcatp, (const float*)w.cat,
                                   (const float*)w.qkv, (const float*)w.gate, (const float*)w.s, B, H, w.dtavu, w.dqkv, w.dg, w.dout);
  else if (H == 512) hipLaunchKernelGGL((coattn_pairs_bwd_wg_kernel<2, NSPLIT_FUSE0>), dim3(B), dim3(256), 0, stream, (const float*)w.dcatp, (const float*)w.cat,
                                        (const float*)w.qkv, (const float*)w.gate, (const float*)w.s, B, H, w.dtavu, w.dqkv, w.dg, w.dout);
  else hipLaunchKernelGGL((coattn_pairs_bwd_wg_kernel<4, NSPLIT_FUSE0>), dim3(B), dim3(512), 0, stream, (const float*)w.dcatp, (const float*)w.cat,
                          (const float*)w.qkv, (const float*)w.gate, (const float*)w.s, B, H, w.dtavu, w.dqkv, w.dg, w.dout);
  UFND_CHECK_LAUNCH();
  // evidence_proj parameter grads
  {
    EvPtrs ep;
    EvGrads eg;
    for (int b = 0; b < 3; ++b) {
      ep.w0[b] = p->ev0_w[b]; ep.b0[b] = p->ev0_b[b]; ep.w2[b] = p->ev2_w[b]; ep.b2[b] = p->ev2_b[b];
      eg.w0[b] = g->ev0_w[b]; eg.b0[b] = g->ev0_b[b]; eg.w2[b] = g->ev2_w[b]; eg.b2[b] = g->ev2_b[b];
    }
    fj.fork(2);   // dout, dqkv, dg are ready
    const int S = param_slices(B);
    if (B <= 64 && fo && fo->np) {
      // fused head step: the classifier's parameter gradients (ready since node_bwd) and these in ONE launch
      const NodeParamJob& j = *fo->np;
      const int nodeb = j.trees * j.depth + 2 + ufnd_cdiv(j.trees * (1 << j.depth) * 2, 8);
      GateRows gr{(const float*)w.evid, (const float*)w.dout, ep, eg, nodeb};
      hipLaunchKernelGGL(node_param_kernel, dim3(nodeb + 3 * ufnd_cdiv(H, 4)), dim3(256), 0, fj.dw(), j.df, j.dlog, j.hh, j.alpha, j.fs, B, H, j.trees, j.depth,
                         j.node_p, state, j.g_gates, j.g_thresh, j.g_leaf, j.g_bw, j.g_bb, (float*)nullptr, B, gr);
    } else if (B <= 64) {
      hipLaunchKernelGGL(gate_param_rows_kernel, dim3(3, ufnd_cdiv(H, 4)), dim3(256), 0, fj.dw(), (const float*)w.evid, (const float*)w.dout, B, H, ep, eg);
    } else {
      hipLaunchKernelGGL(gate_param_kernel, dim3(3, S), dim3(H > 512 ? 512 : H), 0, fj.dw(), (const float*)w.evid,
                         (const float*)w.dout, B, H, ep, eg, S > 1 ? w.gpart : (float*)nullptr, (B + S - 1) / S);
    }
    UFND_CHECK_LAUNCH();
    if (S > 1) {
      hipLaunchKernelGGL(gate_param_finish_kernel, dim3(3, ufnd_cdiv(5 * H + 1, 256)), dim3(256), 0, fj.dw(), (const float*)w.gpart, S, H, eg);
      UFND_CHECK_LAUNCH();
    }
  }
  // stacked q/k/v: dW, db, and the extra gradient into t, v, a, u (accumulated in place)
  {
    const int src_slot[4] = {0, 2, 1, 3}, row0[4] = {0, 2, 5, 7}, nrows[4] = {2, 3, 2, 2};
    NnProb n[4];
    for (int i = 0; i < 4; ++i) {
      float* dst = w.dtavu + (size_t)src_slot[i] * B * H;  // dtavu order is [t a v u]
      n[i] = NnProb{w.dqkv + (size_t)row0[i] * H, p->qkv_w + (size_t)row0[i] * H * H, dst, nullptr, dst, B, nrows[i] * H,
                    H, 9 * H, H, H, 0, H, 0.0f, 0, 0, 1};
    }
    TRY(launch_nn(n, 4, state, stream));
    fj.fork(3);   // every dY of the module is ready
  }
  // stacked q/k/v and projections: dW, db (the projections' inputs are data: no dX)
  ntn += fusion_tn_rest(*d, fusion_factors(w, B, H, text, audio, visual, temporal, gnn), g, B, tn + ntn);
  if (fo && fo->extra_tn)
    for (int i = 0; i < fo->n_extra; ++i) tn[ntn++] = fo->extra_tn[i];
  if (linear) TRY(launch_tn(tn, ntn, fj.dw()));
  }  // do_rest
  if (join) fj.join(4);
  return UFND_OK;
}
}  // namespace

extern "C" int ufnd_fusion_backward_phase(const ufnd_dims* d, const ufnd_fusion_params* p, const ufnd_fusion_params* g,
                                          const float* text, const float* audio, const float* visual, const float* temporal,
                                          const float* gnn, int B, int train, float* workspace, const float* d_fused,
                                          int ld_dfused, const float* d_logits, const ufnd_step_state* state, void* stream_,
                                          void* side_stream_, int join, int phase) {
  return fusion_backward_impl(d, p, g, text, audio, visual, temporal, gnn, B, train, workspace, d_fused, ld_dfused, d_logits, state, stream_,
                              side_stream_, join, phase, nullptr);
}

extern "C" int ufnd_fusion_backward(const ufnd_dims* d, const ufnd_fusion_params* p, const ufnd_fusion_params* g,
                                    const float* text, const float* audio, const float* visual, const float* temporal,
                                    const float* gnn, int B, int train, float* workspace, const float* d_fused,
                                    int ld_dfused, const float* d_logits, const ufnd_step_state* state, void* stream_, void* side_stream_,
                                    int join) {
  return ufnd_fusion_backward_phase(d, p, g, text, audio, visual, temporal, gnn, B, train, workspace, d_fused, ld_dfused, d_logits,
                                    state, stream_, side_stream_, join, UFND_BWD_ALL);
}

// d loss / d gnn_feat = dg . gnn_proj.weight, from the workspace a fusion backward has just filled (the gradient at the
// gnn_proj output lives there).  The main trainer's gnn_feat is data (a detached table, forensic_trainer.py:209-211); the
// integrated variant's comes from a GNN inside the graph (forensic_trainer_integrated.py:203-224) and needs this.
extern "C" int ufnd_fusion_gnn_input_grad(const ufnd_dims* d, const ufnd_fusion_params* p, float* workspace, int B, float* d_gnn,
                                          const ufnd_step_state* state, void* stream_) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(p && workspace && d_gnn && ufnd_aligned(d_gnn, 16), "fusion_gnn_input_grad: null argument");
  FusionWs w = carve_fusion(*d, B, workspace);
  const int H = d->hidden;
  NnProb n{w.dg, p->gnn_w, d_gnn, nullptr, nullptr, B, H, d->gnn_dim, H, d->gnn_dim, d->gnn_dim, 0, 0, 0.0f, 0, 0, 1};
  return launch_nn(&n, 1, state, (hipStream_t)stream_);
}

// d loss / d text_features = dt . text_proj.weight and d loss / d visual_features = dv . visual_proj.weight, from the workspace a
// fusion backward has just filled.  In the reference trainer these inputs are cached data (forensic_trainer.py:60-83); with
// trainable encoders (TrainConfig.train_encoders) they are the encoders' outputs and this is where their backward starts.
extern "C" int ufnd_fusion_feature_grads(const ufnd_dims* d, const ufnd_fusion_params* p, float* workspace, int B, float* d_text,
                                         float* d_visual, const ufnd_step_state* state, void* stream_) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(p && workspace && (d_text || d_visual), "fusion_feature_grads: null argument");
  UFND_REQUIRE((!d_text || ufnd_aligned(d_text, 16)) && (!d_visual || ufnd_aligned(d_visual, 16)), "fusion_feature_grads: alignment");
  FusionWs w = carve_fusion(*d, B, workspace);
  const int H = d->hidden;
  NnProb n[2];
  int k = 0;
  if (d_text) n[k++] = NnProb{w.dtavu, p->text_w, d_text, nullptr, nullptr, B, H, d->text_dim, H, d->text_dim, d->text_dim, 0, 0, 0.0f, 0, 0, 1};
  if (d_visual)
    n[k++] = NnProb{w.dtavu + (size_t)2 * B * H, p->visual_w, d_visual, nullptr, nullptr, B, H, d->visual_dim, H, d->visual_dim, d->visual_dim, 0, 0, 0.0f, 0, 0, 1};
  return launch_nn(n, k, state, (hipStream_t)stream_);
}

namespace {
struct FusedClfFwd {
  bool prep_done;             // the input panel's aux columns and alpha were written by the co-attention launch
  const int64_t* labels;      // non-null: node_head also writes this row's CE term (workspace) and d_logits
  float* d_logits;
};
int classifier_forward_impl(const ufnd_dims* d, const ufnd_clf_params* p, const float* fused, int ld_fused,
                            const float* aux, int B, int train, float* workspace, float* logits,
                            float* probs, const ufnd_step_state* state, void* stream_, const FusedClfFwd* fo) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(p && fused && workspace && logits && probs && state, "classifier_forward: null argument");
  UFND_REQUIRE((d->aux_dim == 0) == (aux == nullptr), "classifier_forward: aux must be given iff aux_dim > 0 (pre.0 is %d wide)",
               d->hidden + d->aux_dim);
  UFND_REQUIRE(ufnd_aligned(workspace, 256), "classifier_forward: workspace alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const int H = d->hidden;
  ClfWs w = carve_clf(*d, B, workspace);
  const float drop = train ? d->clf_dropout : 0.0f;
  const dim3 rows(ufnd_cdiv(B, 4)), blk(256);

  if (!(fo && fo->prep_done)) {
    const ClfPrep prep{fused, aux, (const float*)p->gates, w.xin, w.alpha, ld_fused, d->aux_dim, w.ldx, fused != w.xin ? 1 : 0, B + d->trees * d->depth};
    hipLaunchKernelGGL(clf_prep_kernel, dim3(prep.blocks), blk, 0, stream, prep, B, H);
    UFND_CHECK_LAUNCH();
  }
  {  // pre.0 / pre.3 + GELU(+dropout)                                    deep_truth_classifier.py:121-128
    NtProb a{w.xin, p->pre0_w, p->pre0_b, w.h3, w.z3, B, H, H + d->aux_dim, w.ldx, H + d->aux_dim, H, H, 1, drop, LAYER_PRE0, 1};
    TRY(launch_nt(&a, 1, state, stream));
    NtProb b{w.h3, p->pre3_w, p->pre3_b, w.hh, w.z4, B, H, H, H, H, H, H, 1, drop, LAYER_PRE3, 1};
    TRY(launch_nt(&b, 1, state, stream));
  }
  NI_DISPATCH(H, node_head_kernel, dim3(B), blk, stream, (const float*)w.hh, (const float*)w.alpha, (const float*)p->thresh,
              (const float*)p->leaf, (const float*)p->tau, (const float*)p->bypass_w, (const float*)p->bypass_b,
              (const float*)p->temperature, B, H, d->trees, d->depth, train ? d->node_dropout : 0.0f, state, w.fs, logits,
              probs, fo ? fo->labels : (const int64_t*)nullptr, w.lrow, fo ? fo->d_logits : (float*)nullptr);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
}  // namespace

extern "C" int ufnd_classifier_forward(const ufnd_dims* d, const ufnd_clf_params* p, const float* fused, int ld_fused,
                                       const float* aux, int B, int train, float* workspace, float* logits,
                                       float* probs, const ufnd_step_state* state, void* stream_) {
  return classifier_forward_impl(d, p, fused, ld_fused, aux, B, train, workspace, logits, probs, state, stream_, nullptr);
}

namespace {
struct FusedClfBwd {
  ufnd_step_state* loss_state;   // non-null: block 0 of node_bwd writes the batch-mean loss from the CE rows node_head left
  const float* next_z;           // non-null: the last dX product multiplies by gelu'(next_z) and the dropout mask of the layer that
  float next_drop;               //           produced `fused` (fuse_mlp.3) and writes dZ2 (B, hidden) into d_fused -- act_bwd's job
  bool defer_node_param;         // the parameter-gradient launch is left to the fusion backward (NodeParamJob)
  bool defer_tn;                 // ... and so are the two dW / db problems (FusedBwd::extra_tn)
};
int classifier_backward_impl(const ufnd_dims* d, const ufnd_clf_params* p, const ufnd_clf_params* g, int B,
                             int train, float* workspace, const float* d_logits, float* d_fused,
                             int ld_dfused, const ufnd_step_state* state, void* stream_, void* side_stream_,
                             int join, int flags, const FusedClfBwd* fo) {
  TRY(check_dims(d, B));
  UFND_REQUIRE((flags & ~UFND_BWD_NO_LINEAR_GRADS) == 0, "classifier_backward: flags=%d", flags);
  UFND_REQUIRE(p && g && workspace && d_logits && d_fused && state, "classifier_backward: null argument");
  UFND_REQUIRE(ld_dfused % 4 == 0 && ufnd_aligned(d_fused, 16), "classifier_backward: d_fused alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const ForkJoin fj{stream, (hipStream_t)side_stream_};
  const int H = d->hidden;
  ClfWs w = carve_clf(*d, B, workspace);
  const float drop = train ? d->clf_dropout : 0.0f, ndrop = train ? d->node_dropout : 0.0f;
  const dim3 rows(ufnd_cdiv(B, 4)), blk(256);

  NI_DISPATCH(H, node_bwd_kernel, dim3(B), blk, stream, d_logits, (const float*)w.fs, (const float*)w.alpha,
              (const float*)p->leaf, (const float*)p->tau, (const float*)p->bypass_w, (const float*)w.z4, B, H, d->trees,
              d->depth, ndrop, drop, state, w.df, w.dz4, fo && fo->loss_state ? (const float*)w.lrow : (const float*)nullptr,
              fo ? fo->loss_state : (ufnd_step_state*)nullptr);
  UFND_CHECK_LAUNCH();
  fj.fork(5);   // df, dz4 are ready
  if (!(fo && fo->defer_node_param)) {
    const int S = param_slices(B);
    // (one-pass form: the leaf tables take ceil(2 trees leaves / 8) workgroups of their own; row-sliced form: one per slice)
    const int leaf_blocks = S > 1 ? 1 : ufnd_cdiv(d->trees * (1 << d->depth) * 2, 8);
    hipLaunchKernelGGL(node_param_kernel, dim3(d->trees * d->depth + 2 + leaf_blocks, S), blk, 0, fj.dw(), (const float*)w.df, d_logits,
                       (const float*)w.hh, (const float*)w.alpha, (const float*)w.fs, B, H, d->trees, d->depth, ndrop, state,
                       g->gates, g->thresh, g->leaf, g->bypass_w, g->bypass_b, S > 1 ? w.npart : (float*)nullptr, (B + S - 1) / S, GateRows{});
    UFND_CHECK_LAUNCH();
    if (S > 1) {
      hipLaunchKernelGGL(node_param_finish_kernel, dim3(d->trees * d->depth + 3), blk, 0, fj.dw(), (const float*)w.npart, S,
                         (const float*)w.alpha, H, d->trees, d->depth, g->gates, g->thresh, g->leaf, g->bypass_w, g->bypass_b);
      UFND_CHECK_LAUNCH();
    }
  }
  {  // pre.3: dX -> dz3 (epilogue applies gelu'(z3) * mask)
    NnProb n{w.dz4, p->pre3_w, w.dz3, w.z3, nullptr, B, H, H, H, H, H, H, 0, drop, LAYER_PRE0, H, 1};
    TRY(launch_nn(&n, 1, state, stream));
  }
  {  // both dW products of the module in one grouped launch (pre.0 over the full (hidden + aux) width), then pre.0's dX
     // over the fused columns only
    fj.fork(6);   // dz3 is ready
    TnProb t[2];
    clf_tn(*d, ClfFactors{w.dz4, w.h3, w.dz3, w.xin, w.ldx}, g, B, t);
    if (!(flags & UFND_BWD_NO_LINEAR_GRADS) && !(fo && fo->defer_tn)) TRY(launch_tn(t, 2, fj.dw()));
    NnProb n{w.dz3, p->pre0_w, d_fused, nullptr, nullptr, B, H, H, H, H + d->aux_dim, ld_dfused, 0, 0, 0.0f, 0, 0, 1};
    if (fo && fo->next_z) {      // dZ2 = d_fused gelu'(Z2) mask, the expression of act_bwd_kernel, in this product's epilogue
      n.actZ = fo->next_z;
      n.ldz = H;
      n.drop_p = fo->next_drop;
      n.drop_layer = LAYER_FUSE3;
      n.drop_ld = H;
    }
    TRY(launch_nn(&n, 1, state, stream));
  }
  if (join) fj.join(7);
  return UFND_OK;
}
}  // namespace

extern "C" int ufnd_classifier_backward_ex(const ufnd_dims* d, const ufnd_clf_params* p, const ufnd_clf_params* g, int B,
                                           int train, float* workspace, const float* d_logits, float* d_fused,
                                           int ld_dfused, const ufnd_step_state* state, void* stream_, void* side_stream_,
                                           int join, int flags) {
  return classifier_backward_impl(d, p, g, B, train, workspace, d_logits, d_fused, ld_dfused, state, stream_, side_stream_, join, flags, nullptr);
}

extern "C" int ufnd_classifier_backward(const ufnd_dims* d, const ufnd_clf_params* p, const ufnd_clf_params* g, int B,
                                        int train, float* workspace, const float* d_logits, float* d_fused,
                                        int ld_dfused, const ufnd_step_state* state, void* stream_, void* side_stream_,
                                        int join) {
  return ufnd_classifier_backward_ex(d, p, g, B, train, workspace, d_logits, d_fused, ld_dfused, state, stream_, side_stream_, join, 0);
}

// ------------------------------------------------------------------------------------------------
// Factor form of the head's Linear gradients (data parallel; the reference step is single-process, forensic_trainer.py:285-298).
// Every large gradient of the head is dW = dY^T X over the batch rows, so the SUM over ranks of dW is dY_all^T X_all over all
// ranks' rows: a rank packs its factor panels (2.5 MB at B = 32, hidden 512 -- against the 51 MB gradient), the packs are
// all-gathered, and every rank forms the summed dW / db of all 13 Linears in ONE grouped launch over ranks x B rows.
// Pack layout (floats, every panel on a 64-float boundary, B rows each):
//   [dz2 H | h1 2H | dz1 2H | cat 16H | dqkv 9H | d_text_proj H | d_audio_proj H | d_visual_proj H | d_temporal_proj H | d_gnn_proj H |
//    text | audio | visual | temporal | gnn | dz4 H | h3 H | dz3 H | xin (hidden + 4)]
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int FACTOR_PANELS = 19;
struct FactorLayout {
  size_t off[FACTOR_PANELS], n[FACTOR_PANELS];
  size_t total;
};
FactorLayout factor_layout(const ufnd_dims& d, int B) {
  const size_t H = d.hidden, b = B;
  const size_t width[FACTOR_PANELS] = {H, 2 * H, 2 * H, 16 * H, 9 * H, H, H, H, H, H, (size_t)d.text_dim, (size_t)d.audio_dim, (size_t)d.visual_dim,
                                       (size_t)d.temporal_dim, (size_t)d.gnn_dim, H, H, H, H + 4};
  FactorLayout l;
  size_t o = 0;
  for (int i = 0; i < FACTOR_PANELS; ++i) {
    l.off[i] = o;
    l.n[i] = b * width[i];
    o += al64(l.n[i]);
  }
  l.total = o;
  return l;
}
struct CopyArgs {
  const float* src[FACTOR_PANELS];
  float* dst[FACTOR_PANELS];
  unsigned n4[FACTOR_PANELS];      // 16-byte words
};
__global__ __launch_bounds__(256) void copy_panels_kernel(const CopyArgs a) {
  const int it = blockIdx.y;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n4[it]) st4(a.dst[it] + 4 * (size_t)i, ld4(a.src[it] + 4 * (size_t)i));
}
}  // namespace

extern "C" size_t ufnd_head_factor_floats(const ufnd_dims* d, int B) {
  if (!d || B < 1) return 0;
  return factor_layout(*d, B).total;
}

extern "C" int ufnd_head_pack_factors(const ufnd_dims* d, const float* text, const float* audio, const float* visual, const float* temporal,
                                      const float* gnn, int B, float* fusion_workspace, float* clf_workspace, float* pack, void* stream_) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(text && audio && visual && temporal && (gnn || d->gnn_dim == 0) && fusion_workspace && clf_workspace && pack, "head_pack_factors: null argument");
  UFND_REQUIRE(d->text_dim % 4 == 0 && d->audio_dim % 4 == 0 && d->visual_dim % 4 == 0 && d->temporal_dim % 4 == 0 && d->gnn_dim % 4 == 0,
               "head_pack_factors: input widths must be multiples of 4");
  const FusionWs w = carve_fusion(*d, B, fusion_workspace);
  const ClfWs c = carve_clf(*d, B, clf_workspace);
  const FusionFactors f = fusion_factors(w, B, d->hidden, text, audio, visual, temporal, gnn);
  const FactorLayout l = factor_layout(*d, B);
  const float* src[FACTOR_PANELS] = {f.dz2, f.h1, f.dz1, f.cat, f.dqkv, f.dproj[0], f.dproj[1], f.dproj[2], f.dproj[3], f.dproj[4],
                                     f.x[0], f.x[1], f.x[2], f.x[3], f.x[4], c.dz4, c.h3, c.dz3, c.xin};
  CopyArgs a;
  unsigned most = 0;
  UFND_REQUIRE(ufnd_aligned(pack, 16), "head_pack_factors: pack alignment");
  for (int i = 0; i < FACTOR_PANELS; ++i) {
    const bool used = l.n[i] > 0 && src[i];
    UFND_REQUIRE(!used || ufnd_aligned(src[i], 16), "head_pack_factors: panel %d is not 16-byte aligned", i);
    a.src[i] = src[i];
    a.dst[i] = pack + l.off[i];
    a.n4[i] = used ? (unsigned)(l.n[i] / 4) : 0u;
    most = a.n4[i] > most ? a.n4[i] : most;
  }
  hipLaunchKernelGGL(copy_panels_kernel, dim3(ufnd_cdiv((int)most, 256), FACTOR_PANELS), dim3(256), 0, (hipStream_t)stream_, a);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_head_linear_grads_from_factors(const ufnd_dims* d, const ufnd_fusion_params* fusion_grads, const ufnd_clf_params* clf_grads,
                                                   const float* packs, size_t rank_stride, int ranks, int B, void* stream_) {
  TRY(check_dims(d, B));
  const FactorLayout l = factor_layout(*d, B);
  UFND_REQUIRE(fusion_grads && clf_grads && packs && ranks >= 1, "head_linear_grads_from_factors: null argument");
  UFND_REQUIRE(rank_stride >= l.total && rank_stride % 4 == 0 && rank_stride < (1u << 30) && ufnd_aligned(packs, 16),
               "head_linear_grads_from_factors: rank stride %zu floats (a pack is %zu)", rank_stride, l.total);
  UFND_REQUIRE((long long)ranks * B <= 4096, "head_linear_grads_from_factors: %d ranks x %d rows", ranks, B);
  const float* q = packs;
  const FusionFactors f{q + l.off[0], q + l.off[1], q + l.off[2], q + l.off[3], q + l.off[4],
                        {q + l.off[5], q + l.off[6], q + l.off[7], q + l.off[8], q + l.off[9]},
                        {q + l.off[10], q + l.off[11], q + l.off[12], q + l.off[13], q + l.off[14]}};
  const ClfFactors c{q + l.off[15], q + l.off[16], q + l.off[17], q + l.off[18], d->hidden + 4};
  const Seg sg{B, (int)rank_stride};
  const int M = ranks * B;
  TnProb tn[UFND_GEMM_MAX_PROB];
  int n = fusion_tn_fuse(*d, f, fusion_grads, M, tn, sg);
  n += fusion_tn_rest(*d, f, fusion_grads, M, tn + n, sg);
  // (from 128 rows on -- the batch-split form of the kernel, one vector width per launch -- the classifier's two problems stay a launch
  //  of their own: pre.0 is (hidden + aux) wide, not a multiple of 4, and would put the whole grouped launch on 8-byte accesses)
  const bool aux_vec4 = (d->hidden + d->aux_dim) % 4 == 0 || M < 128;      // (below 128 rows the grouped kernel keeps a vector width per problem)
  if (aux_vec4) n += clf_tn(*d, c, clf_grads, M, tn + n, sg);
  TRY(launch_tn(tn, n, (hipStream_t)stream_));
  if (!aux_vec4) {
    n = clf_tn(*d, c, clf_grads, M, tn, sg);
    TRY(launch_tn(tn, n, (hipStream_t)stream_));
  }
  return UFND_OK;
}

// ------------------------------------------------------------------------------------------------
// The head's train step as TWO entries over both modules (round 4): the same kernels and the same arithmetic as
//   ufnd_fusion_forward -> ufnd_classifier_forward -> ufnd_softmax_ce -> ufnd_classifier_backward -> ufnd_fusion_backward_phase
// (forensic_trainer.py:285-291), with the launches that exist only because those are five calls folded into their neighbours:
//   * the classifier's input preparation (aux columns, alpha = softmax(gates)) rides as extra blocks of the co-attention row kernel;
//   * node_head writes each row's CE term and d_logits, block 0 of node_bwd averages the rows (softmax_ce's order: same bits);
//   * the classifier's last dX product applies gelu'(Z2) and fuse_mlp.3's dropout mask in its epilogue and writes dZ2 (act_bwd);
//   * (whole backward only, B <= 64) the classifier's and the evidence gates' parameter gradients are one launch.
//   * (whole backward only, B < 128) the classifier's two weight-gradient problems join the fusion's grouped launch.
// 26 -> 21 launches per step at B = 32; logits, loss and every gradient bit-identical to the five-call sequence (test).
// ------------------------------------------------------------------------------------------------
extern "C" int ufnd_head_forward_loss(const ufnd_dims* d, const ufnd_fusion_params* fp, const ufnd_clf_params* cp, const ufnd_head_io* io, int B,
                                      int train, ufnd_step_state* state, void* stream_) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(fp && cp && io && state, "head_forward_loss: null argument");
  UFND_REQUIRE(io->fusion_workspace && io->clf_workspace && io->logits && io->probs && io->forensic && io->labels && io->d_logits,
               "head_forward_loss: null buffer");
  UFND_REQUIRE((d->aux_dim == 0) == (io->aux == nullptr), "head_forward_loss: aux must be given iff aux_dim > 0");
  const ClfWs c = carve_clf(*d, B, io->clf_workspace);
  const ClfPrep prep{c.xin, io->aux, (const float*)cp->gates, c.xin, c.alpha, c.ldx, d->aux_dim, c.ldx, 0, B + d->trees * d->depth};
  TRY(fusion_forward_impl(d, fp, io->text, io->audio, io->visual, io->temporal, io->gnn, B, train, io->fusion_workspace, c.xin, c.ldx, nullptr,
                          io->forensic, state, stream_, &prep));
  const FusedClfFwd fo{true, io->labels, io->d_logits};
  return classifier_forward_impl(d, cp, c.xin, c.ldx, io->aux, B, train, io->clf_workspace, io->logits, io->probs, state, stream_, &fo);
}

extern "C" int ufnd_head_backward(const ufnd_dims* d, const ufnd_fusion_params* fp, const ufnd_fusion_params* fg, const ufnd_clf_params* cp,
                                  const ufnd_clf_params* cg, const ufnd_head_io* io, int B, int train, ufnd_step_state* state, void* stream_,
                                  void* side_stream_, int join, int phase) {
  TRY(check_dims(d, B));
  UFND_REQUIRE(fp && fg && cp && cg && io && state && io->fusion_workspace && io->clf_workspace && io->d_logits, "head_backward: null argument");
  const int flags = phase & UFND_BWD_NO_LINEAR_GRADS, ph = phase & ~UFND_BWD_NO_LINEAR_GRADS;
  UFND_REQUIRE(ph == UFND_BWD_ALL || ph == UFND_BWD_FUSE_MLP || ph == UFND_BWD_REST, "head_backward: phase=%d", phase);
  const FusionWs w = carve_fusion(*d, B, io->fusion_workspace);
  const ClfWs c = carve_clf(*d, B, io->clf_workspace);
  const float fdrop = train ? d->fusion_dropout : 0.0f;
  // the parameter-gradient launches merge only when the whole backward is one call (a phased caller starts reducing the classifier's
  // gradients after the first phase: they must be complete there)
  const bool merge_params = ph == UFND_BWD_ALL && B <= 64;
  const NodeParamJob np{c.df, io->d_logits, c.hh, c.alpha, c.fs, d->trees, d->depth, train ? d->node_dropout : 0.0f,
                        cg->gates, cg->thresh, cg->leaf, cg->bypass_w, cg->bypass_b};
  // ... and so do the weight-gradient launches (tn_kernel keeps a vector width per problem: the (hidden + aux)-wide pre.0 does not slow the others)
  const bool merge_tn = ph == UFND_BWD_ALL && B < 128 && !flags;
  TnProb ctn[2];
  clf_tn(*d, ClfFactors{c.dz4, c.h3, c.dz3, c.xin, c.ldx}, cg, B, ctn);
  if (ph != UFND_BWD_REST) {
    const FusedClfBwd fo{state, w.z2, fdrop, merge_params, merge_tn};
    TRY(classifier_backward_impl(d, cp, cg, B, train, io->clf_workspace, io->d_logits, w.dz2, d->hidden, state, stream_, side_stream_, 0, flags, &fo));
  }
  const FusedBwd fb{true, merge_params ? &np : (const NodeParamJob*)nullptr, merge_tn ? ctn : (const TnProb*)nullptr, merge_tn ? 2 : 0};
  return fusion_backward_impl(d, fp, fg, io->text, io->audio, io->visual, io->temporal, io->gnn, B, train, io->fusion_workspace, nullptr, d->hidden,
                              nullptr, state, stream_, side_stream_, join, phase, &fb);
}

extern "C" int ufnd_softmax_ce_weighted(const float* logits, const int64_t* labels, int B, float w0, float w1, float label_smoothing,
                                        float* loss_rows, float* d_logits, ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(logits && labels && state && B >= 1, "softmax_ce_weighted: null argument");
  UFND_REQUIRE(w0 > 0.0f && w1 > 0.0f && label_smoothing >= 0.0f && label_smoothing < 1.0f, "softmax_ce_weighted: w=(%g,%g) eps=%g", w0, w1,
               label_smoothing);
  hipLaunchKernelGGL(softmax_ce_ws_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, logits, labels, B, w0, w1, label_smoothing,
                     loss_rows, d_logits, state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_softmax_ce(const float* logits, const int64_t* labels, int B, float* loss_rows, float* d_logits,
                               ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(logits && labels && state && B >= 1, "softmax_ce: null argument");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, logits, labels, B, loss_rows, d_logits,
                     state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}


// ------------------------------------------------------------------------------------------------
// Batch assembly: CachedTensorDataset.__getitem__ + default collate (forensic_trainer.py:60-83,232-234) and the
// gnn_Z[global_idx] gather of _forward_batch (:240-252) as ONE launch: row idx[r] of every cached tensor goes
// straight into the step's static input buffer (the torch route is one gather + one copy per tensor: 13 launches).
// ------------------------------------------------------------------------------------------------
namespace {
struct GatherArgs {
  const char* src[UFND_GATHER_MAX_ITEMS];
  char* dst[UFND_GATHER_MAX_ITEMS];
  int words[UFND_GATHER_MAX_ITEMS];      // row size in 8-byte words
  long long rows[UFND_GATHER_MAX_ITEMS]; // rows in the source (indices are clamped into it)
};

__global__ __launch_bounds__(128) void gather_rows_kernel(const int64_t* __restrict__ idx, GatherArgs a) {
  const int r = blockIdx.x, it = blockIdx.y;
  long long i = idx[r];
  i = i < 0 ? 0 : (i < a.rows[it] ? i : a.rows[it] - 1);
  const uint64_t* s = reinterpret_cast<const uint64_t*>(a.src[it]) + (size_t)i * a.words[it];
  uint64_t* d = reinterpret_cast<uint64_t*>(a.dst[it]) + (size_t)r * a.words[it];
  for (int w = threadIdx.x; w < a.words[it]; w += 128) d[w] = s[w];
}
}  // namespace

extern "C" int ufnd_gather_rows(const int64_t* idx, int B, const ufnd_gather_item* items, int n_items, void* stream_) {
  UFND_REQUIRE(idx && items && B >= 1 && n_items >= 1 && n_items <= UFND_GATHER_MAX_ITEMS, "gather_rows: B=%d items=%d", B, n_items);
  GatherArgs a;
  for (int i = 0; i < n_items; ++i) {
    const ufnd_gather_item& t = items[i];
    UFND_REQUIRE(t.src && t.dst && t.row_bytes >= 8 && t.row_bytes % 8 == 0 && t.src_rows >= 1, "gather_rows: item %d (row_bytes=%d rows=%lld)",
                 i, t.row_bytes, (long long)t.src_rows);
    UFND_REQUIRE(ufnd_aligned(t.src, 8) && ufnd_aligned(t.dst, 8), "gather_rows: item %d must be 8-B aligned", i);
    a.src[i] = static_cast<const char*>(t.src);
    a.dst[i] = static_cast<char*>(t.dst);
    a.words[i] = t.row_bytes / 8;
    a.rows[i] = t.src_rows;
  }
  hipLaunchKernelGGL(gather_rows_kernel, dim3(B, n_items), dim3(128), 0, (hipStream_t)stream_, idx, a);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
