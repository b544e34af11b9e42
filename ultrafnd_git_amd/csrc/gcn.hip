// Graph side of the trainer's construction, src/training/forensic_trainer.py:
//   build_adj_from_ocr :114-132   O(N^2) Python Jaccard loop   -> ufnd_ocr_adjacency (one block per node)
//   SimpleGCN.forward  :25-53     dense two-layer GCN           -> ufnd_gcn_forward
//   _pretrain_gnn      :214-224   full-graph Adam steps on a degree-regression target -> ufnd_gcn_pretrain_step
// The dense products run on the exact-fp32 MFMA GEMM family of the fusion head (gemm_f32.hip):
//   A_norm @ X is the NN form (rows of A_norm times the node-feature matrix), Linear layers the NT form,
//   weight gradients the TN form; GELU / dropout / their backward are the same fused epilogues.
// Integer work (set intersections) is bit-exact; the threshold test is evaluated in double exactly as the
// reference's Python float arithmetic does: inter / (union + 1e-9) >= thresh.
#include "gemm_f32.hpp"

namespace {

constexpr uint32_t LAYER_GCN = 9;     // dropout stream id (the fusion head uses 1..5)
constexpr int SET_LDS = 2048;         // phrase ids of the block's own set kept in LDS (longer sets are read from memory)

// adj[i][j] = 1 if i == j or Jaccard(set_i, set_j) >= thresh.  Sets are sorted, duplicate-free int32 lists (CSR).
__global__ __launch_bounds__(256) void ocr_adjacency_kernel(const int32_t* __restrict__ offs, const int32_t* __restrict__ toks,
                                                            int N, double thresh, float* __restrict__ adj, int ld) {
  __shared__ int32_t mine[SET_LDS];
  const int i = blockIdx.x;
  const int a0 = offs[i], na = offs[i + 1] - a0;
  const bool in_lds = na <= SET_LDS;
  if (in_lds)
    for (int t = threadIdx.x; t < na; t += 256) mine[t] = toks[a0 + t];
  __syncthreads();
  const int32_t* A = in_lds ? mine : toks + a0;
  for (int j = threadIdx.x; j < N; j += 256) {
    float out = 0.0f;
    if (j == i) {
      out = 1.0f;
    } else {
      const int b0 = offs[j], nb = offs[j + 1] - b0;
      if (na > 0 || nb > 0) {            // jaccard(): both empty -> 0.0
        int p = 0, q = 0, inter = 0;
        while (p < na && q < nb) {       // sorted-merge intersection
          const int32_t x = A[p], y = toks[b0 + q];
          inter += (x == y);
          p += (x <= y);
          q += (y <= x);
        }
        const double jac = (double)inter / ((double)(na + nb - inter) + 1e-9);
        out = jac >= thresh ? 1.0f : 0.0f;
      } else if (0.0 >= thresh) {
        out = 1.0f;
      }
    }
    adj[(size_t)i * ld + j] = out;
  }
}

// dinv[i] = (sum_j adj[i][j] + 1 + 1e-9)^-1/2   (A_hat = A + I; forensic_trainer.py:44-48)
__global__ __launch_bounds__(256) void gcn_degree_kernel(const float* __restrict__ adj, int ld, int N, float* __restrict__ dinv,
                                                         float* __restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.0f;
  for (int c = lane; c < N; c += 64) s += adj[(size_t)row * ld + c];
  s = wave_sum(s);
  if (lane == 0) {
    dinv[row] = powf(s + 1.0f + 1e-9f, -0.5f);
    if (rowsum) rowsum[row] = s;
  }
}

// an[i][j] = dinv[i] (adj[i][j] + [i == j]) dinv[j], zero in the pad columns j >= N (row stride Np)
__global__ __launch_bounds__(256) void gcn_norm_adj_kernel(const float* __restrict__ adj, int ld, const float* __restrict__ dinv,
                                                           int N, int Np, float* __restrict__ an) {
  const int row = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Np) return;
  float v = 0.0f;
  if (c < N) v = dinv[row] * (adj[(size_t)row * ld + c] + (c == row ? 1.0f : 0.0f)) * dinv[c];
  an[(size_t)row * Np + c] = v;
}

// dst rows [0, N) = src rows, rows [N, Np) = 0  (operand of the NN product: its row count is the padded contraction)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int lds_, int N, int Np, int F,
                                                       float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)Np * F) return;
  const int r = (int)(i / F), c = (int)(i % F);
  dst[i] = r < N ? src[(size_t)r * lds_ + c] : 0.0f;
}

// degree-regression head of _pretrain_gnn: pred = sigmoid(z . wh + bh); loss_i = (pred - t)^2 / N with
// t = rowsum(adj)_i / max(1, N); dZ[i][:] = 2 (pred - t) / N * pred (1 - pred) * wh
__global__ __launch_bounds__(256) void gcn_head_kernel(const float* __restrict__ Z, const float* __restrict__ wh, const float* __restrict__ bh,
                                                       const float* __restrict__ rowsum, int N, int D, float* __restrict__ dZ,
                                                       float* __restrict__ loss_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.0f;
  for (int c = lane; c < D; c += 64) s += Z[(size_t)row * D + c] * wh[c];
  s = wave_sum(s) + bh[0];
  const float pred = 1.0f / (1.0f + __expf(-s));
  const float t = rowsum[row] / fmaxf(1.0f, (float)N);
  const float d = pred - t;
  if (lane == 0) loss_rows[row] = d * d / (float)N;
  const float ds = 2.0f * d / (float)N * pred * (1.0f - pred);
  for (int c = lane; c < D; c += 64) dZ[(size_t)row * D + c] = ds * wh[c];
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];     // fixed order: deterministic
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// torch.optim.Adam (L2 weight decay folded into the gradient, NOT AdamW), step t (1-based)
__global__ __launch_bounds__(256) void gcn_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, float lr, float wd, float b1, float b2,
                                                       float eps, float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] + wd * p[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

// ---- the integrated trainer variant's per-mini-batch graph (src/training/forensic_trainer_integrated.py:77-98,203-224;
//      src/models/gnn/gnn_model.py): weighted Jaccard adjacency, GNNModel = lin1 -> A_norm -> ReLU -> dropout -> A_norm -> lin2
constexpr uint32_t LAYER_GNN = 10;    // dropout stream id of GNNModel

// adj[i][j] = Jaccard(set_i, set_j) if >= thresh else 0, for i != j with both sets non-empty; zero diagonal
// (build_adj_from_ocr_sets, forensic_trainer_integrated.py:77-98: s = inter / union in Python floats, stored as float32)
__global__ __launch_bounds__(256) void ocr_adjacency_weighted_kernel(const int32_t* __restrict__ offs, const int32_t* __restrict__ toks,
                                                                     int N, double thresh, float* __restrict__ adj, int ld) {
  __shared__ int32_t mine[SET_LDS];
  const int i = blockIdx.x;
  const int a0 = offs[i], na = offs[i + 1] - a0;
  const bool in_lds = na <= SET_LDS;
  if (in_lds)
    for (int t = threadIdx.x; t < na; t += 256) mine[t] = toks[a0 + t];
  __syncthreads();
  const int32_t* A = in_lds ? mine : toks + a0;
  for (int j = threadIdx.x; j < N; j += 256) {
    float out = 0.0f;
    const int b0 = offs[j], nb = offs[j + 1] - b0;
    if (j != i && na > 0 && nb > 0) {
      int p = 0, q = 0, inter = 0;
      while (p < na && q < nb) {
        const int32_t x = A[p], y = toks[b0 + q];
        inter += (x == y);
        p += (x <= y);
        q += (y <= x);
      }
      const double jac = (double)inter / (double)(na + nb - inter);       // union > 0 here
      if (jac >= thresh) out = (float)jac;
    }
    adj[(size_t)i * ld + j] = out;
  }
}

// H = drop(relu(U)) over N x F (H may have more rows than N: they are zeroed by the caller)
__global__ __launch_bounds__(256) void relu_drop_kernel(const float* __restrict__ U, float* __restrict__ H, size_t n, float drop_p,
                                                        const ufnd_step_state* st) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  H[i] = fmaxf(U[i], 0.0f) * dropout_mul(st, drop_p, LAYER_GNN, (uint32_t)i);
}
// dU = dH * [U > 0] * dropout mask
__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float* __restrict__ dH, const float* __restrict__ U, float* __restrict__ dU,
                                                            size_t n, float drop_p, const ufnd_step_state* st) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  dU[i] = (U[i] > 0.0f ? dH[i] : 0.0f) * dropout_mul(st, drop_p, LAYER_GNN, (uint32_t)i);
}

// node features of a batch: [T[:, :nt], A[:, :na], V[:, :nv], U[:, :nu]] / (||.|| + 1e-9), one wave per row
// (forensic_trainer.py:193-195; the integrated variant's 416-wide node feature)
__global__ __launch_bounds__(256) void node_features_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ V, int ldv, const float* __restrict__ U, int ldu,
                                                            int nt, int na, int nv, int nu, int B, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int F = nt + na + nv + nu;
  float ss = 0.0f;
  for (int c = lane; c < F; c += 64) {
    const float v = c < nt ? T[(size_t)row * ldt + c] : c < nt + na ? A[(size_t)row * lda + c - nt]
                    : c < nt + na + nv ? V[(size_t)row * ldv + c - nt - na] : U[(size_t)row * ldu + c - nt - na - nv];
    out[(size_t)row * F + c] = v;
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / (sqrtf(ss) + 1e-9f);
  for (int c = lane; c < F; c += 64) out[(size_t)row * F + c] *= inv;
}

struct GnnWs {
  float *an, *dinv, *Y1, *U, *H, *G, *dG, *dH, *dU, *dY1;
  int Np;
};
size_t gnn_layout(float* base, int N, int F, int hid, int out, GnnWs* w);

struct GcnWs {
  float *an, *xp, *P, *U1, *H, *Q, *dinv, *rowsum, *dZ, *dQ, *dU1, *grad, *loss_rows;
  int Np;
};

size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

size_t gcn_layout(float* base, int N, int F, int hid, int out, int train, GcnWs* w) {
  const int Np = (N + 31) & ~31;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += align64(n); return p; };
  GcnWs t{};
  t.Np = Np;
  t.an = take((size_t)N * Np);
  t.xp = take((size_t)Np * F);
  t.P = take((size_t)N * F);
  t.U1 = take((size_t)N * hid);
  t.H = take((size_t)Np * hid);
  t.Q = take((size_t)N * hid);
  t.dinv = take(N);
  t.rowsum = take(N);
  if (train) {
    t.dZ = take((size_t)N * out);
    t.dQ = take((size_t)Np * hid);
    t.dU1 = take((size_t)N * hid);
    t.grad = take((size_t)hid * F + hid + (size_t)out * hid + out);
    t.loss_rows = take(N);
  }
  if (w) *w = t;
  return o;
}

size_t gnn_layout(float* base, int N, int F, int hid, int out, GnnWs* w) {
  const int Np = (N + 31) & ~31;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += align64(n); return p; };
  GnnWs t{};
  t.Np = Np;
  t.an = take((size_t)N * Np);
  t.dinv = take(N);
  t.Y1 = take((size_t)Np * hid);      // operands of the N x Np aggregation products carry Np rows (pad rows zero)
  t.U = take((size_t)N * hid);
  t.H = take((size_t)Np * hid);
  t.G = take((size_t)N * hid);
  t.dG = take((size_t)Np * hid);
  t.dH = take((size_t)N * hid);
  t.dU = take((size_t)Np * hid);
  t.dY1 = take((size_t)N * hid);
  (void)F; (void)out;
  if (w) *w = t;
  return o;
}

int gcn_check(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, int N, int F, int hid, int out) {
  UFND_REQUIRE(x && adj && p && p->w1 && p->b1 && p->w2 && p->b2, "gcn: null argument");
  UFND_REQUIRE(N >= 1 && ld_adj >= N && F >= 4 && F % 4 == 0 && hid % 32 == 0 && out % 32 == 0,
               "gcn: N=%d in_dim=%d hid=%d out=%d (in_dim %% 4, hid / out %% 32)", N, F, hid, out);
  UFND_REQUIRE(ufnd_aligned(x, 16) && ufnd_aligned(p->w1, 16) && ufnd_aligned(p->w2, 16) && ufnd_aligned(p->b1, 16) && ufnd_aligned(p->b2, 16),
               "gcn: 16-B alignment required");
  return UFND_OK;
}

// forward into the workspace; drop_p > 0 applies train-mode dropout to gelu(lin1(.)) (keyed by st)
int gcn_forward_ws(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* z, const GcnWs& w, int N, int F,
                   int hid, int out, float drop_p, const ufnd_step_state* st, hipStream_t stream) {
  const int Np = w.Np;
  hipLaunchKernelGGL(gcn_degree_kernel, dim3(ufnd_cdiv(N, 4)), dim3(256), 0, stream, adj, ld_adj, N, w.dinv, w.rowsum);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(gcn_norm_adj_kernel, dim3(ufnd_cdiv(Np, 256), N), dim3(256), 0, stream, adj, ld_adj, w.dinv, N, Np, w.an);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((size_t)Np * F + 255) / 256)), dim3(256), 0, stream, x, F, N, Np, F, w.xp);
  UFND_CHECK_LAUNCH();
  NnProb ax{w.an, w.xp, w.P, nullptr, nullptr, N, Np, F, Np, F, F, 0, 0, 0.0f, 0, 0, 1};                 // P = A_norm X
  int rc = launch_nn(&ax, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  NtProb l1{w.P, p->w1, p->b1, w.H, w.U1, N, hid, F, F, F, hid, hid, 1, drop_p, LAYER_GCN, 1};            // H = drop(gelu(P W1^T + b1))
  rc = launch_nt(&l1, 1, st, stream);
  if (rc != UFND_OK) return rc;
  if (Np > N) {
    hipError_t e = hipMemsetAsync(w.H + (size_t)N * hid, 0, (size_t)(Np - N) * hid * sizeof(float), stream);
    if (e != hipSuccess) { ufnd_set_error("gcn: memset failed: %s", hipGetErrorString(e)); return UFND_ERR_LAUNCH; }
  }
  NnProb ah{w.an, w.H, w.Q, nullptr, nullptr, N, Np, hid, Np, hid, hid, 0, 0, 0.0f, 0, 0, 1};            // Q = A_norm H
  rc = launch_nn(&ah, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  NtProb l2{w.Q, p->w2, p->b2, z, nullptr, N, out, hid, hid, hid, out, 0, 0, 0.0f, 0, 1};                 // Z = Q W2^T + b2
  return launch_nt(&l2, 1, nullptr, stream);
}

}  // namespace

extern "C" int ufnd_ocr_adjacency(const int32_t* offsets, const int32_t* tokens, int N, double thresh, float* adj, int ld,
                                  void* stream_) {
  UFND_REQUIRE(offsets && adj && N >= 1 && ld >= N, "ocr_adjacency: N=%d ld=%d", N, ld);
  hipLaunchKernelGGL(ocr_adjacency_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, offsets, tokens, N, thresh, adj, ld);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" size_t ufnd_gcn_workspace_floats(int N, int in_dim, int hid, int out_dim, int train) {
  if (N < 1 || in_dim < 1 || hid < 1 || out_dim < 1) return 0;
  return gcn_layout(nullptr, N, in_dim, hid, out_dim, train, nullptr);
}

extern "C" int ufnd_gcn_forward(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* z, float* workspace,
                                int N, int in_dim, int hid, int out_dim, float dropout_p, const ufnd_step_state* state,
                                void* stream_) {
  int rc = gcn_check(x, adj, ld_adj, p, N, in_dim, hid, out_dim);
  if (rc != UFND_OK) return rc;
  UFND_REQUIRE(z && workspace && ufnd_aligned(z, 16) && ufnd_aligned(workspace, 16), "gcn_forward: z / workspace");
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (dropout_p == 0.0f || state), "gcn_forward: dropout needs a step state");
  GcnWs w;
  gcn_layout(workspace, N, in_dim, hid, out_dim, 0, &w);
  return gcn_forward_ws(x, adj, ld_adj, p, z, w, N, in_dim, hid, out_dim, dropout_p, state, (hipStream_t)stream_);
}

extern "C" int ufnd_gcn_pretrain_step(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* exp_avg,
                                      float* exp_avg_sq, const float* head_w, const float* head_b, float* z, float* workspace,
                                      int N, int in_dim, int hid, int out_dim, float dropout_p, float lr, float weight_decay,
                                      int step, const ufnd_step_state* state, float* loss, void* stream_) {
  int rc = gcn_check(x, adj, ld_adj, p, N, in_dim, hid, out_dim);
  if (rc != UFND_OK) return rc;
  UFND_REQUIRE(exp_avg && exp_avg_sq && head_w && head_b && z && workspace && loss && step >= 1, "gcn_pretrain_step: null argument");
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (dropout_p == 0.0f || state), "gcn_pretrain_step: dropout needs a step state");
  const size_t n1 = (size_t)hid * in_dim, n2 = (size_t)out_dim * hid, total = n1 + hid + n2 + out_dim;
  UFND_REQUIRE(p->b1 == p->w1 + n1 && p->w2 == p->b1 + hid && p->b2 == p->w2 + n2,
               "gcn_pretrain_step: the parameters must be one flat buffer [w1 | b1 | w2 | b2]");
  hipStream_t stream = (hipStream_t)stream_;
  GcnWs w;
  gcn_layout(workspace, N, in_dim, hid, out_dim, 1, &w);
  rc = gcn_forward_ws(x, adj, ld_adj, p, z, w, N, in_dim, hid, out_dim, dropout_p, state, stream);
  if (rc != UFND_OK) return rc;
  hipLaunchKernelGGL(gcn_head_kernel, dim3(ufnd_cdiv(N, 4)), dim3(256), 0, stream, z, head_w, head_b, w.rowsum, N, out_dim, w.dZ,
                     w.loss_rows);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, stream, w.loss_rows, N, loss);
  UFND_CHECK_LAUNCH();
  float* gw1 = w.grad;
  float* gb1 = gw1 + n1;
  float* gw2 = gb1 + hid;
  float* gb2 = gw2 + n2;
  TnProb t2{w.dZ, w.Q, gw2, gb2, N, out_dim, hid, out_dim, hid, hid};                                      // dW2 = dZ^T Q, db2
  rc = launch_tn(&t2, 1, stream);
  if (rc != UFND_OK) return rc;
  NnProb dq{w.dZ, p->w2, w.dQ, nullptr, nullptr, N, out_dim, hid, out_dim, hid, hid, 0, 0, 0.0f, 0, 0, 1};    // dQ = dZ W2
  rc = launch_nn(&dq, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  if (w.Np > N) {
    hipError_t e = hipMemsetAsync(w.dQ + (size_t)N * hid, 0, (size_t)(w.Np - N) * hid * sizeof(float), stream);
    if (e != hipSuccess) { ufnd_set_error("gcn: memset failed: %s", hipGetErrorString(e)); return UFND_ERR_LAUNCH; }
  }
  // dU1 = (A_norm dQ) * gelu'(U1) * dropout mask     (A_norm is symmetric)
  NnProb du{w.an, w.dQ, w.dU1, w.U1, nullptr, N, w.Np, hid, w.Np, hid, hid, hid, 0, dropout_p, LAYER_GCN, hid, 1};
  rc = launch_nn(&du, 1, state, stream);
  if (rc != UFND_OK) return rc;
  TnProb t1{w.dU1, w.P, gw1, gb1, N, hid, in_dim, hid, in_dim, in_dim};                                    // dW1 = dU1^T P, db1
  rc = launch_tn(&t1, 1, stream);
  if (rc != UFND_OK) return rc;
  const float b1 = 0.9f, b2 = 0.999f;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
  hipLaunchKernelGGL(gcn_adam_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, const_cast<float*>(p->w1), w.grad,
                     exp_avg, exp_avg_sq, total, lr, weight_decay, b1, b2, 1e-8f, bc1, bc2s);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// =============================================================================================
// The integrated trainer variant's in-graph GNN
// =============================================================================================
extern "C" int ufnd_ocr_adjacency_weighted(const int32_t* offsets, const int32_t* tokens, int N, double thresh, float* adj, int ld,
                                           void* stream_) {
  UFND_REQUIRE(offsets && adj && N >= 1 && ld >= N, "ocr_adjacency_weighted: N=%d ld=%d", N, ld);
  hipLaunchKernelGGL(ocr_adjacency_weighted_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, offsets, tokens, N, thresh, adj, ld);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_node_features(const float* text, int ld_text, const float* audio, int ld_audio, const float* visual, int ld_visual,
                                  const float* temporal, int ld_temporal, int n_text, int n_audio, int n_visual, int n_temporal, int B,
                                  float* out, void* stream_) {
  UFND_REQUIRE(text && audio && visual && temporal && out && B >= 1, "node_features: null argument");
  UFND_REQUIRE(n_text >= 1 && n_audio >= 1 && n_visual >= 1 && n_temporal >= 1 && n_text <= ld_text && n_audio <= ld_audio &&
                   n_visual <= ld_visual && n_temporal <= ld_temporal, "node_features: slice widths");
  hipLaunchKernelGGL(node_features_kernel, dim3(ufnd_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream_, text, ld_text, audio, ld_audio, visual,
                     ld_visual, temporal, ld_temporal, n_text, n_audio, n_visual, n_temporal, B, out);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" size_t ufnd_gnn_workspace_floats(int N, int in_dim, int hid, int out_dim) {
  if (N < 1 || in_dim < 1 || hid < 1 || out_dim < 1) return 0;
  return gnn_layout(nullptr, N, in_dim, hid, out_dim, nullptr);
}

// GNNModel.forward (src/models/gnn/gnn_model.py:31-41): Z = lin2(A_norm @ drop(relu(A_norm @ lin1(X)))),
// A_norm = D^-1/2 (A + I) D^-1/2 with D = rowsum(A + I) clamped at 1e-9 (weighted or 0/1 A; zero diagonal expected).
extern "C" int ufnd_gnn_forward(const float* x, const float* adj, int ld_adj, const ufnd_gcn_params* p, float* z, float* workspace,
                                int N, int in_dim, int hid, int out_dim, float dropout_p, const ufnd_step_state* state, void* stream_) {
  int rc = gcn_check(x, adj, ld_adj, p, N, in_dim, hid, out_dim);
  if (rc != UFND_OK) return rc;
  UFND_REQUIRE(z && workspace && ufnd_aligned(z, 16) && ufnd_aligned(workspace, 16), "gnn_forward: z / workspace");
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (dropout_p == 0.0f || state), "gnn_forward: dropout needs a step state");
  hipStream_t stream = (hipStream_t)stream_;
  GnnWs w;
  gnn_layout(workspace, N, in_dim, hid, out_dim, &w);
  const int Np = w.Np;
  hipLaunchKernelGGL(gcn_degree_kernel, dim3(ufnd_cdiv(N, 4)), dim3(256), 0, stream, adj, ld_adj, N, w.dinv, (float*)nullptr);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(gcn_norm_adj_kernel, dim3(ufnd_cdiv(Np, 256), N), dim3(256), 0, stream, adj, ld_adj, w.dinv, N, Np, w.an);
  UFND_CHECK_LAUNCH();
  auto zero_pad = [&](float* buf) -> int {
    if (Np > N) {
      hipError_t e = hipMemsetAsync(buf + (size_t)N * hid, 0, (size_t)(Np - N) * hid * sizeof(float), stream);
      if (e != hipSuccess) { ufnd_set_error("gnn: memset failed: %s", hipGetErrorString(e)); return UFND_ERR_LAUNCH; }
    }
    return UFND_OK;
  };
  NtProb l1{x, p->w1, p->b1, w.Y1, nullptr, N, hid, in_dim, in_dim, in_dim, hid, 0, 0, 0.0f, 0, 1};            // Y1 = X W1^T + b1
  rc = launch_nt(&l1, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  if ((rc = zero_pad(w.Y1)) != UFND_OK) return rc;
  NnProb a1{w.an, w.Y1, w.U, nullptr, nullptr, N, Np, hid, Np, hid, hid, 0, 0, 0.0f, 0, 0, 1};                  // U = A_norm Y1
  rc = launch_nn(&a1, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  const size_t n = (size_t)N * hid;
  hipLaunchKernelGGL(relu_drop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)w.U, w.H, n, dropout_p, state);
  UFND_CHECK_LAUNCH();
  if ((rc = zero_pad(w.H)) != UFND_OK) return rc;
  NnProb a2{w.an, w.H, w.G, nullptr, nullptr, N, Np, hid, Np, hid, hid, 0, 0, 0.0f, 0, 0, 1};                   // G = A_norm H
  rc = launch_nn(&a2, 1, nullptr, stream);
  if (rc != UFND_OK) return rc;
  NtProb l2{w.G, p->w2, p->b2, z, nullptr, N, out_dim, hid, hid, hid, out_dim, 0, 0, 0.0f, 0, 1};               // Z = G W2^T + b2
  return launch_nt(&l2, 1, nullptr, stream);
}

// autograd backward of the above for a gradient d_z (N, out_dim) arriving at Z: writes the four parameter gradients in g
// (overwritten).  The node features and the adjacency are data (no gradient).  `workspace` is the one forward filled.
extern "C" int ufnd_gnn_backward(const float* x, const ufnd_gcn_params* p, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                                 const float* d_z, float* workspace, int N, int in_dim, int hid, int out_dim, float dropout_p,
                                 const ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(x && p && p->w1 && p->w2 && g_w1 && g_b1 && g_w2 && g_b2 && d_z && workspace, "gnn_backward: null argument");
  UFND_REQUIRE(N >= 1 && in_dim % 4 == 0 && hid % 32 == 0 && out_dim % 32 == 0, "gnn_backward: N=%d in_dim=%d hid=%d out=%d", N, in_dim, hid, out_dim);
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (dropout_p == 0.0f || state), "gnn_backward: dropout needs a step state");
  hipStream_t stream = (hipStream_t)stream_;
  GnnWs w;
  gnn_layout(workspace, N, in_dim, hid, out_dim, &w);
  const int Np = w.Np;
  auto zero_pad = [&](float* buf) -> int {
    if (Np > N) {
      hipError_t e = hipMemsetAsync(buf + (size_t)N * hid, 0, (size_t)(Np - N) * hid * sizeof(float), stream);
      if (e != hipSuccess) { ufnd_set_error("gnn: memset failed: %s", hipGetErrorString(e)); return UFND_ERR_LAUNCH; }
    }
    return UFND_OK;
  };
  int rc;
  TnProb t2{d_z, w.G, g_w2, g_b2, N, out_dim, hid, out_dim, hid, hid};                                              // dW2 = dZ^T G, db2
  if ((rc = launch_tn(&t2, 1, stream)) != UFND_OK) return rc;
  NnProb dg{d_z, p->w2, w.dG, nullptr, nullptr, N, out_dim, hid, out_dim, hid, hid, 0, 0, 0.0f, 0, 0, 1};           // dG = dZ W2
  if ((rc = launch_nn(&dg, 1, nullptr, stream)) != UFND_OK) return rc;
  if ((rc = zero_pad(w.dG)) != UFND_OK) return rc;
  NnProb dh{w.an, w.dG, w.dH, nullptr, nullptr, N, Np, hid, Np, hid, hid, 0, 0, 0.0f, 0, 0, 1};                     // dH = A_norm dG (symmetric)
  if ((rc = launch_nn(&dh, 1, nullptr, stream)) != UFND_OK) return rc;
  const size_t n = (size_t)N * hid;
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)w.dH, (const float*)w.U,
                     w.dU, n, dropout_p, state);
  UFND_CHECK_LAUNCH();
  if ((rc = zero_pad(w.dU)) != UFND_OK) return rc;
  NnProb dy{w.an, w.dU, w.dY1, nullptr, nullptr, N, Np, hid, Np, hid, hid, 0, 0, 0.0f, 0, 0, 1};                    // dY1 = A_norm dU
  if ((rc = launch_nn(&dy, 1, nullptr, stream)) != UFND_OK) return rc;
  TnProb t1{w.dY1, x, g_w1, g_b1, N, hid, in_dim, hid, in_dim, in_dim};                                             // dW1 = dY1^T X, db1
  return launch_tn(&t1, 1, stream);
}
