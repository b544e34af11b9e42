// TemporalSyncNet.forward -- the sequence path (src/core_blocks/temporal_blocks.py:16-43 `_TinyTCN`, :141-157):
//   x = [text_seq, vis_seq]  (B, T, C)  channel-last -- the reference transposes to (B, C, T) for nn.Conv1d; here the
//                                       frames stay rows, so every layer is a row-major GEMM over M = B*T rows
//   per layer i (dilation 2^i, padding 'same'):
//     y = conv1d(h)            -> unfold the k taps of each frame into one row (k*C wide, zero outside the clip),
//                                 then the fp32 MFMA skinny GEMM (gemm_f32.hip) against the tap-major packed weight
//     z = dropout(gelu(batchnorm(y)));  h = h + z when the widths match, else z
//   out = head([mean_t h, max_t h])
// BatchNorm1d: eval = running statistics; train = statistics of this batch over (B, T) per channel (biased variance
// for the normalisation, unbiased for the running update, momentum as given).  Forward only: nothing in the
// reference ever trains this module.  fp32 throughout.
#include "gemm_f32.hpp"

namespace {

constexpr uint32_t LAYER_TCN = 16;   // dropout stream ids 16.. (the fusion head uses 1..5, the GCN 9)

static inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// cols[m][j*C + c] = src(b, t - left + j*dil, c)  (0 outside [0, T));  pad columns [k*C, ldc) are zero.
// src is one (M, C0) matrix, or two side by side: columns [0, C0) from s0, [C0, C0+C1) from s1.
__global__ __launch_bounds__(256) void tcn_unfold_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1,
                                                         int T, int k, int dil, int left, float* __restrict__ cols, int ldc) {
  const int m = blockIdx.x;
  const int b = m / T, t = m - b * T;
  const int C = C0 + C1;
  float* row = cols + (size_t)m * ldc;
  for (int x = threadIdx.x; x < ldc; x += 256) {
    float v = 0.0f;
    if (x < k * C) {
      const int j = x / C, c = x - j * C;
      const int tt = t - left + j * dil;
      if (tt >= 0 && tt < T) {
        const size_t r = (size_t)b * T + tt;
        v = c < C0 ? s0[r * C0 + c] : s1[r * C1 + (c - C0)];
      }
    }
    row[x] = v;
  }
}

// first layer's residual source when the input width equals the hidden width: h0 = [s0, s1] as one matrix
__global__ __launch_bounds__(256) void tcn_concat_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1,
                                                         size_t M, float* __restrict__ out) {
  const int C = C0 + C1;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * C) return;
  const size_t m = i / C;
  const int c = (int)(i - m * C);
  out[i] = c < C0 ? s0[m * C0 + c] : s1[m * C1 + (c - C0)];
}

// per-channel batch statistics over the M rows (two passes: mean, then centred second moment); 64 channels per
// block, 4 row groups.  Updates the running statistics the way nn.BatchNorm1d does in train mode.
__global__ __launch_bounds__(256) void tcn_bn_stats_kernel(const float* __restrict__ y, int M, int H, float momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ stats /* [2][H]: mean, biased var */) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.0f;
  if (c < H)
    for (int m = g; m < M; m += 4) s += y[(size_t)m * H + c];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  const float mean = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)M;
  __syncthreads();
  float q = 0.0f;
  if (c < H)
    for (int m = g; m < M; m += 4) {
      const float d = y[(size_t)m * H + c] - mean;
      q += d * d;
    }
  red[g][threadIdx.x & 63] = q;
  __syncthreads();
  if (g == 0 && c < H) {
    const float ss = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const float var = ss / (float)M;
    stats[c] = mean;
    stats[H + c] = var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (ss / (float)(M - 1));
  }
}

// h_out = (res ? res : 0) + dropout(gelu((y - mean) * rsqrt(var + eps) * gamma + beta))
__global__ __launch_bounds__(256) void tcn_bn_act_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ var, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ res, size_t n, int H,
                                                         float eps, float drop_p, uint32_t layer, const ufnd_step_state* st,
                                                         float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % H);
  const float invstd = 1.0f / sqrtf(var[c] + eps);
  float z = gelu_f((y[i] - mean[c]) * invstd * gamma[c] + beta[c]);
  if (drop_p > 0.0f) z *= dropout_mul(st, drop_p, layer, (uint32_t)i);
  out[i] = res ? res[i] + z : z;
}

// pooled[b] = [mean_t h(b, t, :), max_t h(b, t, :)]
__global__ __launch_bounds__(256) void tcn_pool_kernel(const float* __restrict__ h, int T, int H, float* __restrict__ pooled) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= H) return;
  const float* p = h + (size_t)b * T * H + c;
  float s = 0.0f, mx = p[0];
  for (int t = 0; t < T; ++t) {
    const float v = p[(size_t)t * H];
    s += v;
    mx = fmaxf(mx, v);
  }
  pooled[(size_t)b * 2 * H + c] = s / (float)T;
  pooled[(size_t)b * 2 * H + H + c] = mx;
}

}  // namespace

extern "C" int ufnd_tcn_weight_ld(int in_ch, int kernel) { return (in_ch * kernel + 3) & ~3; }

extern "C" size_t ufnd_tcn_workspace_floats(int B, int T, int in_ch, int hid, int kernel) {
  if (B < 1 || T < 1 || in_ch < 1 || hid < 1 || kernel < 1) return 0;
  const size_t M = (size_t)B * T;
  const int wide = in_ch > hid ? in_ch : hid;
  return up64(M * ufnd_tcn_weight_ld(wide, kernel)) + 3 * up64(M * wide) + up64(2 * (size_t)hid) + up64((size_t)B * 2 * hid) + 64;
}

extern "C" int ufnd_tcn_forward(const float* text_seq, int text_dim, const float* vis_seq, int vis_dim, int B, int T,
                                const ufnd_tcn_layer* layers, int n_layers, int kernel, int hid, const float* head_w,
                                const float* head_b, int out_dim, int train, float dropout_p, float momentum, float eps,
                                const ufnd_step_state* state, float* workspace, float* out, void* stream_) {
  UFND_REQUIRE(text_seq && vis_seq && layers && head_w && head_b && workspace && out, "tcn_forward: null argument");
  UFND_REQUIRE(B >= 1 && T >= 1 && text_dim >= 1 && vis_dim >= 1 && n_layers >= 1 && n_layers <= 16 && kernel >= 1 && kernel <= 15,
               "tcn_forward: B=%d T=%d dims=%d+%d layers=%d kernel=%d", B, T, text_dim, vis_dim, n_layers, kernel);
  UFND_REQUIRE(hid % 32 == 0 && out_dim % 32 == 0, "tcn_forward: hid=%d out=%d must be multiples of 32", hid, out_dim);
  UFND_REQUIRE((long long)B * T < (1ll << 24), "tcn_forward: B*T too large");
  UFND_REQUIRE(!train || (long long)B * T > 1, "tcn_forward: batch statistics need more than one value per channel");
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (!(train && dropout_p > 0.0f) || state), "tcn_forward: dropout needs a step state");
  UFND_REQUIRE(ufnd_aligned(workspace, 16) && ufnd_aligned(out, 16) && ufnd_aligned(head_w, 16) && ufnd_aligned(head_b, 16),
               "tcn_forward: 16-B alignment required");
  for (int i = 0; i < n_layers; ++i) {
    const ufnd_tcn_layer& l = layers[i];
    UFND_REQUIRE(l.w && l.b && l.gamma && l.beta && l.running_mean && l.running_var, "tcn_forward: layer %d has a null tensor", i);
    UFND_REQUIRE(ufnd_aligned(l.w, 16) && ufnd_aligned(l.b, 16), "tcn_forward: layer %d weight/bias must be 16-B aligned", i);
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int C = text_dim + vis_dim;
  const size_t M = (size_t)B * T;
  const int wide = C > hid ? C : hid;
  float* cols = workspace;
  float* y = cols + up64(M * ufnd_tcn_weight_ld(wide, kernel));
  float* hA = y + up64(M * wide);
  float* hB = hA + up64(M * wide);
  float* stats = hB + up64(M * wide);
  float* pooled = stats + up64(2 * (size_t)hid);
  const float p = train ? dropout_p : 0.0f;

  const float* h = nullptr;      // current activations (M, ch); null = still the two input sequences
  int ch = C;
  if (C == hid) {                // the first block is residual too: it needs the concatenated input as one matrix
    hipLaunchKernelGGL(tcn_concat_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, stream, text_seq, text_dim, vis_seq,
                       vis_dim, M, hA);
    UFND_CHECK_LAUNCH();
    h = hA;
  }
  for (int i = 0; i < n_layers; ++i) {
    const ufnd_tcn_layer& l = layers[i];
    const int dil = 1 << i;
    const int left = (dil * (kernel - 1)) / 2;          // padding='same': total = dil*(k-1), left = total/2 (torch's split)
    const int ldc = ufnd_tcn_weight_ld(ch, kernel);
    if (h)
      hipLaunchKernelGGL(tcn_unfold_kernel, dim3((unsigned)M), dim3(256), 0, stream, h, ch, (const float*)nullptr, 0, T, kernel, dil,
                         left, cols, ldc);
    else
      hipLaunchKernelGGL(tcn_unfold_kernel, dim3((unsigned)M), dim3(256), 0, stream, text_seq, text_dim, vis_seq, vis_dim, T, kernel,
                         dil, left, cols, ldc);
    UFND_CHECK_LAUNCH();
    NtProb conv{cols, l.w, l.b, y, nullptr, (int)M, hid, kernel * ch, ldc, ldc, hid, 0, 0, 0.0f, 0, 1};
    int rc = launch_nt(&conv, 1, nullptr, stream);
    if (rc != UFND_OK) return rc;
    const float *mean = l.running_mean, *var = l.running_var;
    if (train) {
      hipLaunchKernelGGL(tcn_bn_stats_kernel, dim3(ufnd_cdiv(hid, 64)), dim3(256), 0, stream, y, (int)M, hid, momentum, l.running_mean,
                         l.running_var, stats);
      UFND_CHECK_LAUNCH();
      mean = stats;
      var = stats + hid;
    }
    const float* res = (h && ch == hid) ? h : nullptr;
    float* dst = (h == hA) ? hB : hA;
    hipLaunchKernelGGL(tcn_bn_act_kernel, dim3((unsigned)((M * hid + 255) / 256)), dim3(256), 0, stream, y, mean, var, l.gamma, l.beta,
                       res, M * hid, hid, eps, p, LAYER_TCN + (uint32_t)i, state, dst);
    UFND_CHECK_LAUNCH();
    h = dst;
    ch = hid;
  }
  hipLaunchKernelGGL(tcn_pool_kernel, dim3(ufnd_cdiv(hid, 256), B), dim3(256), 0, stream, h, T, hid, pooled);
  UFND_CHECK_LAUNCH();
  NtProb head{pooled, head_w, head_b, out, nullptr, B, out_dim, 2 * hid, 2 * hid, 2 * hid, out_dim, 0, 0, 0.0f, 0, 1};
  return launch_nt(&head, 1, nullptr, stream);
}
