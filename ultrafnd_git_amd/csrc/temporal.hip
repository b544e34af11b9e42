// TemporalSyncNet.align (src/core_blocks/temporal_blocks.py:102-140), batched:
//   feat = [t, v^, t - v^, t * v^, cos(t, v^)]  (v^ = v zero-padded / truncated to D), 4D+1 wide
//   out  = W3 drop(GELU(W0 feat + b0)) + b3      fp32, forward only (the reference never trains it).  The reference's
//          align() is decorated with torch.inference_mode, which switches autograd off, NOT dropout: the module's
//          Dropout(0.1) is live whenever the module is in its default train mode (the cache builder never calls .eval())
// One wave per sample builds feat (row stride padded to a multiple of 4, pad columns zero); the two
// Linears run on the fp32 MFMA skinny GEMM (gemm_f32.hip) -- W0 must be stored with that same padded
// row stride (ufnd_temporal_weight_ld) because 4D+1 is odd.
#include "gemm_f32.hpp"

namespace {

__global__ __launch_bounds__(256) void align_features_kernel(const float* t, const float* v, int ldv, int B, int D, int Dv,
                                                             float* feat, int ldf) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* tr = t + (size_t)row * D;
  const float* vr = v + (size_t)row * ldv;
  float tt = 0, vv = 0, tv = 0;
  for (int c = lane; c < D; c += 64) {
    const float a = tr[c], b = c < Dv ? vr[c] : 0.0f;
    tt += a * a; vv += b * b; tv += a * b;
  }
  tt = wave_sum(tt); vv = wave_sum(vv); tv = wave_sum(tv);
  // _cosine (:10-13): sum((a / (|a| + eps)) * (b / (|b| + eps)))
  const float cosv = tv / ((sqrtf(tt) + 1e-9f) * (sqrtf(vv) + 1e-9f));
  float* f = feat + (size_t)row * ldf;
  for (int c = lane; c < D; c += 64) {
    const float a = tr[c], b = c < Dv ? vr[c] : 0.0f;
    f[c] = a; f[D + c] = b; f[2 * D + c] = a - b; f[3 * D + c] = a * b;
  }
  for (int c = 4 * D + lane; c < ldf; c += 64) f[c] = (c == 4 * D) ? cosv : 0.0f;
}

}  // namespace

extern "C" int ufnd_temporal_weight_ld(int in_dim) { return (4 * in_dim + 1 + 3) & ~3; }

extern "C" size_t ufnd_temporal_workspace_floats(int B, int in_dim, int hidden) {
  if (B < 1) return 0;
  return (size_t)B * ufnd_temporal_weight_ld(in_dim) + (size_t)B * hidden + 128;
}

extern "C" int ufnd_temporal_align(const float* text, const float* visual, const float* w0, const float* b0, const float* w3,
                                   const float* b3, float* workspace, float* out, int B, int in_dim, int vis_dim, int hidden,
                                   int out_dim, float dropout_p, const ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(text && visual && w0 && b0 && w3 && b3 && workspace && out, "temporal_align: null argument");
  UFND_REQUIRE(B >= 1 && in_dim >= 1 && vis_dim >= 1 && hidden % 32 == 0 && out_dim % 32 == 0,
               "temporal_align: B=%d D=%d Dv=%d hidden=%d out=%d (hidden/out multiples of 32)", B, in_dim, vis_dim, hidden, out_dim);
  UFND_REQUIRE(ufnd_aligned(workspace, 16) && ufnd_aligned(w0, 16) && ufnd_aligned(w3, 16) && ufnd_aligned(out, 16),
               "temporal_align: 16-B alignment required");
  UFND_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && (dropout_p == 0.0f || state), "temporal_align: dropout_p=%g needs a step state", dropout_p);
  hipStream_t stream = (hipStream_t)stream_;
  const int ldf = ufnd_temporal_weight_ld(in_dim);
  float* feat = workspace;
  float* h = workspace + (((size_t)B * ldf + 63) & ~(size_t)63);
  hipLaunchKernelGGL(align_features_kernel, dim3(ufnd_cdiv(B, 4)), dim3(256), 0, stream, text, visual, vis_dim, B, in_dim,
                     vis_dim < in_dim ? vis_dim : in_dim, feat, ldf);
  UFND_CHECK_LAUNCH();
  // note: visual rows are read with their own stride vis_dim; columns >= in_dim are ignored (truncate)
  constexpr uint32_t LAYER_ALIGN = 21;       // dropout stream tag (mask keyed by state->{seed, step})
  NtProb p0{feat, w0, b0, h, nullptr, B, hidden, 4 * in_dim + 1, ldf, ldf, hidden, 0, 1, dropout_p, LAYER_ALIGN, 1};
  int rc = launch_nt(&p0, 1, dropout_p > 0.0f ? state : nullptr, stream);
  if (rc != UFND_OK) return rc;
  NtProb p1{h, w3, b3, out, nullptr, B, out_dim, hidden, hidden, hidden, out_dim, 0, 0, 0.0f, 0, 1};
  return launch_nt(&p1, 1, nullptr, stream);
}
