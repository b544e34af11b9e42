// clip_grad_norm_ + AdamW over one flat fp32 arena (forensic_trainer.py:292-298,176).
// The reference runs ~104 per-tensor norm kernels and a foreach AdamW (56 % of its step,
// SURVEY.md 3); here parameters, gradients and both moments are four flat HBM arrays, so the
// whole optimizer is two streaming passes: 4 B/param read for the norm, 28 B/param for AdamW.
#include "common.hpp"

// The update of one element is ONE function shared by both AdamW kernels, with every fused multiply-add spelled out and
// implicit contraction off: the two-launch and the four-launch form of the optimizer must produce the same bits, and a
// compiler is free to contract a*b + c*d either way round in two separately compiled kernels (it did).
#pragma clang fp contract(off)

namespace {

struct AdamScalars { float b1, b2, eps, decay, gs, step_size, inv_bc2; };
__device__ __forceinline__ void adamw_vec(f32x4& pp, f32x4& mm, f32x4& vv, const f32x4 g_raw, const AdamScalars& k) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float gq = g_raw[q] * k.gs;
    const float pq = pp[q] * k.decay;
    const float mq = __builtin_fmaf(mm[q], k.b1, (1.0f - k.b1) * gq);
    const float vq = __builtin_fmaf(vv[q], k.b2, ((1.0f - k.b2) * gq) * gq);
    const float denom = __builtin_fmaf(sqrtf(vq), k.inv_bc2, k.eps);
    pp[q] = __builtin_fmaf(-k.step_size, mq / denom, pq);
    mm[q] = mq;
    vv[q] = vq;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, size_t n4, float* partials) {
  __shared__ float sh[4];
  float s = 0.0f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// one block: fixed-order sum of the partials (bit-reproducible), then the step's scalars
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* partials, int nblocks, ufnd_step_state* st) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(sh[0]) * fabsf(st->grad_scale);
    st->grad_norm = total;
    float coef = 1.0f;
    if (st->max_norm > 0.0f) coef = fminf(1.0f, st->max_norm / (total + 1e-6f));
    st->clip_coef = coef;
    const double t = (double)(st->step + 1);
    st->bc1 = (float)(1.0 - pow((double)st->beta1, t));
    st->bc2_sqrt = (float)sqrt(1.0 - pow((double)st->beta2, t));
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, size_t n4,
                                                    const ufnd_step_state* st) {
  const float lr = st->lr, b1 = st->beta1, b2 = st->beta2, eps = st->eps;
  const float decay = 1.0f - lr * st->weight_decay;
  const float gs = st->grad_scale * st->clip_coef;
  const float step_size = lr / st->bc1, inv_bc2 = 1.0f / st->bc2_sqrt;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  const AdamScalars k{b1, b2, eps, decay, gs, step_size, inv_bc2};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
    adamw_vec(pp, mm, vv, g4[i], k);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
}

__global__ void step_advance_kernel(ufnd_step_state* st) { st->step += 1; }

// ---- the whole optimizer step as TWO launches (ufnd_clip_adamw_step)
// launch 1: the sum-of-squares partials, as above; its block 0 also advances the step counter, so that launch 2 reads
// the number of the step it applies (and the next forward's dropout masks get a new key) without a launch of its own.
__global__ __launch_bounds__(256) void sumsq_advance_kernel(const float* g, size_t n4, float* partials, ufnd_step_state* st) {
  __shared__ float sh[4];
  float s = 0.0f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (blockIdx.x == 0) st->step += 1;
  }
}
// launch 2: every block repeats norm_finalize_kernel's fixed-order reduction of the partials (<= 1024 floats from L2: the
// same bits in every block and in the three-launch form), derives the clip coefficient and the bias corrections of
// step t = st->step (already advanced), and applies AdamW to its share; block 0 publishes the scalars.
__global__ __launch_bounds__(256) void adamw_clip_kernel(float* p, const float* g, float* m, float* v, size_t n4, const float* partials,
                                                         int nblocks, ufnd_step_state* st) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const float total = (float)sqrt(sh[0]) * fabsf(st->grad_scale);
  float coef = 1.0f;
  if (st->max_norm > 0.0f) coef = fminf(1.0f, st->max_norm / (total + 1e-6f));
  const double t = (double)st->step;
  const float bc1 = (float)(1.0 - pow((double)st->beta1, t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)st->beta2, t));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->grad_norm = total;
    st->clip_coef = coef;
    st->bc1 = bc1;
    st->bc2_sqrt = bc2_sqrt;
  }
  const float lr = st->lr, b1 = st->beta1, b2 = st->beta2, eps = st->eps;
  const float decay = 1.0f - lr * st->weight_decay;
  const float gs = st->grad_scale * coef;
  const float step_size = lr / bc1, inv_bc2 = 1.0f / bc2_sqrt;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  const AdamScalars k{b1, b2, eps, decay, gs, step_size, inv_bc2};
#ifndef UFND_ADAMW_UNROLL
#define UFND_ADAMW_UNROLL 2
#endif
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if constexpr (UFND_ADAMW_UNROLL == 2) {      // two elements' loads in flight before the first store (the arrays alias nothing, but the compiler cannot know)
    for (; i + stride < n4; i += 2 * stride) {
      const size_t j = i + stride;
      f32x4 pa = p4[i], ma = m4[i], va = v4[i], ga = g4[i];
      f32x4 pb = p4[j], mb = m4[j], vb = v4[j], gb = g4[j];
      adamw_vec(pa, ma, va, ga, k);
      adamw_vec(pb, mb, vb, gb, k);
      p4[i] = pa; m4[i] = ma; v4[i] = va;
      p4[j] = pb; m4[j] = mb; v4[j] = vb;
    }
  }
  for (; i < n4; i += stride) {
    f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
    adamw_vec(pp, mm, vv, g4[i], k);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
}

}  // namespace

extern "C" int ufnd_grad_norm(const float* grad, size_t n, float* partials, ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(grad && partials && state && n > 0, "grad_norm: null argument");
  UFND_REQUIRE(n % 4 == 0 && ufnd_aligned(grad, 16), "grad_norm: n %% 4 == 0 and 16-B alignment required");
  hipStream_t stream = (hipStream_t)stream_;
  size_t want = (n / 4 + 1023) / 1024;  // >= 4 float4 per thread
  const int blocks = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, grad, n / 4, partials);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)partials, blocks, state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                               const ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && n > 0, "adamw_step: null argument");
  UFND_REQUIRE(n % 4 == 0 && ufnd_aligned(param, 16) && ufnd_aligned(grad, 16) && ufnd_aligned(exp_avg, 16) &&
                   ufnd_aligned(exp_avg_sq, 16), "adamw_step: n %% 4 == 0 and 16-B alignment required");
  size_t want = (n / 4 + 511) / 512;
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq,
                     n / 4, state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_step_advance(ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(state, "step_advance: null state");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// clip_grad_norm_ + AdamW.step + the step counter as TWO launches (ufnd_grad_norm + ufnd_adamw_step + ufnd_step_advance are
// four): the same arithmetic in the same order -- parameters, moments and the published scalars are bit-identical.
extern "C" int ufnd_clip_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* partials,
                                    ufnd_step_state* state, void* stream_) {
  UFND_REQUIRE(param && grad && exp_avg && exp_avg_sq && partials && state && n > 0, "clip_adamw_step: null argument");
  UFND_REQUIRE(n % 4 == 0 && ufnd_aligned(param, 16) && ufnd_aligned(grad, 16) && ufnd_aligned(exp_avg, 16) &&
                   ufnd_aligned(exp_avg_sq, 16), "clip_adamw_step: n %% 4 == 0 and 16-B alignment required");
  hipStream_t stream = (hipStream_t)stream_;
  size_t want = (n / 4 + 1023) / 1024;
  const int nb = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  hipLaunchKernelGGL(sumsq_advance_kernel, dim3(nb), dim3(256), 0, stream, grad, n / 4, partials, state);
  UFND_CHECK_LAUNCH();
  size_t want2 = (n / 4 + 511) / 512;
  const int blocks = (int)(want2 < 1 ? 1 : (want2 > 2048 ? 2048 : want2));
  hipLaunchKernelGGL(adamw_clip_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n / 4, (const float*)partials, nb, state);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
