// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ultrafnd_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define UFND_WAVE 64

// ------------------------------------------------------------------ error plumbing (host)
void ufnd_set_error(const char* fmt, ...);
#define UFND_REQUIRE(cond, ...)                      \
  do {                                               \
    if (!(cond)) {                                   \
      ufnd_set_error(__VA_ARGS__);                   \
      return UFND_ERR_INVALID;                       \
    }                                                \
  } while (0)
#define UFND_CHECK_LAUNCH()                                              \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      ufnd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                     hipGetErrorString(e_));                             \
      return UFND_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

static inline bool ufnd_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int ufnd_cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ math
__device__ __forceinline__ float gelu_f(float x) {            // nn.GELU() default: exact erf
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// XCD-aware block order (bijective): hardware block id n lands on XCD n % 8 (observed round-robin placement: speed only, never
// correctness); the logical id returned gives every XCD a CONTIGUOUS range of the grid, so that workgroups with neighbouring
// logical ids -- the query blocks of one (sample, head), which walk the same K / V rows -- share one XCD's L2 instead of fetching
// those rows through up to eight of them.
__device__ __forceinline__ int xcd_contiguous_id(int n, int total) {
  const int q = total >> 3, r = total & 7, xcd = n & 7, idx = n >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 2^x as ONE v_exp_f32 (exp2f wraps it in five instructions of denormal-range scaling: compare, select, add, select, ldexp).  The
// attention kernels call it with x <= 0 on softmax scores: the same bits as exp2f down to 2^-126, zero below (a probability of
// 1e-38 is zero after the bf16 rounding that follows anyway); a 64-key block needs 32 of them per lane.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {      // d/dx [x sigmoid(1.702 x)]
  const float s = 1.0f / (1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
// bf16-output epilogues.  Contraction is switched OFF inside them (HIP's __fmul_rn / __fadd_rn are plain
// operators, which the compiler may still fuse differently per kernel instantiation) and every fused
// multiply-add is spelled out: every GEMM tile shape must emit the same operation sequence, so that a
// row's result never depends on the batch it is computed in.
// v_exp + v_rcp (1 ulp) instead of the IEEE division ladder.
__device__ __forceinline__ float sigmoid_fast_f(float x) {
#pragma clang fp contract(off)
  return __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-x)));
}
__device__ __forceinline__ float quick_gelu_fast_f(float x) {      // x * sigmoid(1.702 x)
#pragma clang fp contract(off)
  return __fmul_rn(x, sigmoid_fast_f(__fmul_rn(1.702f, x)));
}
// erf-GELU for bf16 outputs: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below a bf16 ulp),
// one v_exp + one v_rcp + 8 FMAs instead of libm's erff polynomial ladder.
__device__ __forceinline__ float gelu_fast_f(float x) {
#pragma clang fp contract(off)
  const float z = __fmul_rn(fabsf(x), 0.70710678118654752440f);
  const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, z, 1.0f));   // v_rcp_f32; __frcp_rn expands to the division ladder
  float p = __fmaf_rn(t, 1.061405429f, -1.453152027f);
  p = __fmaf_rn(t, p, 1.421413741f);
  p = __fmaf_rn(t, p, -0.284496736f);
  p = __fmaf_rn(t, p, 0.254829592f);
  const float poly = __fmul_rn(t, p);
  const float erf_abs = __fmaf_rn(-poly, __expf(__fmul_rn(-z, z)), 1.0f);
  const float h = __fmul_rn(0.5f, x);
  return __fmaf_rn(h, copysignf(erf_abs, x), h);
}

// Two elements at a time on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32: one issue slot for two lanes' worth of
// IEEE operations -- the same roundings as the scalar forms above, element for element, so results are bit-identical; the
// reciprocal and the exponential stay scalar).  The GELU epilogue of a 256x192 FFN1 tile is ~6 us of vector issue.
__device__ __forceinline__ f32x2 gelu_fast_f2(f32x2 x) {
#pragma clang fp contract(off)
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 u = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
  f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(t, p, f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(t, p, f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(t, p, f32x2{0.254829592f, 0.254829592f});
  const f32x2 poly = t * p;
  const f32x2 zz = (-z) * z;
  const f32x2 e = {__expf(zz.x), __expf(zz.y)};
  const f32x2 erf_abs = __builtin_elementwise_fma(-poly, e, f32x2{1.0f, 1.0f});
  const f32x2 h = x * 0.5f;
  const f32x2 sg = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
  return __builtin_elementwise_fma(h, sg, h);
}
__device__ __forceinline__ f32x2 quick_gelu_fast_f2(f32x2 x) {
#pragma clang fp contract(off)
  const f32x2 m = x * 1.702f;
  const f32x2 d = f32x2{__expf(-m.x), __expf(-m.y)} + 1.0f;
  const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return x * r;
}

// butterfly partner inside a quad (lanes 4k..4k+3) by DPP quad_perm: no LDS round trip (a __shfl_xor compiles to
// ds_bpermute, ~100 cycles of dependent latency each)
__device__ __forceinline__ float quad_xor1(float v) {       // lane ^ 1: quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_xor2(float v) {       // lane ^ 2: quad_perm [2,3,0,1]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}

// Wave-wide reductions by DPP (no LDS round trips: a __shfl_xor butterfly is six dependent ds_bpermute, ~100 cycles
// each -- the row kernels of the head spend most of their time in them).  Rows of 16 lanes reduce by quad_perm /
// row_half_mirror / row_mirror, the four rows combine by row_bcast15 / row_bcast31 (wave64), the total lands in lane 63
// and is broadcast through an SGPR.  Every lane returns the same value; the association order is fixed.
#define UFND_DPP(v, old, ctrl, rmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, rmask, 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += UFND_DPP(v, 0.0f, 0xB1, 0xF);     // quad_perm [1,0,3,2]
  v += UFND_DPP(v, 0.0f, 0x4E, 0xF);     // quad_perm [2,3,0,1]
  v += UFND_DPP(v, 0.0f, 0x141, 0xF);    // row_half_mirror
  v += UFND_DPP(v, 0.0f, 0x140, 0xF);    // row_mirror: every lane holds its row's sum
  v += UFND_DPP(v, 0.0f, 0x142, 0xA);    // row_bcast15 into rows 1 and 3
  v += UFND_DPP(v, 0.0f, 0x143, 0xC);    // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, UFND_DPP(v, v, 0xB1, 0xF));
  v = fmaxf(v, UFND_DPP(v, v, 0x4E, 0xF));
  v = fmaxf(v, UFND_DPP(v, v, 0x141, 0xF));
  v = fmaxf(v, UFND_DPP(v, v, 0x140, 0xF));
  v = fmaxf(v, UFND_DPP(v, v, 0x142, 0xA));     // (lanes outside the row mask keep `old` = their own value)
  v = fmaxf(v, UFND_DPP(v, v, 0x143, 0xC));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ------------------------------------------------------------------ counter-based dropout RNG
// Philox4x32-10 keyed by the run's seed; counter = (element / 4, layer tag, step), element e takes output word e % 4.  The same
// (seed, step, layer, element) always yields the same keep/drop decision, so backward regenerates the forward mask instead of
// storing it.  One evaluation (10 rounds, ~110 instructions) serves four consecutive elements: the row kernels and GEMM epilogues
// that own four aligned elements per lane call dropout_mul4 (round 3; until then every element ran its own evaluation and kept one
// word of four -- in the GEMM epilogues that was the longest piece of straight-line code).
__device__ __forceinline__ void philox_4x32(uint64_t seed, uint64_t step, uint32_t layer, uint32_t ctr, uint32_t (&out)[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = ctr, c1 = layer, c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float dropout_keep(uint32_t r, float p) {
  const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
  return (u >= p) ? 1.0f / (1.0f - p) : 0.0f;
}
// multiplier applied to an activation: 0 (dropped) or 1/(1-p) (kept).  p == 0 -> 1.
__device__ __forceinline__ float dropout_mul(const ufnd_step_state* st, float p, uint32_t layer, uint32_t elem) {
  if (p <= 0.0f) return 1.0f;
  uint32_t w[4];
  philox_4x32(st->seed, st->step, layer, elem >> 2, w);
  const uint32_t k = elem & 3u;
  return dropout_keep(k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3])), p);
}
// the multipliers of elements elem4 .. elem4 + 3 (elem4 a multiple of 4): one evaluation
__device__ __forceinline__ void dropout_mul4(const ufnd_step_state* st, float p, uint32_t layer, uint32_t elem4, float (&m)[4]) {
  if (p <= 0.0f) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
  uint32_t w[4];
  philox_4x32(st->seed, st->step, layer, elem4 >> 2, w);
#pragma unroll
  for (int q = 0; q < 4; ++q) m[q] = dropout_keep(w[q], p);
}
