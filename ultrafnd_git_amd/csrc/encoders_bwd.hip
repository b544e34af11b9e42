// Row-wise backward pieces of the two encoders (Tier-B backward, SURVEY.md 8b: layernorm_bwd, masked_meanpool_l2_bwd, the
// embedding / patch-embedding gradients).  The reference keeps its encoders frozen (src/core_blocks/text_blocks.py:52,63);
// these serve TrainConfig.train_encoders.  All HBM-bound row kernels: one wave per row, 16-B loads, DPP wave reductions.
// Parameter gradients that sum over rows are two-stage and ordered (block partials, then a finish pass over the blocks in
// ascending order): no atomics, bitwise reproducible.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// y = (x - mean) rstd gamma + beta  ->  dx = rstd (g - mean(g) - xh mean(g xh)),  g = dy gamma,  xh = (x - mean) rstd
// (+ `add`: the gradient arriving over the residual branch).  dgamma / dbeta partials per block: part[blk][0/1][H].
template <int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, int lddy, const float* __restrict__ add, int ldadd,
                                                            float* __restrict__ dx, __bf16* __restrict__ dxb, int lddx, float* __restrict__ part,
                                                            int M, int H, float eps) {
  __shared__ f32x4 sh[2][4][NI][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 dg[NI], db[NI], gm[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    gm[i] = ld4(gamma + 4 * lane + 256 * i);
  }
  const float inv_h = 1.0f / (float)H;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    f32x4 v[NI], g[NI];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      v[i] = ld4(x + (size_t)row * ldx + 4 * lane + 256 * i);
      g[i] = ld4(dy + (size_t)row * lddy + 4 * lane + 256 * i);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) * inv_h;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = v[i][k] - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(wave_sum(q) * inv_h + eps);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (v[i][k] - mean) * rstd;
        const float gy = g[i][k] * gm[i][k];
        dg[i][k] += g[i][k] * xh;
        db[i][k] += g[i][k];
        v[i][k] = xh;
        g[i][k] = gy;
        s1 += gy;
        s2 += gy * xh;
      }
    s1 = wave_sum(s1) * inv_h;
    s2 = wave_sum(s2) * inv_h;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * lane + 256 * i;
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = rstd * (g[i][k] - s1 - v[i][k] * s2);
      if (add) o += ld4(add + (size_t)row * ldadd + col);
      if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)row * lddx + col) = o;
      if (dxb) {
        bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
        *reinterpret_cast<bf16x4*>(dxb + (size_t)row * lddx + col) = ob;
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sh[0][wave][i][lane] = dg[i];
    sh[1][wave][i][lane] = db[i];
  }
  __syncthreads();
  if (wave < 2) {        // wave 0 finishes dgamma, wave 1 dbeta: waves added in ascending order
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const f32x4 t = ((sh[wave][0][i][lane] + sh[wave][1][i][lane]) + sh[wave][2][i][lane]) + sh[wave][3][i][lane];
      *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + wave) * H + 4 * lane + 256 * i) = t;
    }
  }
}

// out[which][c] (+)= sum over blocks of part[blk][which][c].  grid (H / 16, 2): 16 columns x 16 block-groups per workgroup, group j
// adds blocks j, j + 16, ... in ascending order, the groups combine in a fixed tree (three workgroups of serial adders took 89 us
// per call: 22 % of a training step).
__global__ __launch_bounds__(256) void row_partials_finish_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ out0,
                                                                  float* __restrict__ out1, int accumulate) {
  __shared__ float sh[16][17];
  const int which = blockIdx.y, cl = threadIdx.x & 15, grp = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float* out = which ? out1 : out0;
  if (!out) return;                     // (block-uniform)
  float s = 0.f;
  if (c < H)
    for (int k = grp; k < nblk; k += 16) s += part[((size_t)k * 2 + which) * H + c];
  sh[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < H) {
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = sh[k][cl];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; ++k) a[k] += a[k + w];
    out[c] = accumulate ? out[c] + a[0] : a[0];
  }
}

// masked mean-pool + L2 backward (text_blocks.py:82-86,100).
//   rep = sum_t m_t h_t / max(count, 1e-6);  f = rep / (|rep| + 1e-9)
//   drep = df / (n + e) - rep (rep . df) / (n (n + e)^2);  dh_t = m_t / max(count, 1e-6) drep
// grid (B, MP_SPLIT): every block of a sample rebuilds rep (its 4 row groups take tokens l = g, g + 4, ... with 16-B loads, eight
// in flight; the groups meet in LDS and are added in group order) and writes the gradient rows of its quarter of the tokens.  (One
// block of 256 threads per sample walking the tokens one 4-B load at a time took 189 us at 32 x 128 tokens: the 32 blocks'
// dependent loads, not bytes.)
constexpr int MP_SPLIT = 4, MP_GROUPS = 4;
__global__ __launch_bounds__(1024) void meanpool_l2_bwd_kernel(const float* __restrict__ hidden, const int32_t* __restrict__ mask,
                                                               const float* __restrict__ dfeat, float* __restrict__ dhidden, int L, int H) {
  __shared__ f32x4 acc[MP_GROUPS][256];
  __shared__ float red[2][16];
  __shared__ float cnt_sh[MP_GROUPS];
  const int b = blockIdx.x, tid = threadIdx.x, g = tid >> 8, c4 = tid & 255;      // 256 threads x 4 columns cover H <= 1024
  const bool live_col = 4 * c4 < H;
  f32x4 rep = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.0f;
  const int32_t* mrow = mask + (size_t)b * L;
  const float* hb = hidden + (size_t)b * L * H + 4 * c4;
  for (int l0 = g; l0 < L; l0 += MP_GROUPS * 8) {
    f32x4 v[8];
    bool on[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int l = l0 + MP_GROUPS * u;
      on[u] = l < L && mrow[l] != 0;
      v[u] = (on[u] && live_col) ? ld4(hb + (size_t)l * H) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (on[u]) { rep += v[u]; cnt += 1.0f; }
  }
  acc[g][c4] = rep;
  if (c4 == 0) cnt_sh[g] = cnt;
  __syncthreads();
  rep = ((acc[0][c4] + acc[1][c4]) + acc[2][c4]) + acc[3][c4];
  cnt = ((cnt_sh[0] + cnt_sh[1]) + cnt_sh[2]) + cnt_sh[3];
  const float denom = fmaxf(cnt, 1e-6f);
  const f32x4 df = live_col ? ld4(dfeat + (size_t)b * H + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  float sq = 0.0f, dot = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rep[k] /= denom;
    sq += rep[k] * rep[k];
    dot += rep[k] * df[k];
  }
  if (g != 0) { sq = 0.0f; dot = 0.0f; }      // (every group holds the same rep: group 0's lanes carry the sums)
  sq = wave_sum(sq);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sq; red[1][tid >> 6] = dot; }
  __syncthreads();
  const float nrm = sqrtf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
  const float dt = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float ne = nrm + 1e-9f;
  const float k1 = 1.0f / ne, k2 = nrm > 0.0f ? dt / (nrm * ne * ne) : 0.0f;
  f32x4 drep;
#pragma unroll
  for (int k = 0; k < 4; ++k) drep[k] = (df[k] * k1 - rep[k] * k2) / denom;
  if (!live_col) return;
  const int per = (L + MP_SPLIT - 1) / MP_SPLIT, l_lo = blockIdx.y * per, l_hi = l_lo + per < L ? l_lo + per : L;
  float* db_ = dhidden + (size_t)b * L * H + 4 * c4;
  for (int l = l_lo + g; l < l_hi; l += MP_GROUPS)
    *reinterpret_cast<f32x4*>(db_ + (size_t)l * H) = mrow[l] != 0 ? drep : f32x4{0.f, 0.f, 0.f, 0.f};
}

// frame pooling backward (l2norm_frames_kernel): u_f = e_f / (|e_f| + eps); F == 1: feat = u_0; else m = mean_f u_f, feat = m / (|m| + eps)
__global__ __launch_bounds__(256) void l2norm_frames_bwd_kernel(const float* __restrict__ e, const float* __restrict__ dfeat, float* __restrict__ de, int F, int D) {
  __shared__ float sh[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  auto block2 = [&](float a, float c, float& oa, float& oc) {
    a = wave_sum(a);
    c = wave_sum(c);
    __syncthreads();
    if ((tid & 63) == 0) { sh[0][tid >> 6] = a; sh[1][tid >> 6] = c; }
    __syncthreads();
    oa = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    oc = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  };
  float du[4];            // gradient with respect to every u_f (the same for all frames)
  {
    int n = 0;
    for (int c = tid; c < D; c += 256, ++n) du[n] = dfeat[(size_t)b * D + c];
  }
  if (F > 1) {
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < F; ++f) {
      const float* row = e + ((size_t)b * F + f) * D;
      float sq = 0.0f, dummy = 0.0f, o1, o2;
      for (int c = tid; c < D; c += 256) sq += row[c] * row[c];
      block2(sq, dummy, o1, o2);
      const float nrm = sqrtf(o1) + 1e-9f;
      int n = 0;
      for (int c = tid; c < D; c += 256, ++n) m[n] += row[c] / nrm;
    }
    float sq = 0.0f, dot = 0.0f, o1, o2;
    {
      int n = 0;
      for (int c = tid; c < D; c += 256, ++n) {
        m[n] /= (float)F;
        sq += m[n] * m[n];
        dot += m[n] * du[n];
      }
    }
    block2(sq, dot, o1, o2);
    const float nm = sqrtf(o1), ne = nm + 1e-9f;
    const float k1 = 1.0f / ne, k2 = nm > 0.0f ? o2 / (nm * ne * ne) : 0.0f;
    int n = 0;
    for (int c = tid; c < D; c += 256, ++n) du[n] = (du[n] * k1 - m[n] * k2) / (float)F;
  }
  for (int f = 0; f < F; ++f) {
    const float* row = e + ((size_t)b * F + f) * D;
    float sq = 0.0f, dot = 0.0f, o1, o2;
    {
      int n = 0;
      for (int c = tid; c < D; c += 256, ++n) {
        sq += row[c] * row[c];
        dot += row[c] * du[n];
      }
    }
    block2(sq, dot, o1, o2);
    const float nf = sqrtf(o1), ne = nf + 1e-9f;
    const float k1 = 1.0f / ne, k2 = nf > 0.0f ? o2 / (nf * ne * ne) : 0.0f;
    int n = 0;
    for (int c = tid; c < D; c += 256, ++n) de[((size_t)b * F + f) * D + c] = du[n] * k1 - row[c] * k2;
  }
}

// word-embedding gradient: dword[id] = sum of ds rows of the tokens carrying that id, in token order.  One wave per token: the
// FIRST occurrence of an id owns its row and adds every later occurrence; other waves leave.  (dword is zeroed by the caller.)
template <int NI>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const int64_t* __restrict__ ids, const float* __restrict__ ds, float* __restrict__ dword,
                                                             int M, int H, int vocab) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= M) return;
  auto clampid = [&](long long id) { return id < 0 ? 0LL : (id >= vocab ? (long long)vocab - 1 : id); };
  const long long id = clampid(ids[t]);
  for (int base = 0; base < t; base += 64) {
    const int j = base + lane;
    const bool m = j < t && clampid(ids[j < M ? j : M - 1]) == id;
    if (__ballot(m) != 0ULL) return;      // an earlier token owns this row (wave-uniform)
  }
  f32x4 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = ld4(ds + (size_t)t * H + 4 * lane + 256 * i);
  for (int base = t + 1; base < M; base += 64) {
    const int j = base + lane;
    const bool m = j < M && clampid(ids[j < M ? j : M - 1]) == id;
    unsigned long long bal = __ballot(m);
    while (bal) {
      const int k = __builtin_ctzll(bal);
      bal &= bal - 1;
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += ld4(ds + (size_t)(base + k) * H + 4 * lane + 256 * i);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(dword + (size_t)id * H + 4 * lane + 256 * i) = acc[i];
}

// out[t][:] = sum_n ds[n * T + t][:]  (position-embedding gradients: rows of the same position over the batch, in batch order)
__global__ __launch_bounds__(256) void position_sum_kernel(const float* __restrict__ ds, float* __restrict__ out, int N, int T, int H) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x * 4; c < H; c += 1024) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) a += ld4(ds + ((size_t)n * T + t) * H + c);
    *reinterpret_cast<f32x4*>(out + (size_t)t * H + c) = a;
  }
}
// out[:] = sum_t rows[t][:], t ascending
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ rows, float* __restrict__ out, int T, int H) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < H; c += gridDim.x * 256) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += rows[(size_t)t * H + c];
    out[c] = a;
  }
}
// dpe[n * P + p][:] = bf16(ds[n * (P + 1) + 1 + p][:])  (the patch rows of the assembled token gradient, as a GEMM operand)
__global__ __launch_bounds__(256) void patch_rows_kernel(const float* __restrict__ ds, __bf16* __restrict__ dpe, int N, int P, int H) {
  const size_t total4 = (size_t)N * P * H / 4;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t e = i * 4, row = e / H;
    const int c = (int)(e % H);
    const size_t n = row / P, p = row % P;
    const f32x4 v = ld4(ds + (n * (P + 1) + 1 + p) * H + c);
    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(dpe + e) = o;
  }
}

// out = act(x), bf16 -> bf16, 8 elements per thread (training forward: FFN1 keeps its pre-activations AND their activation; the
// same erf-GELU / quick-GELU as the GEMM epilogues)
__global__ __launch_bounds__(256) void act_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ out, size_t n8, int act) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (__bf16)(act == UFND_ACT_GELU ? gelu_fast_f((float)v[q]) : quick_gelu_fast_f((float)v[q]));
    *reinterpret_cast<bf16x8*>(out + i * 8) = o;
  }
}

#define NI_LAUNCH(H, KERNEL, GRID, STREAM, ...)                                                       \
  do {                                                                                                \
    if ((H) == 256) hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);         \
    else if ((H) == 512) hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);    \
    else if ((H) == 768) hipLaunchKernelGGL((KERNEL<3>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);                    \
  } while (0)

inline bool h_ok(int H) { return H == 256 || H == 512 || H == 768 || H == 1024; }
inline int ln_bwd_blocks(int M) { const int b = ufnd_cdiv(M, 8); return b < 1 ? 1 : (b > 256 ? 256 : b); }

}  // namespace

extern "C" int ufnd_act_bf16(const void* x, void* out, size_t n, int act, void* stream_) {
  UFND_REQUIRE(x && out && n >= 8 && n % 8 == 0 && ufnd_aligned(x, 16) && ufnd_aligned(out, 16), "act_bf16: n=%zu (multiple of 8), 16-B alignment", n);
  UFND_REQUIRE(act == UFND_ACT_GELU || act == UFND_ACT_QUICK_GELU, "act_bf16: act=%d", act);
  size_t want = (n / 8 + 255) / 256;
  hipLaunchKernelGGL(act_bf16_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream_, (const __bf16*)x, (__bf16*)out, n / 8, act);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" size_t ufnd_layernorm_bwd_workspace_floats(int M, int H) { return (size_t)ln_bwd_blocks(M) * 2 * (size_t)H; }

extern "C" int ufnd_layernorm_bwd(const float* x, int ldx, const float* gamma, const float* dy, int lddy, const float* add, int ldadd,
                                  float* dx_f32, void* dx_bf16, int lddx, float* dgamma, float* dbeta, float* workspace, int accumulate,
                                  int M, int H, float eps, void* stream_) {
  UFND_REQUIRE(x && gamma && dy && (dx_f32 || dx_bf16) && M >= 1, "layernorm_bwd: null argument");
  UFND_REQUIRE(h_ok(H), "layernorm_bwd: H=%d (supported 256/512/768/1024)", H);
  UFND_REQUIRE(ldx % 4 == 0 && ldx >= H && lddy % 4 == 0 && lddy >= H && lddx % 8 == 0 && lddx >= H && (!add || (ldadd % 4 == 0 && ldadd >= H)),
               "layernorm_bwd: strides");
  UFND_REQUIRE(ufnd_aligned(x, 16) && ufnd_aligned(gamma, 16) && ufnd_aligned(dy, 16) && (!add || ufnd_aligned(add, 16)) &&
                   (!dx_f32 || ufnd_aligned(dx_f32, 16)) && (!dx_bf16 || ufnd_aligned(dx_bf16, 8)), "layernorm_bwd: alignment");
  UFND_REQUIRE((!dgamma && !dbeta) || workspace, "layernorm_bwd: parameter gradients need the workspace (ufnd_layernorm_bwd_workspace_floats)");
  hipStream_t stream = (hipStream_t)stream_;
  const int nblk = ln_bwd_blocks(M);
  const bool defer = accumulate == UFND_PARTIALS_DEFER;
  UFND_REQUIRE(!defer || workspace, "layernorm_bwd: deferred parameter gradients need the workspace");
  float* part = (dgamma || dbeta || defer) ? workspace : nullptr;
  NI_LAUNCH(H, layernorm_bwd_kernel, dim3(nblk), stream, x, ldx, gamma, dy, lddy, add, ldadd, dx_f32, (__bf16*)dx_bf16, lddx, part, M, H, eps);
  UFND_CHECK_LAUNCH();
  if (part && !defer) {
    hipLaunchKernelGGL(row_partials_finish_kernel, dim3(ufnd_cdiv(H, 16), 2), dim3(256), 0, stream, part, nblk, H, dgamma, dbeta, accumulate);
    UFND_CHECK_LAUNCH();
  }
  return UFND_OK;
}

extern "C" int ufnd_layernorm_bwd_blocks(int M) { return M >= 1 ? ln_bwd_blocks(M) : 0; }

extern "C" int ufnd_row_partials_finish(const ufnd_partials_job* job, int accumulate, void* stream_) {
  UFND_REQUIRE(job && job->part && job->nblk >= 1 && job->H >= 16 && (job->out0 || job->out1) && (accumulate == 0 || accumulate == 1), "row_partials_finish: null argument");
  hipLaunchKernelGGL(row_partials_finish_kernel, dim3(ufnd_cdiv(job->H, 16), 2), dim3(256), 0, (hipStream_t)stream_, job->part, job->nblk, job->H, job->out0,
                     job->out1, accumulate);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_masked_meanpool_l2_bwd(const float* hidden, const int32_t* mask, const float* dfeat, float* dhidden, int B, int L, int H,
                                           void* stream_) {
  UFND_REQUIRE(hidden && mask && dfeat && dhidden && B >= 1 && L >= 1, "meanpool_bwd: null argument");
  UFND_REQUIRE(H >= 4 && H <= 1024 && H % 4 == 0, "meanpool_bwd: H=%d (a multiple of 4, <= 1024)", H);
  UFND_REQUIRE(ufnd_aligned(hidden, 16) && ufnd_aligned(dfeat, 16) && ufnd_aligned(dhidden, 16), "meanpool_bwd: 16-B alignment");
  hipLaunchKernelGGL(meanpool_l2_bwd_kernel, dim3(B, MP_SPLIT), dim3(1024), 0, (hipStream_t)stream_, hidden, mask, dfeat, dhidden, L, H);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_l2norm_frames_bwd(const float* e, const float* dfeat, float* de, int B, int F, int D, void* stream_) {
  UFND_REQUIRE(e && dfeat && de && B >= 1 && F >= 1 && D >= 1 && D <= 1024, "l2norm_frames_bwd: B=%d F=%d D=%d", B, F, D);
  hipLaunchKernelGGL(l2norm_frames_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, e, dfeat, de, F, D);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// BERT embeddings backward, given ds = the gradient of the summed embeddings (M = B L rows; ufnd_layernorm_bwd of the embedding
// LayerNorm): dword (vocab, H), dpos (max_pos, H), dtype (type_vocab, H) are OVERWRITTEN (rows nobody used: zero).
extern "C" int ufnd_bert_embed_bwd(const int64_t* ids, const float* ds, float* dword, float* dpos, float* dtype, int B, int L, int H, int vocab,
                                   int max_pos, int type_vocab, void* stream_) {
  UFND_REQUIRE(ids && ds && dword && dpos && dtype, "bert_embed_bwd: null argument");
  UFND_REQUIRE(h_ok(H) && B >= 1 && L >= 1 && L <= max_pos && vocab >= 1 && type_vocab >= 1, "bert_embed_bwd: B=%d L=%d H=%d", B, L, H);
  UFND_REQUIRE(ufnd_aligned(ds, 16) && ufnd_aligned(dword, 16) && ufnd_aligned(dpos, 16) && ufnd_aligned(dtype, 16), "bert_embed_bwd: alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const int M = B * L;
  if (hipMemsetAsync(dword, 0, (size_t)vocab * H * sizeof(float), stream) != hipSuccess ||
      hipMemsetAsync(dpos, 0, (size_t)max_pos * H * sizeof(float), stream) != hipSuccess ||
      hipMemsetAsync(dtype, 0, (size_t)type_vocab * H * sizeof(float), stream) != hipSuccess) {
    ufnd_set_error("bert_embed_bwd: memset failed");
    return UFND_ERR_LAUNCH;
  }
  NI_LAUNCH(H, embedding_grad_kernel, dim3(ufnd_cdiv(M, 4)), stream, ids, ds, dword, M, H, vocab);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(position_sum_kernel, dim3(L), dim3(256), 0, stream, ds, dpos, B, L, H);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(rows_sum_kernel, dim3(ufnd_cdiv(H, 256)), dim3(256), 0, stream, (const float*)dpos, dtype, L, H);     // token type 0 everywhere
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// ViT token assembly backward, given ds = the gradient of the assembled tokens before the pre-LayerNorm (N (P + 1) rows):
// dpos (P + 1, H) and dcls (H) overwritten, dpe (N P, H) bf16 = the patch rows (operand of the patch-embedding weight gradient).
extern "C" int ufnd_vit_assemble_bwd(const float* ds, float* dcls, float* dpos, void* dpe_bf16, int N, int P, int H, void* stream_) {
  UFND_REQUIRE(ds && dcls && dpos && dpe_bf16 && N >= 1 && P >= 1 && H % 4 == 0, "vit_assemble_bwd: null argument");
  UFND_REQUIRE(ufnd_aligned(ds, 16) && ufnd_aligned(dpos, 16) && ufnd_aligned(dpe_bf16, 8), "vit_assemble_bwd: alignment");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(position_sum_kernel, dim3(P + 1), dim3(256), 0, stream, ds, dpos, N, P + 1, H);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(rows_sum_kernel, dim3(ufnd_cdiv(H, 256)), dim3(256), 0, stream, (const float*)dpos, dcls, 1, H);       // x[n][0] = cls + pos[0]
  UFND_CHECK_LAUNCH();
  size_t want = ((size_t)N * P * H / 4 + 255) / 256;
  hipLaunchKernelGGL(patch_rows_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, stream, ds, (__bf16*)dpe_bf16, N, P, H);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
