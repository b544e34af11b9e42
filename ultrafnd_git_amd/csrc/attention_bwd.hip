// Multi-head self-attention, backward (Tier-B backward, SURVEY.md 8b `attention_bwd`; the reference keeps its encoders frozen,
// src/core_blocks/text_blocks.py:52,63 -- this serves TrainConfig.train_encoders).  Flash-style: nothing of size L x L exists;
// P is recomputed from Q, K and the forward's per-query log-sum-exp (ufnd_attention_bf16_lse).
//
//   S = Q K^T / 8 (+ key mask),  P = softmax(S),  O = P V          (head_dim 64)
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),  delta_i = sum_d dO_id O_id,  dQ = dS K / 8,  dK = dS^T Q / 8
//
// Two launches over (128-row block, head, batch), each a loop over 64-row blocks of the OTHER side, built from the two MFMA
// product forms of attention.hip (v_mfma_f32_16x16x32_bf16; C/D: column = lane & 15, row = 4 (lane >> 4) + r):
//   form 1   C[x][y] = X Y^T          X rows from an LDS image (ds_read_b128, swizzled), Y rows from registers: both operands are
//                                      row-major with the contraction (d = 64) contiguous;
//   form 2   C[d][y] = Z^T W          W = a form-1 accumulator, rounded to bf16 IN PLACE (its layout is the B-operand layout once
//                                      the contraction index is ordered (rows 4g..4g+3 of tile 2s, rows 4g..4g+3 of tile 2s+1)),
//                                      Z^T fragments in that same order from a row-major LDS image by ds_read_b64_tr_b16.
//   pass 1 (dQ: a workgroup owns 128 queries, walks the keys)    S^T = K Q^T, dP^T = V dO^T (form 1);  dQ^T += K^T dS^T (form 2)
//   pass 2 (dK, dV: a workgroup owns 128 keys, walks the queries) S = Q K^T, dP = dO V^T (form 1);  dV^T += dO^T P, dK^T += Q^T dS (form 2)
// S and dP are computed twice (7 products instead of 5): no gradient is summed across workgroups -- no atomics, no second
// pass, bitwise reproducible -- and attention is a few percent of an encoder layer's FLOPs at these lengths.
// Masked keys (HF semantics, as the forward): score = the finfo.min-like constant, so P = 0 for them wherever a row has a
// live key; keys / queries beyond L contribute exactly zero.
#include "common.hpp"

namespace {

constexpr float NEG_MASK = -3.0e38f;

__device__ __forceinline__ bf16x8 a_frag(const char* img, int row, int chunk) {      // form-1 image: chunk ^ ((row >> 1) & 7)
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// form-2 fragment: Z^T[d = 16 dt + fr][k-step ksd] from a row-major image with the transposed-read swizzle (chunk ^ (((row >> 1) & 3) << 1))
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int ksd, int dt, int fr, int g) {
  const int qq = fr >> 2, pp = fr & 3;
  const int r0 = 32 * ksd + 4 * g + qq, r1 = r0 + 16;
  const int ch = 2 * dt + (pp >> 1);
  const int off0 = r0 * 128 + ((ch ^ (((r0 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
  const int off1 = r1 * 128 + ((ch ^ (((r1 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
  union { s16x4 s2[2]; bf16x8 v; } u;
  u.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + off0));
  u.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + off1));
  return u.v;
}

// 64 rows x 64 columns (bf16) of a row-major matrix -> one or two LDS images (form-1 swizzle / form-2 swizzle); rows beyond
// `rows_valid` are clamped to the last valid row (their contribution is zeroed by the caller through the bias / lse words)
__device__ __forceinline__ void stage64(const __bf16* src, size_t ld, int row0, int rows_valid, char* img1, char* img2, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + 256 * it, r = idx >> 3, c = idx & 7;
    const int rr = (row0 + r) < rows_valid ? (row0 + r) : rows_valid - 1;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * ld + c * 8);
    if (img1) *reinterpret_cast<bf16x8*>(img1 + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
    if (img2) *reinterpret_cast<bf16x8*>(img2 + r * 128 + ((c ^ (((r >> 1) & 3) << 1)) << 4)) = v;
  }
}

// delta[token][head] = sum_d dO O  (one wave per token, 12 heads x 64 = 768 columns: lane pairs ... generic: loop)
__global__ __launch_bounds__(256) void attn_delta_kernel(const __bf16* dctx, const __bf16* ctx, float* delta, int tokens, int heads) {
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tok >= tokens) return;
  const int H = heads * 64;
  // 8 lanes per head (8 columns each): 8 heads per wave-pass
  for (int h0 = 0; h0 < heads; h0 += 8) {
    const int h = h0 + (lane >> 3), c = (lane & 7) * 8;
    float s = 0.0f;
    if (h < heads) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(dctx + (size_t)tok * H + h * 64 + c);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(ctx + (size_t)tok * H + h * 64 + c);
#pragma unroll
      for (int q = 0; q < 8; ++q) s += (float)a[q] * (float)b[q];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (h < heads && (lane & 7) == 0) delta[(size_t)tok * heads + h] = s;
  }
}

template <int PASS>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const __bf16* qkv, const __bf16* dctx, const float* lse, const float* delta,
                                                            const int32_t* mask, __bf16* dqkv, int L, int heads, float scale_log2e, float scale, int nob) {
  // images of the walked side's current 64-row block
  __shared__ __attribute__((aligned(16))) char imgA[64 * 128];      // pass 1: K (form 1)      pass 2: Q  (form 1)
  __shared__ __attribute__((aligned(16))) char imgB[64 * 128];      // pass 1: V (form 1)      pass 2: dO (form 1)
  __shared__ __attribute__((aligned(16))) char imgC[64 * 128];      // pass 1: K (form 2)      pass 2: Q  (form 2)
  __shared__ __attribute__((aligned(16))) char imgD[64 * 128];      //                          pass 2: dO (form 2)
  __shared__ __attribute__((aligned(16))) float w0[64];             // pass 1: key bias         pass 2: lse of the block's queries
  __shared__ __attribute__((aligned(16))) float w1[64];             //                          pass 2: delta of the block's queries

  // (owned block, head, sample), owned block fastest, XCD-contiguous: the blocks of one (sample, head) walk the same rows
  const int lid = xcd_contiguous_id((int)blockIdx.x, (int)gridDim.x);
  const int ob = lid % nob, h = (lid / nob) % heads, b = lid / (nob * heads);
  const int H = heads * 64, ld = 3 * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, g = lane >> 4;
  const size_t tok0 = (size_t)b * L;

  // ---- the owned side: two column tiles of 16 rows per wave, held in registers as B operands
  //      pass 1: Q and dO rows of my queries; pass 2: K and V rows of my keys
  bf16x8 own0[2][2], own1[2][2];
  int orow[2];
  float c0[2], c1[2];      // pass 1: lse, delta of my queries; pass 2: key bias of my keys
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = ob * 128 + wave * 32 + t * 16 + fr;
    orow[t] = r;
    const int rc = r < L ? r : L - 1;
    if constexpr (PASS == 1) {
      const __bf16* qs = qkv + (tok0 + rc) * ld + h * 64 + 8 * g;
      const __bf16* ds = dctx + (tok0 + rc) * H + h * 64 + 8 * g;
      own0[t][0] = *reinterpret_cast<const bf16x8*>(qs);
      own0[t][1] = *reinterpret_cast<const bf16x8*>(qs + 32);
      own1[t][0] = *reinterpret_cast<const bf16x8*>(ds);
      own1[t][1] = *reinterpret_cast<const bf16x8*>(ds + 32);
      c0[t] = r < L ? lse[(tok0 + rc) * heads + h] : INFINITY;      // (a query beyond L: P = exp2(s - inf) = 0)
      c1[t] = r < L ? delta[(tok0 + rc) * heads + h] : 0.0f;
    } else {
      const __bf16* ks = qkv + (tok0 + rc) * ld + H + h * 64 + 8 * g;
      own0[t][0] = *reinterpret_cast<const bf16x8*>(ks);
      own0[t][1] = *reinterpret_cast<const bf16x8*>(ks + 32);
      own1[t][0] = *reinterpret_cast<const bf16x8*>(ks + H);
      own1[t][1] = *reinterpret_cast<const bf16x8*>(ks + H + 32);
      c0[t] = r < L ? ((!mask || mask[tok0 + rc] != 0) ? 0.0f : NEG_MASK) : -INFINITY;
      c1[t] = 0.0f;
    }
  }

  f32x4 acc0[4][2], acc1[4][2];      // pass 1: dQ^T (acc0); pass 2: dK^T (acc0), dV^T (acc1)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc0[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int wb0 = 0; wb0 < L; wb0 += 64) {
    __syncthreads();      // the previous block's LDS reads are done
    if constexpr (PASS == 1) {
      stage64(qkv + tok0 * ld + H + h * 64, ld, wb0, L, imgA, imgC, tid);           // K: both forms
      stage64(qkv + tok0 * ld + 2 * H + h * 64, ld, wb0, L, imgB, nullptr, tid);   // V: form 1
      if (tid < 64) {
        const int key = wb0 + tid;
        w0[tid] = key < L ? ((!mask || mask[tok0 + key] != 0) ? 0.0f : NEG_MASK) : -INFINITY;
      }
    } else {
      stage64(qkv + tok0 * ld + h * 64, ld, wb0, L, imgA, imgC, tid);              // Q: both forms
      stage64(dctx + tok0 * H + h * 64, H, wb0, L, imgB, imgD, tid);               // dO: both forms
      if (tid < 64) {
        const int q = wb0 + tid;
        w0[tid] = q < L ? lse[(tok0 + q) * heads + h] : INFINITY;
        w1[tid] = q < L ? delta[(tok0 + q) * heads + h] : 0.0f;
      }
    }
    __syncthreads();

    // ---- form 1: s = (walked rows) x (owned rows)^T, dp likewise: 4 row tiles x 2 column tiles, d = 64 in two k-halves
    f32x4 s[4][2], dp[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t) { s[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const bf16x8 fa = a_frag(imgA, rt * 16 + fr, g + 4 * kk);
        const bf16x8 fb = a_frag(imgB, rt * 16 + fr, g + 4 * kk);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (PASS == 1) {       // S^T = K Q^T ; dP^T = V dO^T
            s[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, own0[t][kk], s[rt][t], 0, 0, 0);
            dp[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, own1[t][kk], dp[rt][t], 0, 0, 0);
          } else {                         // S = Q K^T ; dP = dO V^T
            s[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, own0[t][kk], s[rt][t], 0, 0, 0);
            dp[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, own1[t][kk], dp[rt][t], 0, 0, 0);
          }
        }
      }

    // ---- P = exp2(score - lse), dS = P (dP - delta) / 8, rounded to bf16 into the form-2 B-operand order
    bf16x8 pf[2][2], dsf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const f32x4 wv0 = *reinterpret_cast<const f32x4*>(w0 + rt * 16 + 4 * g);
        f32x4 wv1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PASS == 2) wv1 = *reinterpret_cast<const f32x4*>(w1 + rt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc2, ls, dl;
          if constexpr (PASS == 1) {       // rows = keys (bias by row), columns = my queries (lse, delta by lane)
            sc2 = (wv0[r] == 0.0f) ? s[rt][t][r] * scale_log2e : wv0[r];
            ls = c0[t];
            dl = c1[t];
          } else {                         // rows = queries (lse, delta by row), columns = my keys (bias by lane)
            sc2 = (c0[t] == 0.0f) ? s[rt][t][r] * scale_log2e : c0[t];
            ls = wv0[r];
            dl = wv1[r];
          }
          const float p = fast_exp2(sc2 - ls);
          const float dsv = p * (dp[rt][t][r] - dl) * scale;
          pf[rt >> 1][t][(rt & 1) * 4 + r] = (__bf16)p;
          dsf[rt >> 1][t][(rt & 1) * 4 + r] = (__bf16)dsv;
        }
      }

    // ---- form 2: contraction over the block's 64 walked rows in two steps of 32
#pragma unroll
    for (int ksd = 0; ksd < 2; ++ksd)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if constexpr (PASS == 1) {         // dQ^T += K^T dS^T
          const bf16x8 zt = tr_frag(imgC, ksd, dt, fr, g);
#pragma unroll
          for (int t = 0; t < 2; ++t) acc0[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zt, dsf[ksd][t], acc0[dt][t], 0, 0, 0);
        } else {                           // dK^T += Q^T dS ; dV^T += dO^T P
          const bf16x8 zq = tr_frag(imgC, ksd, dt, fr, g);
          const bf16x8 zo = tr_frag(imgD, ksd, dt, fr, g);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc0[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zq, dsf[ksd][t], acc0[dt][t], 0, 0, 0);
            acc1[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zo, pf[ksd][t], acc1[dt][t], 0, 0, 0);
          }
        }
      }
  }

  // ---- store: lane holds X^T[d = 16 dt + 4 g + r][owned row fr]
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (orow[t] >= L) continue;
    __bf16* base = dqkv + (tok0 + orow[t]) * ld + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if constexpr (PASS == 1) {
        bf16x4 v = {(__bf16)acc0[dt][t][0], (__bf16)acc0[dt][t][1], (__bf16)acc0[dt][t][2], (__bf16)acc0[dt][t][3]};
        *reinterpret_cast<bf16x4*>(base + dt * 16) = v;
      } else {
        bf16x4 vk = {(__bf16)acc0[dt][t][0], (__bf16)acc0[dt][t][1], (__bf16)acc0[dt][t][2], (__bf16)acc0[dt][t][3]};
        bf16x4 vv = {(__bf16)acc1[dt][t][0], (__bf16)acc1[dt][t][1], (__bf16)acc1[dt][t][2], (__bf16)acc1[dt][t][3]};
        *reinterpret_cast<bf16x4*>(base + H + dt * 16) = vk;
        *reinterpret_cast<bf16x4*>(base + 2 * H + dt * 16) = vv;
      }
    }
  }
}

}  // namespace

extern "C" size_t ufnd_attention_bwd_workspace_floats(int B, int L, int heads) { return (size_t)B * L * heads; }

extern "C" int ufnd_attention_bf16_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, const int32_t* key_mask,
                                       void* dqkv, float* workspace, int B, int L, int heads, void* stream_) {
  UFND_REQUIRE(qkv && ctx && dctx && lse && dqkv && workspace, "attention_bwd: null operand");
  UFND_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && L <= 4096 && heads >= 1 && heads <= 64, "attention_bwd: B=%d L=%d heads=%d", B, L, heads);
  UFND_REQUIRE(ufnd_aligned(qkv, 16) && ufnd_aligned(ctx, 16) && ufnd_aligned(dctx, 16) && ufnd_aligned(dqkv, 16), "attention_bwd: 16-B alignment required");
  hipStream_t stream = (hipStream_t)stream_;
  const float scale = 0.125f, scale_log2e = 0.125f * 1.44269504088896340736f;
  const int tokens = B * L;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(ufnd_cdiv(tokens, 4)), dim3(256), 0, stream, (const __bf16*)dctx, (const __bf16*)ctx, workspace, tokens, heads);
  UFND_CHECK_LAUNCH();
  const int nob = ufnd_cdiv(L, 128);
  UFND_REQUIRE((long long)nob * heads * B < (1ll << 31), "attention_bwd: grid too large");
  const dim3 grid(nob * heads * B);
  hipLaunchKernelGGL(attention_bwd_kernel<1>, grid, dim3(256), 0, stream, (const __bf16*)qkv, (const __bf16*)dctx, lse, (const float*)workspace, key_mask,
                     (__bf16*)dqkv, L, heads, scale_log2e, scale, nob);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(attention_bwd_kernel<2>, grid, dim3(256), 0, stream, (const __bf16*)qkv, (const __bf16*)dctx, lse, (const float*)workspace, key_mask,
                     (__bf16*)dqkv, L, heads, scale_log2e, scale, nob);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
