// fp32 skinny GEMMs on v_mfma_f32_32x32x2_f32 (see gemm_f32.hpp).
//
// MFMA 32x32x2 f32 operand maps (guide section 3): lane l, i = j = l & 31, h = l >> 5
//   A[i][k=h], B[k=h][j];  C/D: column j = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h, r in [0,16).
// The contraction index may be permuted freely as long as A and B agree, so each lane loads
// FOUR consecutive contraction elements (16 B) and feeds element e to MFMA step e: half h of
// the wave covers k = kb + 4h + e of every 8-wide chunk.  The free index of the streamed
// weight matrix is likewise permuted so that a lane owns VEC consecutive output columns.
#include "gemm_f32.hpp"

namespace {

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&o)[4]) {
  if constexpr (VEC == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  } else {
    const f32x2 a = *reinterpret_cast<const f32x2*>(p);
    const f32x2 b = *reinterpret_cast<const f32x2*>(p + 2);
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  return z;
}

// ------------------------------------------------------------------------------------------
// NT: Y = act(X W^T + b).   block = 4 waves splitting the block's K range; LDS tree-free sum.
// ------------------------------------------------------------------------------------------
struct NtArgs {
  NtProb p[UFND_GEMM_MAX_PROB];
  int begin[UFND_GEMM_MAX_PROB + 1];
  int nprob;
  const ufnd_step_state* st;
};

template <int MT, int WVEC>
__global__ __launch_bounds__(256) void nt_kernel(const NtArgs args) {
  __shared__ float red[4][MT][16][64];
  int pi = 0;
  for (int q = 1; q < args.nprob; ++q)
    if ((int)blockIdx.x >= args.begin[q]) pi = q;
  const NtProb& P = args.p[pi];
  int local = blockIdx.x - args.begin[pi];
  const int ks = local % P.ksplit;
  local /= P.ksplit;
  const int n_tiles = P.N >> 5;
  const int n0 = (local % n_tiles) << 5;
  const int m0 = (local / n_tiles) * (32 * MT);

  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  // K range of this block, then of this wave (multiples of 8 except at the very end of K)
  int kblk = (P.K + P.ksplit - 1) / P.ksplit;
  kblk = (kblk + 31) & ~31;
  const int kb0 = ks * kblk;
  const int kb1 = min(P.K, kb0 + kblk);
  const int sub = kblk >> 2;
  const int k0 = kb0 + w * sub;
  const int k1 = min(kb1, k0 + sub);

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = zero16();

  const float* wrow = P.W + (size_t)(n0 + j) * P.ldw;
  const float* xrow[MT];
  bool xok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + 32 * t + j;
    xok[t] = m < P.M;
    xrow[t] = P.X + (size_t)(xok[t] ? m : 0) * P.ldx;
  }

  // Batches of U 8-wide chunks: every load of a batch is issued before its first MFMA, so a wave
  // keeps U x (1 + MT) 16-B loads in flight (the loop is latency-bound, not bandwidth-bound).
  constexpr int U = 4;      // (8 chunks per batch measured in round 4: no change, 0.2337 vs 0.2334 ms per head step)
  int k = k0;
  for (; k + 8 * U <= k1; k += 8 * U) {
    float a[U][4], b[U][MT][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k + 8 * u + 4 * h;
      load_vec<WVEC>(wrow + kk, a[u]);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (xok[t]) {
          load_vec<4>(xrow[t] + kk, b[u][t]);
        } else {
          b[u][t][0] = b[u][t][1] = b[u][t][2] = b[u][t][3] = 0.0f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][t][e], acc[t], 0, 0, 0);
  }
  for (; k + 8 <= k1; k += 8) {
    const int kk = k + 4 * h;
    float a[4];
    load_vec<WVEC>(wrow + kk, a);
    float b[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (xok[t]) {
        load_vec<4>(xrow[t] + kk, b[t]);
      } else {
        b[t][0] = b[t][1] = b[t][2] = b[t][3] = 0.0f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[t][e], acc[t], 0, 0, 0);
  }
  if (k < k1) {  // ragged tail (K not a multiple of 8), element-guarded
    const int kk = k + 4 * h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = (kk + e) < k1;
      const float a = ok ? wrow[kk + e] : 0.0f;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const float b = (ok && xok[t]) ? xrow[t][kk + e] : 0.0f;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][t][r][lane] = acc[t][r];
  __syncthreads();

  // wave w finalises accumulator registers 4w..4w+3: rows n = n0 + 8w + 4h + q, column m
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += red[ww][t][4 * w + q][lane];
    const int m = m0 + 32 * t + j;
    const int n = n0 + 8 * w + 4 * h;
    if (m >= P.M) continue;
    if (P.ksplit > 1) {
      *reinterpret_cast<f32x4*>(P.Y + ((size_t)ks * P.M + m) * P.N + n) = v;
      continue;
    }
    if (P.bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(P.bias + n);
      v += bv;
    }
    if (P.Z) *reinterpret_cast<f32x4*>(P.Z + (size_t)m * P.ldz + n) = v;
    if (P.act == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
    }
    if (P.drop_p > 0.0f) {
      float dm[4];      // (n is a multiple of 4 and N of 32: four aligned elements, one Philox evaluation)
      dropout_mul4(args.st, P.drop_p, P.drop_layer, (uint32_t)(m * P.N + n), dm);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] *= dm[q];
    }
    *reinterpret_cast<f32x4*>(P.Y + (size_t)m * P.ldy + n) = v;
  }
}

// ------------------------------------------------------------------------------------------
// NT on 16x16 tiles (v_mfma_f32_16x16x4_f32), for launches that are too small to fill the chip with 32x32 tiles (round 3).
// The fp32 MFMA rate is per SIMD, so a 32-row batch against N = 512 outputs -- 16 tiles of 32x32 -- runs its whole contraction
// on 16 CUs: 128 dependent MFMAs per wave at K = 1,024.  With 16x16 tiles the same launch has 64 workgroups and a quarter of
// the chain per wave.  Operand maps: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; C/D: column j = l & 15, row 4 (l >> 4) + r.
// A lane loads four consecutive contraction elements (16 B) and feeds element e to MFMA step e (a chunk = 16 contraction
// elements = 4 MFMAs); rows n of W are the A operand, batch rows m the B operand, so a lane ends with four consecutive n of one m.
// The four waves split the K range; their partial tiles are added in wave order in LDS and wave 0 runs the epilogue.
// ------------------------------------------------------------------------------------------
template <int WVEC>
__global__ __launch_bounds__(256) void nt16_kernel(const NtArgs args) {
  __shared__ float red[4][4][64];
  int pi = 0;
  for (int q = 1; q < args.nprob; ++q)
    if ((int)blockIdx.x >= args.begin[q]) pi = q;
  const NtProb& P = args.p[pi];
  int local = blockIdx.x - args.begin[pi];
  const int n_tiles = P.N >> 4;
  const int n0 = (local % n_tiles) << 4;
  const int m0 = (local / n_tiles) << 4;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  // K range of this wave (multiples of 16 except at the very end of K); ksplit == 1 in this form
  int sub = (P.K + 3) / 4;
  sub = (sub + 15) & ~15;
  const int k0 = min(P.K, w * sub), k1 = min(P.K, k0 + sub);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* wrow = P.W + (size_t)(n0 + j) * P.ldw;
  const int m = m0 + j;
  const bool xok = m < P.M;
  const float* xrow = P.X + (size_t)(xok ? m : 0) * P.ldx;
  constexpr int U = 8;      // 8 chunks = 128 contraction elements per batch: 16 loads of 16 B in flight, then 32 MFMAs
  int k = k0;
  for (; k + 16 * U <= k1; k += 16 * U) {
    float a[U][4], b[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k + 16 * u + 4 * g;
      load_vec<WVEC>(wrow + kk, a[u]);
      if (xok) {
        load_vec<4>(xrow + kk, b[u]);
      } else {
        b[u][0] = b[u][1] = b[u][2] = b[u][3] = 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
  }
  for (; k + 16 <= k1; k += 16) {
    const int kk = k + 4 * g;
    float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
    load_vec<WVEC>(wrow + kk, a);
    if (xok) load_vec<4>(xrow + kk, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
  if (k < k1) {  // ragged tail (K not a multiple of 16), element-guarded
    const int kk = k + 4 * g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = (kk + e) < k1;
      const float a = ok ? wrow[kk + e] : 0.0f;
      const float b = (ok && xok) ? xrow[kk + e] : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][r][lane] = acc[r];
  __syncthreads();
  if (w != 0 || !xok) return;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
  const int n = n0 + 4 * g;      // this lane: outputs n .. n + 3 of batch row m
  if (P.bias) v += *reinterpret_cast<const f32x4*>(P.bias + n);
  if (P.Z) *reinterpret_cast<f32x4*>(P.Z + (size_t)m * P.ldz + n) = v;
  if (P.act == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
  }
  if (P.drop_p > 0.0f) {
    float dm[4];
    dropout_mul4(args.st, P.drop_p, P.drop_layer, (uint32_t)(m * P.N + n), dm);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] *= dm[q];
  }
  *reinterpret_cast<f32x4*>(P.Y + (size_t)m * P.ldy + n) = v;
}

// ------------------------------------------------------------------------------------------
// NN: dX = dY W  (contraction over the weight's ROW index n; weight rows stream coalesced).
// A wave owns a strip of 32*VEC output columns (lane j owns columns kc + VEC*j .. +VEC-1).
// ------------------------------------------------------------------------------------------
struct NnArgs {
  NnProb p[UFND_GEMM_MAX_PROB];
  int begin[UFND_GEMM_MAX_PROB + 1];
  int nprob;
  const ufnd_step_state* st;
};

template <int VEC>
__global__ __launch_bounds__(256) void nn_kernel(const NnArgs args) {
  __shared__ float red[4][VEC][16][64];
  int pi = 0;
  for (int q = 1; q < args.nprob; ++q)
    if ((int)blockIdx.x >= args.begin[q]) pi = q;
  const NnProb& P = args.p[pi];
  int local = blockIdx.x - args.begin[pi];
  const int ns = local % P.nsplit;
  local /= P.nsplit;
  const int strips = (P.K + 32 * VEC - 1) / (32 * VEC);
  const int kc = (local % strips) * 32 * VEC;
  const int m0 = (local / strips) * 32;

  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  int nblk = (P.N + P.nsplit - 1) / P.nsplit;
  nblk = (nblk + 31) & ~31;
  const int nb0 = ns * nblk;
  const int nb1 = min(P.N, nb0 + nblk);
  const int sub = nblk >> 2;
  const int n0 = nb0 + w * sub;
  const int n1 = min(nb1, n0 + sub);  // N % 32 == 0 (host-checked) => multiples of 8

  f32x16 acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = zero16();

  const int m = m0 + j;
  const bool mok = m < P.M;
  const float* dyrow = P.dY + (size_t)(mok ? m : 0) * P.lddy;
  const int kcol = kc + VEC * j;
  const bool kok = kcol < P.K;  // K % VEC == 0 (host-checked)
  const float* wcol = P.W + (kok ? kcol : 0);

  // batches of U chunks (8 weight rows each): all 5U loads of a batch in flight before its MFMAs
  constexpr int U = (VEC == 4) ? 4 : (VEC == 2 ? 6 : 8);
  for (int n = n0; n < n1; n += 8 * U) {
    float a[U][4], b[U][4][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int nn = n + 8 * u + 4 * h;
      const bool nok = (n + 8 * u) < n1;
      if (mok && nok) {
        load_vec<4>(dyrow + nn, a[u]);
      } else {
        a[u][0] = a[u][1] = a[u][2] = a[u][3] = 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* src = wcol + (size_t)(nn + e) * P.ldw;
        if (kok && nok) {
          if constexpr (VEC == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src);
            b[u][e][0] = t[0]; b[u][e][1] = t[1]; b[u][e][2] = t[2]; b[u][e][3] = t[3];
          } else if constexpr (VEC == 2) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(src);
            b[u][e][0] = t[0]; b[u][e][1] = t[1];
          } else {
            b[u][e][0] = *src;
          }
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) b[u][e][v] = 0.0f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e][v], acc[v], 0, 0, 0);
  }

#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][v][r][lane] = acc[v][r];
  __syncthreads();

  // wave w finalises registers 4w..4w+3: rows mrow = m0 + q + 8w + 4h, columns kc + VEC*j + v
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float val[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float s = 0.0f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) s += red[ww][v][4 * w + q][lane];
      val[v] = s;
    }
    const int mrow = m0 + q + 8 * w + 4 * h;
    if (mrow >= P.M || !kok) continue;
    if (P.nsplit > 1) {
      float* dst = P.out + ((size_t)ns * P.M + mrow) * P.ldo + kcol;      // partials [nsplit][M][ldo] (the one caller has ldo == K or the consumer's row stride)
#pragma unroll
      for (int v = 0; v < VEC; ++v) dst[v] = val[v];
      continue;
    }
    if (P.actZ) {
      const float* z = P.actZ + (size_t)mrow * P.ldz + kcol;
      float dm[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if constexpr (VEC == 4) {      // four aligned elements (kcol and drop_ld are multiples of 4 in this form): one Philox evaluation
        if ((P.drop_ld & 3) == 0) dropout_mul4(args.st, P.drop_p, P.drop_layer, (uint32_t)(mrow * P.drop_ld + kcol), dm);
        else
#pragma unroll
          for (int v = 0; v < 4; ++v) dm[v] = dropout_mul(args.st, P.drop_p, P.drop_layer, (uint32_t)(mrow * P.drop_ld + kcol + v));
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dm[v] = dropout_mul(args.st, P.drop_p, P.drop_layer, (uint32_t)(mrow * P.drop_ld + kcol + v));
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) val[v] = val[v] * gelu_grad_f(z[v]) * dm[v];      // (d g) m: the order of act_bwd_kernel and of autograd (the head's fused backward swaps one for the other)
    }
    if (P.add) {
      const float* ad = P.add + (size_t)mrow * P.ldadd + kcol;
#pragma unroll
      for (int v = 0; v < VEC; ++v) val[v] += ad[v];
    }
    float* dst = P.out + (size_t)mrow * P.ldo + kcol;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = val[v];
  }
}

// ------------------------------------------------------------------------------------------
// NN on 16x16 tiles (v_mfma_f32_16x16x4_f32) for narrow layers (see nt16_kernel): tile = 16 batch rows x 16 output columns.
// A = dY rows (lane (i, g) loads dY[m0 + i][n + 4 g .. + 3] as 16 B), B = W[n + 4 g + e][kcol0 + j] (four 4-B loads, 16 lanes
// contiguous); C/D: column kcol0 + (l & 15), rows m0 + 4 (l >> 4) + r.  The four waves split the contraction range, partial
// tiles are added in wave order in LDS, wave 0 runs the epilogue.  nsplit == 1 only.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn16_kernel(const NnArgs args) {
  __shared__ float red[4][4][64];
  int pi = 0;
  for (int q = 1; q < args.nprob; ++q)
    if ((int)blockIdx.x >= args.begin[q]) pi = q;
  const NnProb& P = args.p[pi];
  const int local = blockIdx.x - args.begin[pi];
  const int strips = (P.K + 15) >> 4;
  const int kc = (local % strips) << 4;
  const int m0 = (local / strips) << 4;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  int sub = (P.N + 3) / 4;
  sub = (sub + 15) & ~15;
  const int n0 = min(P.N, w * sub), n1 = min(P.N, n0 + sub);      // N % 32 == 0 (host-checked) => multiples of 16
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int m = m0 + j;
  const bool mok = m < P.M;
  const float* dyrow = P.dY + (size_t)(mok ? m : 0) * P.lddy;
  const int kcol = kc + j;
  const bool kok = kcol < P.K;
  const float* wcol = P.W + (kok ? kcol : 0);
  constexpr int U = 4;      // 4 chunks = 64 contraction rows per batch: 4 + 16 loads in flight, then 16 MFMAs
  for (int n = n0; n < n1; n += 16 * U) {
    float a[U][4], b[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int nn = n + 16 * u + 4 * g;
      const bool nok = (n + 16 * u) < n1;
      if (mok && nok) {
        load_vec<4>(dyrow + nn, a[u]);
      } else {
        a[u][0] = a[u][1] = a[u][2] = a[u][3] = 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) b[u][e] = (kok && nok) ? wcol[(size_t)(nn + e) * P.ldw] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][r][lane] = acc[r];
  __syncthreads();
  if (w != 0 || !kok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mrow = m0 + 4 * g + r;
    if (mrow >= P.M) continue;
    float val = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    if (P.actZ) {
      val = val * gelu_grad_f(P.actZ[(size_t)mrow * P.ldz + kcol]) * dropout_mul(args.st, P.drop_p, P.drop_layer, (uint32_t)(mrow * P.drop_ld + kcol));
    }
    if (P.add) val += P.add[(size_t)mrow * P.ldadd + kcol];
    P.out[(size_t)mrow * P.ldo + kcol] = val;
  }
}

// ------------------------------------------------------------------------------------------
// TN: dW = dY^T X (contraction over the batch rows), db = column sums of dY.
// One wave per (32 weight rows) x (32*VEC weight columns) tile; no LDS, no barrier.
// ------------------------------------------------------------------------------------------
struct TnArgs {
  TnProb p[UFND_GEMM_MAX_PROB];
  int begin[UFND_GEMM_MAX_PROB + 1];  // in wave-tiles
  int vec[UFND_GEMM_MAX_PROB];        // per-problem vector width (tn_kernel<-1>)
  int nprob;
};

// MSPLIT = 0: one wave per tile (4 tiles per workgroup), the whole batch in one wave's chain -- small batches.
// MSPLIT = 1 (M >= 128): one WORKGROUP per tile, wave w takes the w-th quarter of the batch rows (rounded to 8), the four
// partial tiles meet in LDS and are added in wave order; wave w stores accumulator registers 4w..4w+3.  At B = 256 the
// one-wave form is a chain of 8 load-then-MFMA passes with about one wave per SIMD to hide them behind (139 us for the
// grouped launch, 72 us for the classifier's 272 tiles); this form has 4 x the waves and a quarter of the chain.
// SEG = 1: the batch rows come in segments (TnProb::seg_rows; the factor form of the data-parallel exchange) -- its own
// instantiation, so that the one-panel kernels keep their code.
template <int VEC, int MSPLIT, int SEG>
__device__ __forceinline__ void tn_tile(const TnProb& P, int local, int w, int lane) {
  const int j = lane & 31, h = lane >> 5;
  const int strips = (P.K + 32 * VEC - 1) / (32 * VEC);
  const int strip = local % strips;
  const int n0 = (local / strips) << 5;
  const int kcol = strip * 32 * VEC + VEC * j;
  const bool kok = kcol < P.K;

  f32x16 acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = zero16();
  float dbsum = 0.0f;
  const float* dycol = P.dY + n0 + j;
  const float* xcol = P.X + (kok ? kcol : 0);

  int m_lo = 0, m_hi = P.M;
  if constexpr (MSPLIT) {
    const int per = ((P.M + 3) / 4 + 7) & ~7;
    m_lo = w * per;
    m_hi = m_lo + per < P.M ? m_lo + per : P.M;
  }
  constexpr int U = 4;   // 32 batch rows per pass: every load of the pass is issued before its MFMAs
  for (int mb = m_lo; mb < m_hi; mb += 8 * U) {
    float a[U][4];
    float b[U][4][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mb + 8 * u + 4 * h + e;
        const bool ok = m < m_hi;
        size_t oy = (size_t)m * P.lddy, ox = (size_t)m * P.ldx;
        if constexpr (SEG) {
          const int sg = m / P.seg_rows, r = m - sg * P.seg_rows;
          oy = (size_t)sg * P.seg_dy + (size_t)r * P.lddy;
          ox = (size_t)sg * P.seg_x + (size_t)r * P.ldx;
        }
        a[u][e] = ok ? dycol[oy] : 0.0f;
        if (ok && kok) {
          const float* src = xcol + ox;
          if constexpr (VEC == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src);
            b[u][e][0] = t[0]; b[u][e][1] = t[1]; b[u][e][2] = t[2]; b[u][e][3] = t[3];
          } else {
            const f32x2 t = *reinterpret_cast<const f32x2*>(src);
            b[u][e][0] = t[0]; b[u][e][1] = t[1];
          }
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) b[u][e][v] = 0.0f;
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dbsum += a[u][e];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e][v], acc[v], 0, 0, 0);
      }
  }

  if constexpr (MSPLIT) {
    __shared__ float red[4][VEC][16][64];
    __shared__ float dbr[4][64];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w][v][r][lane] = acc[v][r];
    dbr[w][lane] = dbsum;
    __syncthreads();
    // wave w finalises accumulator registers 4w..4w+3: weight rows n0 + q + 8w + 4h
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float val[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t = 0.0f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) t += red[ww][v][4 * w + q][lane];
        val[v] = t;
      }
      if (!kok) continue;
      float* dst = P.dW + (size_t)(n0 + q + 8 * w + 4 * h) * P.ldw + kcol;
      if constexpr (VEC == 4) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{val[0], val[1], val[2], val[3]};
      } else {
        *reinterpret_cast<f32x2*>(dst) = f32x2{val[0], val[1]};
      }
    }
    if (strip == 0 && P.db && w == 0) {
      float t = 0.0f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) t += dbr[ww][lane];
      t += __shfl_xor(t, 32, 64);
      if (h == 0) P.db[n0 + j] = t;
    }
    return;
  } else {
  if (kok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float* dst = P.dW + (size_t)n * P.ldw + kcol;
      if constexpr (VEC == 4) {
        f32x4 t = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *reinterpret_cast<f32x4*>(dst) = t;
      } else {
        f32x2 t = {acc[0][r], acc[1][r]};
        *reinterpret_cast<f32x2*>(dst) = t;
      }
    }
  }
  if (strip == 0 && P.db) {
    const float s = dbsum + __shfl_xor(dbsum, 32, 64);
    if (h == 0) P.db[n0 + j] = s;
  }
  }
}
// VEC < 0: every problem with its OWN vector width (TnArgs::vec: 4 where its operands allow 16-byte accesses, else 2) -- a
// (hidden + aux)-wide problem then no longer puts a whole grouped launch on 8-byte accesses (one-wave-per-tile form only).
template <int VEC, int MSPLIT = 0, int SEG = 0>
__global__ __launch_bounds__(256) void tn_kernel(const TnArgs args) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = MSPLIT ? (int)blockIdx.x : blockIdx.x * 4 + w;
  if (tile >= args.begin[args.nprob]) return;
  int pi = 0;
  for (int q = 1; q < args.nprob; ++q)
    if (tile >= args.begin[q]) pi = q;
  const TnProb& P = args.p[pi];
  const int local = tile - args.begin[pi];
  if constexpr (VEC > 0) {
    tn_tile<VEC, MSPLIT, SEG>(P, local, w, lane);
  } else {
    static_assert(VEC > 0 || MSPLIT == 0, "per-problem widths: one wave per tile");
    if (args.vec[pi] == 4) tn_tile<4, 0, SEG>(P, local, w, lane);      // (wave-uniform)
    else tn_tile<2, 0, SEG>(P, local, w, lane);
  }
}


}  // namespace

// ------------------------------------------------------------------------------------------
// host launchers (argument checks guard every assumption the kernels make)
// ------------------------------------------------------------------------------------------
int launch_nt(const NtProb* probs, int nprob, const ufnd_step_state* st, hipStream_t stream) {
  UFND_REQUIRE(nprob >= 1 && nprob <= UFND_GEMM_MAX_PROB, "nt: %d problems", nprob);
  NtArgs a;
  a.nprob = nprob;
  a.st = st;
  int total = 0, maxM = 0;
  bool vec4 = true;
  for (int i = 0; i < nprob; ++i) {
    const NtProb& p = probs[i];
    UFND_REQUIRE(p.X && p.W && p.Y && p.M > 0 && p.K > 0, "nt[%d]: null/empty operand", i);
    UFND_REQUIRE(p.N % 32 == 0, "nt[%d]: N=%d not a multiple of 32", i, p.N);
    UFND_REQUIRE(p.ldx % 4 == 0 && ufnd_aligned(p.X, 16), "nt[%d]: X must be 16-B aligned with ldx%%4==0", i);
    UFND_REQUIRE(p.ldw % 2 == 0 && ufnd_aligned(p.W, 8), "nt[%d]: W must be 8-B aligned with even ldw", i);
    UFND_REQUIRE(p.ksplit >= 1 && p.ksplit <= 64, "nt[%d]: ksplit=%d", i, p.ksplit);
    UFND_REQUIRE(ufnd_aligned(p.Y, 16) && (p.ksplit > 1 || p.ldy % 4 == 0), "nt[%d]: Y alignment", i);
    UFND_REQUIRE(!p.Z || (ufnd_aligned(p.Z, 16) && p.ldz % 4 == 0), "nt[%d]: Z alignment", i);
    UFND_REQUIRE(!p.bias || ufnd_aligned(p.bias, 16), "nt[%d]: bias alignment", i);
    UFND_REQUIRE((long long)p.M * p.N < (1ll << 31), "nt[%d]: M*N too large", i);
    if (!(p.ldw % 4 == 0 && ufnd_aligned(p.W, 16))) vec4 = false;
    a.p[i] = p;
    maxM = p.M > maxM ? p.M : maxM;
  }
  {      // narrow layers: 16x16 tiles (four times the workgroups, a quarter of the MFMA chain per wave).  The choice looks at the
         // layers' widths only, never at the batch: a row's arithmetic must not depend on the batch it is computed in
    int t32 = 0;
    bool plain = true;
    for (int i = 0; i < nprob; ++i) {
      t32 += a.p[i].N / 32;
      if (a.p[i].ksplit != 1) plain = false;
    }
    if (plain && t32 <= 128) {
      for (int i = 0; i < nprob; ++i) {
        a.begin[i] = total;
        total += (a.p[i].N / 16) * ufnd_cdiv(a.p[i].M, 16);
      }
      a.begin[nprob] = total;
      if (vec4) hipLaunchKernelGGL((nt16_kernel<4>), dim3(total), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((nt16_kernel<2>), dim3(total), dim3(256), 0, stream, a);
      UFND_CHECK_LAUNCH();
      return UFND_OK;
    }
  }
  // two 32-row tiles per workgroup (the weight tile is read once for 64 rows) only when the launch still fills the chip that way:
  // the fp32 MFMA rate is per SIMD, so a launch of a few dozen workgroups is matrix-bound on the CUs it occupies
  int blocks1 = 0;
  for (int i = 0; i < nprob; ++i) blocks1 += (a.p[i].N / 32) * ufnd_cdiv(a.p[i].M, 32) * a.p[i].ksplit;
  const int MT = (maxM > 32 && blocks1 >= 512) ? 2 : 1;
  for (int i = 0; i < nprob; ++i) {
    a.begin[i] = total;
    total += (a.p[i].N / 32) * ufnd_cdiv(a.p[i].M, 32 * MT) * a.p[i].ksplit;
  }
  a.begin[nprob] = total;
  if (MT == 1) {
    if (vec4) hipLaunchKernelGGL((nt_kernel<1, 4>), dim3(total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((nt_kernel<1, 2>), dim3(total), dim3(256), 0, stream, a);
  } else {
    if (vec4) hipLaunchKernelGGL((nt_kernel<2, 4>), dim3(total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((nt_kernel<2, 2>), dim3(total), dim3(256), 0, stream, a);
  }
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

int launch_nn(const NnProb* probs, int nprob, const ufnd_step_state* st, hipStream_t stream) {
  UFND_REQUIRE(nprob >= 1 && nprob <= UFND_GEMM_MAX_PROB, "nn: %d problems", nprob);
  NnArgs a;
  a.nprob = nprob;
  a.st = st;
  bool vec4 = true;
  for (int i = 0; i < nprob; ++i) {
    const NnProb& p = probs[i];
    UFND_REQUIRE(p.dY && p.W && p.out && p.M > 0 && p.K > 0, "nn[%d]: null/empty operand", i);
    UFND_REQUIRE(p.N % 32 == 0, "nn[%d]: N=%d not a multiple of 32", i, p.N);
    UFND_REQUIRE(p.lddy % 4 == 0 && ufnd_aligned(p.dY, 16), "nn[%d]: dY alignment", i);
    UFND_REQUIRE(p.K % 2 == 0 && p.ldw % 2 == 0 && ufnd_aligned(p.W, 8), "nn[%d]: W alignment", i);
    UFND_REQUIRE(p.nsplit >= 1 && p.nsplit <= 64, "nn[%d]: nsplit=%d", i, p.nsplit);
    UFND_REQUIRE((long long)p.M * p.K < (1ll << 31), "nn[%d]: M*K too large", i);
    if (!(p.K % 4 == 0 && p.ldw % 4 == 0 && ufnd_aligned(p.W, 16))) vec4 = false;
    a.p[i] = p;
  }
  {      // narrow layers: 16x16 tiles (chosen by the layers' widths only, never by the batch: see launch_nt)
    int s32 = 0;
    bool plain = true;
    for (int i = 0; i < nprob; ++i) {
      s32 += ufnd_cdiv(a.p[i].K, 32);
      if (a.p[i].nsplit != 1) plain = false;
    }
    if (plain && s32 <= 128) {
      int total = 0;
      for (int i = 0; i < nprob; ++i) {
        a.begin[i] = total;
        total += ufnd_cdiv(a.p[i].K, 16) * ufnd_cdiv(a.p[i].M, 16);
      }
      a.begin[nprob] = total;
      hipLaunchKernelGGL(nn16_kernel, dim3(total), dim3(256), 0, stream, a);
      UFND_CHECK_LAUNCH();
      return UFND_OK;
    }
  }
  int VEC = vec4 ? 4 : 2;
  auto count = [&](int vec) {
    int t = 0;
    for (int i = 0; i < nprob; ++i) t += ufnd_cdiv(a.p[i].K, 32 * vec) * ufnd_cdiv(a.p[i].M, 32) * a.p[i].nsplit;
    return t;
  };
  // A wave's MFMA chain is (contraction / 32) * VEC long at 64 cycles each: when a launch has only a
  // handful of blocks, narrow the strips (more blocks, shorter chains) -- the work is latency-, not
  // bandwidth-bound at that size.
  // (the fp32 MFMA rate is per SIMD: with few workgroups the launch is matrix-bound on the CUs it occupies, so the strips narrow
  //  until the launch has about one workgroup per CU)
  if (count(VEC) < 64) VEC = 1;
  else if (count(VEC) < 256) VEC = (VEC == 4 && count(2) >= 256) ? 2 : 1;
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    a.begin[i] = total;
    total += ufnd_cdiv(a.p[i].K, 32 * VEC) * ufnd_cdiv(a.p[i].M, 32) * a.p[i].nsplit;
  }
  a.begin[nprob] = total;
  if (VEC == 4) hipLaunchKernelGGL((nn_kernel<4>), dim3(total), dim3(256), 0, stream, a);
  else if (VEC == 2) hipLaunchKernelGGL((nn_kernel<2>), dim3(total), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((nn_kernel<1>), dim3(total), dim3(256), 0, stream, a);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

int launch_tn(const TnProb* probs, int nprob, hipStream_t stream) {
  UFND_REQUIRE(nprob >= 1 && nprob <= UFND_GEMM_MAX_PROB, "tn: %d problems", nprob);
  TnArgs a;
  a.nprob = nprob;
  bool vec4 = true;
  for (int i = 0; i < nprob; ++i) {
    const TnProb& p = probs[i];
    UFND_REQUIRE(p.dY && p.X && p.dW && p.M > 0 && p.K > 0, "tn[%d]: null/empty operand", i);
    UFND_REQUIRE(p.N % 32 == 0, "tn[%d]: N=%d not a multiple of 32", i, p.N);
    UFND_REQUIRE(p.K % 2 == 0 && p.ldx % 2 == 0 && p.ldw % 2 == 0 && ufnd_aligned(p.X, 8) && ufnd_aligned(p.dW, 8),
                 "tn[%d]: X/dW alignment", i);
    if (!(p.K % 4 == 0 && p.ldx % 4 == 0 && p.ldw % 4 == 0 && ufnd_aligned(p.X, 16) && ufnd_aligned(p.dW, 16)))
      vec4 = false;
    UFND_REQUIRE(p.seg_rows >= 0 && (p.seg_rows == 0 || (p.M % p.seg_rows == 0 && p.seg_dy > 0 && p.seg_x > 0)), "tn[%d]: %d rows in segments of %d",
                 i, p.M, p.seg_rows);
    UFND_REQUIRE((p.seg_rows != 0) == (probs[0].seg_rows != 0), "tn[%d]: segmented and one-panel problems in one launch", i);
    if (p.seg_rows && !(p.seg_x % 4 == 0)) vec4 = false;
    UFND_REQUIRE(p.seg_rows == 0 || p.seg_x % 2 == 0, "tn[%d]: segment stride alignment", i);
    a.p[i] = p;
  }
  const bool seg = probs[0].seg_rows != 0;
  // Measured for the 32-row grouped launch in round 4 and not kept: 32 x 64 tiles (VEC = 2, twice the waves, half the MFMA chain each):
  // head-only step 0.254-0.256 -> 0.264 ms; a form whose accumulator v owns 32 CONSECUTIVE columns (4-byte accesses) and stores each
  // accumulator as soon as it is complete, so that the store phase starts under the matrix phase: 182 VGPRs (two waves per SIMD instead
  // of three), 24.7 -> 31.0 us for the launch (profiles/r04_head_tn_small.txt).  The 32 x 128 form with 16-B accesses stays.
  int minM = a.p[0].M;
  for (int i = 1; i < nprob; ++i) minM = a.p[i].M < minM ? a.p[i].M : minM;
  // one wave per tile (small batches) and mixed alignments: each problem keeps its own width (the same tiles and bits it would
  // have in a launch of its own)
  const bool mixed = minM < 128 && !vec4 && nprob > 1;
  const int VEC = vec4 ? 4 : 2;
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    const TnProb& q = a.p[i];
    const bool q4 = q.K % 4 == 0 && q.ldx % 4 == 0 && q.ldw % 4 == 0 && ufnd_aligned(q.X, 16) && ufnd_aligned(q.dW, 16) && (!q.seg_rows || q.seg_x % 4 == 0);
    a.vec[i] = mixed ? (q4 ? 4 : 2) : VEC;
    a.begin[i] = total;
    total += (q.N / 32) * ufnd_cdiv(q.K, 32 * a.vec[i]);
  }
  a.begin[nprob] = total;
  if (seg) {              // gathered factors: the same two forms, rows addressed by segment
    if (minM >= 128) {
      if (vec4) hipLaunchKernelGGL((tn_kernel<4, 1, 1>), dim3(total), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((tn_kernel<2, 1, 1>), dim3(total), dim3(256), 0, stream, a);
    } else {
      if (mixed) hipLaunchKernelGGL((tn_kernel<-1, 0, 1>), dim3(ufnd_cdiv(total, 4)), dim3(256), 0, stream, a);
      else if (vec4) hipLaunchKernelGGL((tn_kernel<4, 0, 1>), dim3(ufnd_cdiv(total, 4)), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((tn_kernel<2, 0, 1>), dim3(ufnd_cdiv(total, 4)), dim3(256), 0, stream, a);
    }
  } else if (minM >= 128) {      // batch rows split over the four waves of a workgroup (one workgroup per tile)
    if (vec4) hipLaunchKernelGGL((tn_kernel<4, 1>), dim3(total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((tn_kernel<2, 1>), dim3(total), dim3(256), 0, stream, a);
  } else {
    const int blocks = ufnd_cdiv(total, 4);
    if (mixed) hipLaunchKernelGGL((tn_kernel<-1>), dim3(blocks), dim3(256), 0, stream, a);
    else if (vec4) hipLaunchKernelGGL((tn_kernel<4>), dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((tn_kernel<2>), dim3(blocks), dim3(256), 0, stream, a);
  }
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
