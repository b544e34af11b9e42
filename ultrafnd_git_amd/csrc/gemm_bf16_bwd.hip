// Backward forms of the encoder Linears (Tier-B backward; the reference keeps its encoders frozen -- src/core_blocks/
// text_blocks.py:52,63 -- so these serve TrainConfig.train_encoders, SURVEY.md 8b's *_bwd list, not reference parity).
//
//   y = x W^T + b           x (M, K_in) bf16, W (N_out, K_in), y (M, N_out)
//   dgrad  dx = dy W        = dy (M, N_out) x Wt (K_in, N_out)^T   -> the forward's NT kernel on the TRANSPOSED weight copy
//                             (the trainable encoder re-casts fp32 masters to bf16 W and W^T once per step anyway); epilogue
//                             options: + fp32 residual (the other branch's gradient), x act'(pre) (GELU / quick-GELU fused)
//   wgrad  dW = dy^T x      = dyT (N_out, M) x xT (K_in, M)^T       -> the NT kernel again, on transposed activations, with the
//                             token dimension as K: long K loops, few output tiles, so the K range is cut into slices
//                             (grid = tiles x slices, fp32 partial slabs) and a reduce pass adds the slabs in slice order
//                             (deterministic: no atomics)
//   db     = column sums of dy, produced by the same pass that transposes dy (two-stage, fixed order)
#define UFND_GEMM_ONLY_BWD 1
#include "gemm_bf16_kernel.hpp"

namespace {

// (rows, cols) -> (cols, rows_pad) through a 64 x 64 LDS tile; 16-B global accesses on both sides.  TIn = float: cast to bf16
// on the way (weight masters -> transposed operand copies).  Optional column sums of the source (bias gradients): the block of
// row-tile rt writes its 64 partial sums to part[rt][col]; colsum_finish adds the row-tiles in ascending order.
template <typename TIn>
__device__ __forceinline__ void transpose_tile(const TIn* __restrict__ src, int rows, int cols, int lds, __bf16* __restrict__ dst,
                                               int ldd, int rows_pad, float* __restrict__ part, int bx, int by) {
  __shared__ __bf16 tile[64][72];                    // +8 columns: the transposed reads walk rows
  const int r0 = by * 64, c0 = bx * 64;
  const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;     // 32 rows x 8 chunks of 8 columns per pass
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = r0 + tr + 32 * p, c = c0 + tc;
    bf16x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (__bf16)0.0f;
    if (r < rows && c < cols) {                        // (cols is a multiple of 8)
      if constexpr (std::is_same<TIn, float>::value) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)r * lds + c), b = *reinterpret_cast<const f32x4*>(src + (size_t)r * lds + c + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = (__bf16)a[q]; v[4 + q] = (__bf16)b[q]; }
      } else {
        v = *reinterpret_cast<const bf16x8*>(src + (size_t)r * lds + c);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) csum[q] += (float)v[q];
    *reinterpret_cast<bf16x8*>(&tile[tr + 32 * p][tc]) = v;
  }
  __syncthreads();
  if (part) {     // column sums over the tile's 64 rows: thread (tr, tc) holds rows {tr, tr+32}; combine the 32 row-threads through LDS
    __shared__ float cs[32][64];
#pragma unroll
    for (int q = 0; q < 8; ++q) cs[tr][tc + q] = csum[q];
    __syncthreads();
    if (threadIdx.x < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) s += cs[k][threadIdx.x];
      if (c0 + (int)threadIdx.x < cols) part[(size_t)by * cols + c0 + threadIdx.x] = s;
    }
  }
  // write: dst row = source column, 8 consecutive source rows per 16-B store, transposed out of the tile by the hardware
  // (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block, lane i receives column i of the 4 rows; lane 4q+p
  // supplies the address of row q, columns 4p..4p+3).  Wave w takes source columns 16w..16w+15, group g row blocks 8g and 32+8g.
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  const int C = 16 * wave;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int R = 32 * it + 8 * g;
    union { s16x4 s2[2]; bf16x8 v; } u;
    u.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&tile[R + q][C + 4 * pq]));
    u.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&tile[R + 4 + q][C + 4 * pq]));
    if (c0 + C + i < cols && r0 + R < rows_pad) *reinterpret_cast<bf16x8*>(dst + (size_t)(c0 + C + i) * ldd + r0 + R) = u.v;
  }
}
template <typename TIn>
__global__ __launch_bounds__(256) void transpose_kernel(const TIn* __restrict__ src, int rows, int cols, int lds, __bf16* __restrict__ dst,
                                                        int ldd, int rows_pad, float* __restrict__ part) {
  transpose_tile<TIn>(src, rows, cols, lds, dst, ldd, rows_pad, part, blockIdx.x, blockIdx.y);
}
// Both operands of a weight gradient in ONE launch: blockIdx.z = 0: dy (M, N) -> dyT with the column-sum partials (the bias
// gradient), z = 1: x (M, K) -> xT.  grid.x covers the wider of the two; the other's surplus blocks leave at once.
__global__ __launch_bounds__(256) void transpose_pair_kernel(const __bf16* __restrict__ dy, int lddy, int N, const __bf16* __restrict__ x, int ldx, int K,
                                                             int rows, __bf16* __restrict__ dyT, __bf16* __restrict__ xT, int ldt, int rows_pad,
                                                             float* __restrict__ part) {
  if (blockIdx.z == 0) {
    if ((int)blockIdx.x * 64 < N) transpose_tile<__bf16>(dy, rows, N, lddy, dyT, ldt, rows_pad, part, blockIdx.x, blockIdx.y);
  } else {
    if ((int)blockIdx.x * 64 < K) transpose_tile<__bf16>(x, rows, K, ldx, xT, ldt, rows_pad, nullptr, blockIdx.x, blockIdx.y);
  }
}

// out[c] (+)= sum_t part[t][c].  16 columns x 16 row-groups per block: group j adds rows j, j + 16, ... in ascending order, the
// groups combine in a fixed tree -- enough blocks in flight for a (tiles x cols) panel of a few MB (3 blocks of 256 serial
// adders took 12 us per call), and still one fixed order of additions.
__device__ __forceinline__ void colsum_finish_block(const float* __restrict__ part, int tiles, int cols, float* __restrict__ out, int accumulate, int bx) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4, c = bx * 16 + cl;
  float s = 0.f;
  if (c < cols)
    for (int t = grp; t < tiles; t += 16) s += part[(size_t)t * cols + c];
  sh[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < cols) {
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
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int tiles, int cols, float* __restrict__ out, int accumulate) {
  colsum_finish_block(part, tiles, cols, out, accumulate, blockIdx.x);
}

__device__ __forceinline__ void slab_reduce_block(const float* __restrict__ slabs, int nslab, size_t slab_stride, size_t n, float* __restrict__ out,
                                                  int accumulate, int bx, int nb) {
  for (size_t i = ((size_t)bx * 256 + threadIdx.x) * 4; i < n; i += (size_t)nb * 256 * 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
    for (int k = 1; k < nslab; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (size_t)k * slab_stride + i);
      s += v;
    }
    if (accumulate) s += *reinterpret_cast<const f32x4*>(out + i);
    *reinterpret_cast<f32x4*>(out + i) = s;
  }
}
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, size_t n, float* __restrict__ out,
                                                          int accumulate) {
  slab_reduce_block(slabs, nslab, slab_stride, n, out, accumulate, blockIdx.x, gridDim.x);
}

// Every ordered finish pass behind one Linear's weight-gradient product in ONE launch, by block range:
//   [0, nb)                 the slab reduce of dW (slices added in slice order)
//   [nb, nb + cb)           the bias gradient: column-sum partials of dy added in row-tile order (colsum_finish)
//   [nb + cb, ... + 2 jb)   optionally a LayerNorm's dgamma / dbeta: the block partials of a layernorm_bwd launch that ran earlier
//                           (ufnd_layernorm_bwd with UFND_PARTIALS_DEFER) added in block order -- the same arithmetic as
//                           row_partials_finish_kernel in encoders_bwd.hip
// (three launches per Linear and one per LayerNorm were 245 launches of 4-9 us in a trainable-encoder step).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, size_t n, float* __restrict__ dW,
                                                           int accumulate, int nb, const float* __restrict__ cs_part, int cs_tiles, int cols,
                                                           float* __restrict__ db, int cb, ufnd_partials_job job) {
  const int b = blockIdx.x;
  if (b < nb) {
    slab_reduce_block(slabs, nslab, slab_stride, n, dW, accumulate, b, nb);
  } else if (b < nb + cb) {
    colsum_finish_block(cs_part, cs_tiles, cols, db, accumulate, b - nb);
  } else {
    const int jb = (job.H + 15) / 16, k = b - nb - cb, which = k / jb;
    float* out = which ? job.out1 : job.out0;
    if (!out) return;                     // (block-uniform)
    // part[blk][which][H]: a (nblk x H) panel with row stride 2 H
    __shared__ float sh[16][17];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4, c = (k - which * jb) * 16 + cl;
    float s = 0.f;
    if (c < job.H)
      for (int t = grp; t < job.nblk; t += 16) s += job.part[((size_t)t * 2 + which) * job.H + c];
    sh[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && c < job.H) {
      float a[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = sh[q][cl];
#pragma unroll
      for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int q = 0; q < w; ++q) a[q] += a[q + w];
      out[c] = a[0];
    }
  }
}

// bf16 W and W^T of MANY Linears from their fp32 masters in one launch (after an optimizer step): tile t of the launch belongs to
// the item whose [tile0, tile0 + tiles) range holds it (looked up in the device-resident table); a tile is 64 x 64: read
// once (fp32, 16-B loads), written twice (bf16 rows of W; bf16 rows of W^T through the LDS tile and the transposing LDS read).
__global__ __launch_bounds__(256) void refresh_operands_kernel(const ufnd_refresh_item* __restrict__ items, int n_items) {
  // the item whose [tile0, next tile0) holds this block: every thread tests one item (ONE round of loads; a binary search by one
  // thread is seven DEPENDENT loads, 5 us in front of a block that moves 32 KB -- 230 us per launch instead of 140)
  __shared__ int owner;
  const int t = blockIdx.x;
  for (int i = threadIdx.x; i < n_items; i += 256) {
    const int lo = items[i].tile0, hi = i + 1 < n_items ? items[i + 1].tile0 : 0x7fffffff;
    if (t >= lo && t < hi) owner = i;
  }
  __syncthreads();
  const ufnd_refresh_item it = items[owner];
  const int local = t - it.tile0, ctiles = it.cols >> 6;
  const int by = local / ctiles, bx = local - by * ctiles;
  __shared__ __bf16 tile[64][72];
  const float* src = static_cast<const float*>(it.master);
  __bf16* w = static_cast<__bf16*>(it.w);
  __bf16* wt = static_cast<__bf16*>(it.wt);
  const int r0 = by * 64, c0 = bx * 64;
  const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = r0 + tr + 32 * p, c = c0 + tc;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)r * it.ld_master + c), b = *reinterpret_cast<const f32x4*>(src + (size_t)r * it.ld_master + c + 4);
    bf16x8 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = (__bf16)a[q]; v[4 + q] = (__bf16)b[q]; }
    *reinterpret_cast<bf16x8*>(w + (size_t)r * it.ld_w + c) = v;
    *reinterpret_cast<bf16x8*>(&tile[tr + 32 * p][tc]) = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
  const int C = 16 * wave;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int R = 32 * k + 8 * g;
    union { s16x4 s2[2]; bf16x8 v; } u;
    u.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&tile[R + q][C + 4 * pq]));
    u.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&tile[R + 4 + q][C + 4 * pq]));
    *reinterpret_cast<bf16x8*>(wt + (size_t)(c0 + C + i) * it.ld_wt + r0 + R) = u.v;
  }
}

// A weight-gradient product has few output tiles and a long K (the tokens): the K range is cut into slices of at least eight
// 64-token K-steps.  Tile: 128 x 128 first when it fills a round of workgroups with the slices available (round 4: with 256 x 256
// tiles first an FFN gradient needed 8 slices -- nine passes over a 9.4 MB slab set per Linear -- where 128 x 128 needs 4: the step
// 10.58-10.73 -> 10.18 ms on one box, A/B/A; 128 x 64 first: another 0.8 %, inside the noise), else the largest that does.
int wgrad_cfg(int n_out, int k_in, int m_tokens) {
  const int smax = m_tokens / 64 / 8 > 1 ? m_tokens / 64 / 8 : 1;
#ifndef UFND_WGRAD_FIRST
#define UFND_WGRAD_FIRST 16
#endif
  const int cand[5] = {UFND_WGRAD_FIRST, 15, 22, 16, 20};      // 256x256, 256x192, 128x128, 128x64 (experiments: --defs=UFND_WGRAD_FIRST=16 tries that tile first)
  for (int i = 0; i < 5; ++i) {
    const TileCfg& t = kTiles[cand[i]];
    if (k_in % t.bn) continue;
    const long long tiles = (long long)ufnd_cdiv(n_out, t.bm) * (k_in / t.bn);
    if (tiles * smax >= 256 || i == 4) return cand[i];
  }
  return 20;
}
int wgrad_slices(int n_out, int k_in, int m_tokens, int cfg) {
  const long long tiles = (long long)ufnd_cdiv(n_out, kTiles[cfg].bm) * (k_in / kTiles[cfg].bn);
  const int nk = m_tokens / 64;
#ifndef UFND_WGRAD_BLOCKS
#define UFND_WGRAD_BLOCKS 128
#endif
  long long s = (UFND_WGRAD_BLOCKS + tiles - 1) / tiles;             // half a round of workgroups: the two encoders' backward chains run side by side, so a
                                                                     // launch that leaves CUs free costs little and every slice is another pass over the slab set
                                                                     // (step, one box: 512 blocks 9.93-10.03 ms, 256 9.68-9.82, 192 9.53, 128 9.36-9.41, 64 9.47, no split 9.59)
  if (s > nk / 8) s = nk / 8;
  if (s < 1) s = 1;
  const int per = (nk + (int)s - 1) / (int)s;          // every slice must own at least one K-step
  return (nk + per - 1) / per;
}

}  // namespace

extern "C" int ufnd_gemm_bf16_dgrad(const void* dY, const void* Wt, const float* residual, const void* aux, void* out_bf16, float* out_f32,
                                    int M, int N, int K, int lda, int ldw, int ldr, int ldaux, int ldo, int ldf, int act, void* stream_) {
  UFND_REQUIRE(dY && Wt && (out_bf16 || out_f32), "gemm_bf16_dgrad: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "gemm_bf16_dgrad: M=%d N=%d K=%d (need N%%64==0, K%%64==0)", M, N, K);
  UFND_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ufnd_aligned(dY, 16) && ufnd_aligned(Wt, 16),
               "gemm_bf16_dgrad: operand strides must be multiples of 8 and pointers 16-B aligned");
  UFND_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= N && ufnd_aligned(residual, 16)), "gemm_bf16_dgrad: residual alignment");
  UFND_REQUIRE(!out_f32 || (ldf % 4 == 0 && ldf >= N && ufnd_aligned(out_f32, 16)), "gemm_bf16_dgrad: out_f32 alignment");
  UFND_REQUIRE(!out_bf16 || (ldo % 8 == 0 && ldo >= N && ufnd_aligned(out_bf16, 16)), "gemm_bf16_dgrad: out_bf16 alignment");
  UFND_REQUIRE(act == UFND_ACT_NONE || act == UFND_ACT_GELU_BWD || act == UFND_ACT_QUICK_GELU_BWD, "gemm_bf16_dgrad: act=%d", act);
  UFND_REQUIRE((act == UFND_ACT_NONE) == (aux == nullptr), "gemm_bf16_dgrad: aux (the pre-activations) goes with an activation backward, and only with one");
  UFND_REQUIRE(!aux || (!residual && ldaux % 8 == 0 && ldaux >= N && ufnd_aligned(aux, 16)), "gemm_bf16_dgrad: aux alignment (and no residual beside it)");
  GemmArgs a{(const __bf16*)dY, (const __bf16*)Wt, nullptr, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0, nullptr};
  a.aux = (const __bf16*)aux;
  a.ldaux = ldaux;
  a.ksplit = 1;
  const int cfg = auto_cfg(M, N, K);
  UFND_REQUIRE(N % kTiles[cfg].bn == 0, "gemm_bf16_dgrad: tile %d needs N %% %d == 0", cfg, kTiles[cfg].bn);
  int rc = launch_cfg(cfg, 6, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" size_t ufnd_gemm_bf16_wgrad_workspace_floats(int n_out, int k_in, int m_tokens) {
  if (n_out < 64 || k_in < 64 || m_tokens < 64 || k_in % 64 || m_tokens % 64) return 0;
  const int cfg = wgrad_cfg(n_out, k_in, m_tokens);
  return (size_t)wgrad_slices(n_out, k_in, m_tokens, cfg) * (size_t)n_out * (size_t)k_in;
}

extern "C" int ufnd_gemm_bf16_wgrad(const void* dYt, const void* Xt, float* dW, int n_out, int k_in, int m_tokens, int lda, int ldb, int ldw,
                                    float* workspace, int accumulate, void* stream_) {
  UFND_REQUIRE(dYt && Xt && dW && workspace, "gemm_bf16_wgrad: null operand");
  UFND_REQUIRE(n_out >= 1 && k_in >= 64 && k_in % 64 == 0 && m_tokens >= 64 && m_tokens % 64 == 0,
               "gemm_bf16_wgrad: N_out=%d K_in=%d tokens=%d (K_in and the padded token count must be multiples of 64)", n_out, k_in, m_tokens);
  UFND_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= m_tokens && ldb >= m_tokens && ufnd_aligned(dYt, 16) && ufnd_aligned(Xt, 16),
               "gemm_bf16_wgrad: transposed operands need 16-B aligned rows of >= tokens elements");
  UFND_REQUIRE(ldw == k_in && ufnd_aligned(dW, 16) && ufnd_aligned(workspace, 16), "gemm_bf16_wgrad: dW must be dense (ldw == K_in) and 16-B aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int cfg = wgrad_cfg(n_out, k_in, m_tokens);
  UFND_REQUIRE(k_in % kTiles[cfg].bn == 0, "gemm_bf16_wgrad: tile %d needs K_in %% %d == 0", cfg, kTiles[cfg].bn);
  const int S = wgrad_slices(n_out, k_in, m_tokens, cfg);
  const size_t n = (size_t)n_out * k_in;
  GemmArgs a{(const __bf16*)dYt, (const __bf16*)Xt, nullptr, nullptr, nullptr, workspace, n_out, k_in, m_tokens, lda, ldb, 0, 0, k_in, UFND_ACT_NONE, 0, 0, nullptr};
  a.ksplit = S;
  a.slab_stride = n;
  int rc = launch_cfg(cfg, 6, a, stream);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  size_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, S, n, n, dW, accumulate);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_linear_wgrad(const void* dY, int lddy, const void* X, int ldx, int M, int N, int K, float* dW, float* db, void* dYt, void* Xt, int ldt,
                                 float* slab_workspace, float* colsum_workspace, const ufnd_partials_job* extra, int part, void* stream_) {
  UFND_REQUIRE(part == UFND_WGRAD_ALL || part == UFND_WGRAD_TRANSPOSE || part == UFND_WGRAD_PRODUCT, "linear_wgrad: part=%d", part);
  UFND_REQUIRE(dY && X && dW && dYt && Xt && slab_workspace, "linear_wgrad: null operand");
  UFND_REQUIRE(M >= 1 && N >= 8 && N % 8 == 0 && K >= 64 && K % 64 == 0, "linear_wgrad: M=%d N=%d K=%d (N %% 8 == 0, K %% 64 == 0)", M, N, K);
  const int Mp = (M + 63) / 64 * 64;
  UFND_REQUIRE(ldt >= Mp && ldt % 8 == 0 && lddy >= N && lddy % 8 == 0 && ldx >= K && ldx % 8 == 0, "linear_wgrad: strides (ldt >= %d)", Mp);
  UFND_REQUIRE(ufnd_aligned(dY, 16) && ufnd_aligned(X, 16) && ufnd_aligned(dYt, 16) && ufnd_aligned(Xt, 16) && ufnd_aligned(dW, 16) && ufnd_aligned(slab_workspace, 16),
               "linear_wgrad: 16-B alignment");
  UFND_REQUIRE(!db || colsum_workspace, "linear_wgrad: the bias gradient needs the column-sum workspace (ufnd_transpose_colsum_workspace_floats)");
  UFND_REQUIRE(!extra || (extra->part && extra->nblk >= 1 && extra->H >= 16 && (extra->out0 || extra->out1)), "linear_wgrad: extra finish job");
  hipStream_t stream = (hipStream_t)stream_;
  const int wide = N > K ? N : K, row_tiles = Mp / 64;
  if (part != UFND_WGRAD_PRODUCT) {
    hipLaunchKernelGGL(transpose_pair_kernel, dim3(ufnd_cdiv(wide, 64), row_tiles, 2), dim3(256), 0, stream, (const __bf16*)dY, lddy, N, (const __bf16*)X, ldx, K, M,
                       (__bf16*)dYt, (__bf16*)Xt, ldt, Mp, db ? colsum_workspace : (float*)nullptr);
    UFND_CHECK_LAUNCH();
    if (part == UFND_WGRAD_TRANSPOSE) return UFND_OK;
  }
  const int cfg = wgrad_cfg(N, K, Mp);
  UFND_REQUIRE(K % kTiles[cfg].bn == 0, "linear_wgrad: tile %d needs K %% %d == 0", cfg, kTiles[cfg].bn);
  const int S = wgrad_slices(N, K, Mp, cfg);
  const size_t n = (size_t)N * K;
  GemmArgs a{(const __bf16*)dYt, (const __bf16*)Xt, nullptr, nullptr, nullptr, slab_workspace, N, K, Mp, ldt, ldt, 0, 0, K, UFND_ACT_NONE, 0, 0, nullptr};
  a.ksplit = S;
  a.slab_stride = n;
  int rc = launch_cfg(cfg, 6, a, stream);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  size_t want = (n / 4 + 255) / 256;
  const int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  const int cb = db ? ufnd_cdiv(N, 16) : 0;
  ufnd_partials_job job{nullptr, 0, 16, nullptr, nullptr};
  int jb = 0;
  if (extra) {
    job = *extra;
    jb = 2 * ufnd_cdiv(job.H, 16);
  }
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(nb + cb + jb), dim3(256), 0, stream, (const float*)slab_workspace, S, n, n, dW, 0, nb,
                     (const float*)colsum_workspace, row_tiles, N, db, cb, job);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_refresh_operands(const ufnd_refresh_item* items_device, int n_items, int total_tiles, void* stream_) {
  UFND_REQUIRE(items_device && n_items >= 1 && total_tiles >= 1, "refresh_operands: items=%d tiles=%d", n_items, total_tiles);
  hipLaunchKernelGGL(refresh_operands_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream_, items_device, n_items);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" size_t ufnd_transpose_colsum_workspace_floats(int rows_pad, int cols) { return (size_t)ufnd_cdiv(rows_pad, 64) * (size_t)cols; }

extern "C" int ufnd_transpose_bf16(const void* src, int src_is_f32, int rows, int cols, int lds, void* dst, int ldd, int rows_pad,
                                   float* colsum, float* colsum_ws, int colsum_accumulate, void* stream_) {
  UFND_REQUIRE(src && dst && rows >= 1 && cols >= 8 && cols % 8 == 0, "transpose_bf16: rows=%d cols=%d (cols %% 8 == 0)", rows, cols);
  UFND_REQUIRE(rows_pad >= rows && rows_pad % 8 == 0 && ldd >= rows_pad && ldd % 8 == 0 && lds >= cols && lds % (src_is_f32 ? 4 : 8) == 0,
               "transpose_bf16: rows_pad=%d ldd=%d lds=%d", rows_pad, ldd, lds);
  UFND_REQUIRE(ufnd_aligned(src, 16) && ufnd_aligned(dst, 16), "transpose_bf16: 16-B alignment");
  UFND_REQUIRE(!colsum || (colsum_ws && !src_is_f32), "transpose_bf16: column sums need a workspace (ufnd_transpose_colsum_workspace_floats) and a bf16 source");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(ufnd_cdiv(cols, 64), ufnd_cdiv(rows_pad, 64));
  if (src_is_f32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, stream, (const float*)src, rows, cols, lds, (__bf16*)dst, ldd, rows_pad, (float*)nullptr);
  else
    hipLaunchKernelGGL(transpose_kernel<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)src, rows, cols, lds, (__bf16*)dst, ldd, rows_pad,
                       colsum ? colsum_ws : (float*)nullptr);
  UFND_CHECK_LAUNCH();
  if (colsum) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(ufnd_cdiv(cols, 16)), dim3(256), 0, stream, colsum_ws, (int)grid.y, cols, colsum, colsum_accumulate);
    UFND_CHECK_LAUNCH();
  }
  return UFND_OK;
}
