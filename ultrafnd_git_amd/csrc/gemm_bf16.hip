// bf16 GEMM for the encoder Linears: out = act(A W^T + bias) + residual.
//
// A (M,K) and W (N,K) are both K-contiguous, which is exactly the operand shape of
// v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15]).
// Tile 128 x BN x 64, 4 waves as 2x2 (wave tile 64 x BN/2 = 4 x BN/32 MFMA tiles).
// Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane): the LDS image is
// lane-linear, so the bank-conflict swizzle (16-B chunk ^= (row>>1)&7, conflict-free for
// ds_read_b128 of 128-B rows) is applied to each lane's SOURCE address and to the read address.
// Two LDS stages: the DMA of K-step t+1 is in flight while step t is multiplied.
// Epilogue: accumulators -> LDS (per wave) -> whole rows: bias, activation, fp32 residual,
// 16-B bf16 and/or 32-B fp32 stores per lane (full cache lines per row).
// Grid: one block per tile, XCD-aware (bijective) remap so that the blocks sharing an A
// row-panel land on the same XCD's L2.
#include "common.hpp"

namespace {

constexpr int BM = 128, BK = 64;

struct GemmArgs {
  const __bf16* A;
  const __bf16* W;
  const float* bias;
  const float* residual;
  __bf16* out_bf16;
  float* out_f32;
  int M, N, K, lda, ldw, ldr, ldo, ldf, act;
  int m_tiles, n_tiles;
};

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

// stage ROWS rows x 64 bf16 (128 B per row) of a K-contiguous matrix into LDS at `tile`
template <int ROWS>
__device__ __forceinline__ void stage_tile(const __bf16* G, int ld, int row0, int row_max, int k0, char* tile, int wave,
                                           int lane) {
#pragma unroll
  for (int ii = 0; ii < ROWS / 32; ++ii) {
    const int i = wave + 4 * ii;
    const int r = 8 * i + (lane >> 3), pos = lane & 7;
    const int c = pos ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_max ? gr : row_max - 1;
    dma16(G + (size_t)gr * ld + k0 + c * 8, tile + i * 1024);
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int BN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs a) {
  constexpr int WN = BN / 2;            // wave tile width
  constexpr int NT = WN / 16;           // MFMA tiles across
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int CP = WN + 4;            // fp32 C-staging pitch (floats)
  constexpr int CBYTES = 4 * 64 * CP * 4;
  constexpr int SMEM = (2 * STAGE > CBYTES) ? 2 * STAGE : CBYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  // XCD-aware bijective remap of the block id
  const int nblk = a.m_tiles * a.n_tiles;
  int bid;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / a.n_tiles) * BM, n0 = (bid % a.n_tiles) * BN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, g = lane >> 4;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
  stage_tile<BM>(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
  stage_tile<BN>(a.W, a.ldw, n0, a.N, 0, smem + BM * 128, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * STAGE;
    if (t + 1 < nk) {
      char* nxt = smem + ((t + 1) & 1) * STAGE;
      stage_tile<BM>(a.A, a.lda, m0, a.M, (t + 1) * BK, nxt, wave, lane);
      stage_tile<BN>(a.W, a.ldw, n0, a.N, (t + 1) * BK, nxt + BM * 128, wave, lane);
    }
    const char* At = cur;
    const char* Bt = cur + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_frag(At, wm * 64 + i * 16 + fr, g + 4 * kk);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = lds_frag(Bt, wn * WN + j * 16 + fr, g + 4 * kk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of stage t+1 have landed
    __syncthreads();                                   // ... everyone's have; stage t is free to overwrite
  }

  // ---- epilogue: per-wave transpose through LDS, then whole-row stores
  float* cst = reinterpret_cast<float*>(smem) + wave * 64 * CP;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cst[(i * 16 + 4 * g + r) * CP + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes done (wave-private region)
  constexpr int LPR = WN / 8;           // lanes per row (8 columns each)
  constexpr int RPI = 64 / LPR;         // rows per iteration
  const int cl = (lane % LPR) * 8, rl = lane / LPR;
  const int col = n0 + wn * WN + cl;
  float bias[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bias[q] = a.bias ? a.bias[col + q] : 0.0f;
#pragma unroll
  for (int it = 0; it < 64 / RPI; ++it) {
    const int rr = it * RPI + rl;
    const int row = m0 + wm * 64 + rr;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl + 4);
    if (row >= a.M) continue;
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float x = v[q] + bias[q];
      if (a.act == UFND_ACT_GELU) x = gelu_f(x);
      else if (a.act == UFND_ACT_QUICK_GELU) x = x * sigmoid_f(1.702f * x);
      v[q] = x;
    }
    if (a.residual) {
      const float* rp = a.residual + (size_t)row * a.ldr + col;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] += r0[q]; v[4 + q] += r1[q]; }
    }
    if (a.out_f32) {
      float* op = a.out_f32 + (size_t)row * a.ldf + col;
      *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (a.out_bf16) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (__bf16)v[q];
      *reinterpret_cast<bf16x8*>(a.out_bf16 + (size_t)row * a.ldo + col) = o;
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, __bf16* dst, size_t n) {
  for (size_t i = (blockIdx.x * (size_t)256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
    if (i + 4 <= n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
      bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<bf16x4*>(dst + i) = o;
    } else {
      for (size_t j = i; j < n; ++j) dst[j] = (__bf16)src[j];
    }
  }
}

}  // namespace

extern "C" int ufnd_gemm_bf16(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                              float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                              void* stream_) {
  UFND_REQUIRE(A && W && (out_bf16 || out_f32), "gemm_bf16: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "gemm_bf16: M=%d N=%d K=%d (need N%%64==0, K%%64==0)", M, N, K);
  UFND_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ufnd_aligned(A, 16) && ufnd_aligned(W, 16),
               "gemm_bf16: A/W strides must be multiples of 8 and pointers 16-B aligned");
  UFND_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= N && ufnd_aligned(residual, 16)), "gemm_bf16: residual alignment");
  UFND_REQUIRE(!out_f32 || (ldf % 4 == 0 && ldf >= N && ufnd_aligned(out_f32, 16)), "gemm_bf16: out_f32 alignment");
  UFND_REQUIRE(!out_bf16 || (ldo % 8 == 0 && ldo >= N && ufnd_aligned(out_bf16, 16)), "gemm_bf16: out_bf16 alignment");
  UFND_REQUIRE(!bias || ufnd_aligned(bias, 4), "gemm_bf16: bias alignment");
  UFND_REQUIRE(act >= 0 && act <= 2, "gemm_bf16: act=%d", act);
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0};
  a.m_tiles = ufnd_cdiv(M, BM);
  // pick the tile width that gives the chip (256 CUs) enough blocks
  const bool wide = (N % 128 == 0) && ((long long)a.m_tiles * (N / 128) >= 384);
  hipStream_t stream = (hipStream_t)stream_;
  if (wide) {
    a.n_tiles = N / 128;
    hipLaunchKernelGGL((gemm_bf16_kernel<128>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, stream, a);
  } else {
    a.n_tiles = N / 64;
    hipLaunchKernelGGL((gemm_bf16_kernel<64>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, stream, a);
  }
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_cast_bf16(const float* src, void* dst, size_t n, void* stream_) {
  UFND_REQUIRE(src && dst && n > 0, "cast_bf16: null argument");
  UFND_REQUIRE(ufnd_aligned(src, 16) && ufnd_aligned(dst, 8), "cast_bf16: alignment");
  size_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, src, (__bf16*)dst, n);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
