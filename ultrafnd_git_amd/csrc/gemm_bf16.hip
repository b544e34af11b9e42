// bf16 GEMM for the encoder Linears: out = act(A W^T + bias) + residual.
//
// A (M,K) and W (N,K) are both K-contiguous, which is exactly the operand shape of
// v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15]).
// Tile BM x BN x 64, WM x WN waves, wave tile (BM/WM) x (BN/WN) of 16x16 MFMA tiles.
// Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane): the LDS image is
// lane-linear, so the bank-conflict swizzle (16-B chunk ^= (row>>1)&7, conflict-free for
// ds_read_b128 of 128-B rows) is applied to each lane's SOURCE address and to the read address.
// ST-deep LDS ring with counted vmcnt and ONE raw s_barrier per K-step: the DMA of steps
// t+1..t+ST-2 stays in flight across the barrier while step t is multiplied.
// Epilogue: accumulators -> LDS (per wave) -> whole rows: bias, activation, fp32 residual,
// 16-B bf16 and/or 32-B fp32 stores per lane (full cache lines per row).
// Grid: one block per tile, XCD-aware (bijective) remap so that the blocks sharing an A
// row-panel land on the same XCD's L2.
#include "common.hpp"

namespace {

constexpr int BK = 64;

struct GemmArgs {
  const __bf16* A;
  const __bf16* W;
  const float* bias;
  const float* residual;
  __bf16* out_bf16;
  float* out_f32;
  int M, N, K, lda, ldw, ldr, ldo, ldf, act;
  int m_tiles, n_tiles;
  int ksplit;            // > 1: block (tile, s) multiplies K-slice s and writes raw fp32 partials to slab s
};

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x BN x 64 block tile, WM x WN waves, wave tile (BM/WM) x (BN/WN).  The A rows and the W rows of
// a K-step form ONE list of (BM+BN)/8 1-KiB DMA pieces dealt round-robin to the waves, so every
// wave has exactly PW pieces per stage in flight and a counted s_waitcnt vmcnt is exact.
// ST-deep LDS ring:
//   step t:  wait (my pieces of stage t landed; stages t+1.. stay in flight)  ->  ONE s_barrier
//            (everyone's pieces landed, and everyone is done reading stage t-1)  ->  issue stage
//            t+ST-1 into the buffer stage t-1 used  ->  multiply stage t.
// ABL (timing experiments only, results are garbage): 1 = no MFMA/LDS reads, 2 = no DMA in the loop
// SCH: 0 = compiler's schedule; 1 = explicit: both k-halves' fragment reads issued first (the second
// half lands while the first half multiplies), then the stage's DMA pieces spread between MFMA groups.
template <int BM, int BN, int WM, int WN, int ST, int ABL = 0, int SCH = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const GemmArgs a) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;      // wave tile
  constexpr int MT = TM / 16, NT = TN / 16;      // MFMA tiles per wave
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int PIECES = (BM + BN) / 8;
  constexpr int PW = PIECES / NW;                // DMA wave-instructions per wave per stage
  constexpr int CP = TN + 4;                     // fp32 C-staging pitch (floats), 16 rows per wave
  constexpr int CBYTES = NW * 16 * CP * 4;
  constexpr int SMEM = (ST * STAGE > CBYTES) ? ST * STAGE : CBYTES;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(PIECES % NW == 0 && TM % 16 == 0 && TN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile split");
  static_assert(PW * (ST - 2 > 0 ? ST - 2 : 0) <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  // XCD-aware bijective remap of the block id
  const int nblk = a.m_tiles * a.n_tiles * a.ksplit;
  int bid;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ks = bid % a.ksplit;
  bid /= a.ksplit;
  const int m0 = (bid / a.n_tiles) * BM, n0 = (bid % a.n_tiles) * BN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN, fr = lane & 15, g = lane >> 4;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK / a.ksplit;         // K-steps of this block's slice
  const int kbase = ks * nk * BK;
  auto issue = [&](int t) {
    char* buf = smem + (t % ST) * STAGE;
    const int k0 = kbase + t * BK;
#pragma unroll
    for (int ii = 0; ii < PW; ++ii) {
      const int p = wave + NW * ii;                  // piece id: rows 8p..8p+7 of [A tile ; W tile]
      const int r = 8 * p + (lane >> 3), pos = lane & 7;
      const int c = pos ^ ((r >> 1) & 7);            // (BM % 16 == 0 keeps the swizzle phase per tile)
      const __bf16* src;
      if (p < BM / 8) {
        int gr = m0 + r;
        gr = gr < a.M ? gr : a.M - 1;
        src = a.A + (size_t)gr * a.lda + k0 + c * 8;
      } else {
        int gr = n0 + (r - BM);
        gr = gr < a.N ? gr : a.N - 1;
        src = a.W + (size_t)gr * a.ldw + k0 + c * 8;
      }
      dma16(src, buf + p * 1024);
    }
  };
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < nk) issue(t);

  for (int t = 0; t < nk; ++t) {
    // stages issued so far: min(nk, t + ST - 1); those after t may stay in flight
    if (t + ST - 1 <= nk) wait_vmcnt<PW*(ST - 2 > 0 ? ST - 2 : 0)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* At = smem + (t % ST) * STAGE;
    const char* Bt = At + BM * 128;
    if constexpr (SCH == 0 || ABL != 0) {
      if (ABL != 2 && t + ST - 1 < nk) issue(t + ST - 1);
#pragma unroll
      for (int kk = 0; kk < (ABL == 1 ? 0 : 2); ++kk) {
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = lds_frag(At, wm * TM + i * 16 + fr, g + 4 * kk);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[j] = lds_frag(Bt, wn * TN + j * 16 + fr, g + 4 * kk);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
      bf16x8 af[2][MT], bfr[2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = lds_frag(At, wm * TM + i * 16 + fr, g + 4 * kk);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[kk][j] = lds_frag(Bt, wn * TN + j * 16 + fr, g + 4 * kk);
      }
      if (t + ST - 1 < nk) issue(t + ST - 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
      // shape of the emitted stream: all 2(MT+NT) LDS reads, then {G MFMAs, 1 DMA piece} x PW, then the rest
      constexpr int NMF = 2 * MT * NT, GRP = NMF / (PW + 1) > 0 ? NMF / (PW + 1) : 1;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, GRP, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMF - GRP * PW > 0 ? NMF - GRP * PW : 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's LDS reads retired before the next barrier
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();   // everyone is done reading the ring: reuse it for the epilogue
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue.  Bias and activation are elementwise, so they run on the accumulators where they
  // lie (straight-line VALU, no LDS dependency); only the layout change for whole-row stores goes
  // through a wave-private LDS patch, 16 rows at a time, where the fp32 residual is added.
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float bj = a.bias[n0 + wn * TN + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] += bj;
    }
  }
  if (a.act == UFND_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_fast_f(acc[i][j][r]);
  } else if (a.act == UFND_ACT_QUICK_GELU) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * sigmoid_f(1.702f * acc[i][j][r]);
  }
  float* cst = reinterpret_cast<float*>(smem) + wave * 16 * CP;
  constexpr int CPR = TN / 8;                    // 8-column chunks per row
  constexpr int CHUNKS = 16 * CPR;               // chunks per 16-row patch
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cst[(4 * g + r) * CP + j * 16 + fr] = acc[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < (CHUNKS + 63) / 64; ++it) {
      const int id = lane + 64 * it;
      if (id >= CHUNKS) break;
      const int rr = id / CPR, cl = (id % CPR) * 8;
      const int row = m0 + wm * TM + i * 16 + rr;
      const int col = n0 + wn * TN + cl;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl + 4);
      if (row >= a.M) continue;
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (a.residual) {
        const float* rp = a.residual + (size_t)row * a.ldr + col;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += r0[q]; v[4 + q] += r1[q]; }
      }
      if (a.out_f32) {
        float* op = a.out_f32 + (size_t)ks * a.M * a.ldf + (size_t)row * a.ldf + col;
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before the next row-tile overwrites it
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

// tile configurations (BM x BN, waves WM x WN, LDS stages).  Exact-fit widths exist because a
// 256-CU chip wants ~256 (or 512 at 2 blocks/CU) equal tiles per launch, not "as many as it takes".
struct TileCfg { int bm, bn, threads; };
static const TileCfg kTiles[] = {
    {128, 128, 256},  //  0: 128x128 2x2 3-stage   96 KiB
    {128, 64, 256},   //  1: 128x64  2x2 3-stage   72 KiB (2 blocks/CU)
    {256, 128, 512},  //  2: 256x128 4x2 3-stage  144 KiB
    {128, 128, 256},  //  3: 128x128 2x2 2-stage   64 KiB (2 blocks/CU)
    {128, 64, 256},   //  4: 128x64  2x2 4-stage   96 KiB
    {256, 64, 512},   //  5: 256x64  4x2 3-stage  120 KiB
    {128, 288, 256},  //  6: 128x288 2x2 3-stage  156 KiB  (N=2304 -> 8 column tiles)
    {128, 96, 256},   //  7: 128x96  2x2 3-stage   84 KiB  (N=768 -> 8 column tiles)
    {256, 192, 512},  //  8: 256x192 4x2 2-stage  112 KiB  (N=3072 -> 16 column tiles)
    {128, 384, 256},  //  9: 128x384 2x2 2-stage  128 KiB  (N=3072 -> 8 column tiles)
    {128, 192, 256},  // 10: 128x192 2x2 3-stage  120 KiB
    {128, 96, 256},   // 11: 128x96  2x2 4-stage  112 KiB
    {64, 96, 128},    // 12: 64x96   1x2 4-stage   80 KiB  (2 blocks/CU; ViT M=1600 -> 25 row tiles)
    {64, 192, 128},   // 13: 64x192  1x2 3-stage   96 KiB
    {128, 256, 256},  // 14: 128x256 2x2 2-stage   96 KiB
    {256, 256, 512},  // 15: 256x256 4x2 2-stage  128 KiB
    {128, 128, 512},  // 16: 128x128 4x2 3-stage   96 KiB (8 waves, wave tile 32x64)
    {128, 192, 512},  // 17: 128x192 4x2 3-stage  120 KiB (8 waves, wave tile 32x96)
    {256, 64, 512},   // 18: 256x64  4x2 4-stage  160 KiB
    {64, 64, 128},    // 19: 64x64   1x2 4-stage   64 KiB (2 blocks/CU)
    {128, 64, 512},   // 20: 128x64  4x2 4-stage   96 KiB (8 waves, wave tile 32x32)
    {256, 192, 512},  // 21: 256x192 2x4 2-stage  112 KiB (wave tile 128x48)
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

static int launch_cfg(int cfg, int abl, int sch, GemmArgs& a, hipStream_t stream) {
  const TileCfg& t = kTiles[cfg];
  a.m_tiles = ufnd_cdiv(a.M, t.bm);
  a.n_tiles = a.N / t.bn;
  const dim3 grid(a.m_tiles * a.n_tiles * a.ksplit), block(t.threads);
#define GO(BM_, BN_, WM_, WN_, ST_)                                                                               \
  do {                                                                                                              \
    if (abl == 0 && sch == 1) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, ST_, 0, 1>), grid, block, 0, stream, a); \
    else if (abl == 0) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, ST_, 0>), grid, block, 0, stream, a);   \
    else if (abl == 1) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, ST_, 1>), grid, block, 0, stream, a); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, ST_, 2>), grid, block, 0, stream, a);           \
  } while (0)
  switch (cfg) {
    case 0: GO(128, 128, 2, 2, 3); break;
    case 1: GO(128, 64, 2, 2, 3); break;
    case 2: GO(256, 128, 4, 2, 3); break;
    case 3: GO(128, 128, 2, 2, 2); break;
    case 4: GO(128, 64, 2, 2, 4); break;
    case 5: GO(256, 64, 4, 2, 3); break;
    case 6: GO(128, 288, 2, 2, 3); break;
    case 7: GO(128, 96, 2, 2, 3); break;
    case 8: GO(256, 192, 4, 2, 2); break;
    case 9: GO(128, 384, 2, 2, 2); break;
    case 10: GO(128, 192, 2, 2, 3); break;
    case 11: GO(128, 96, 2, 2, 4); break;
    case 12: GO(64, 96, 1, 2, 4); break;
    case 13: GO(64, 192, 1, 2, 3); break;
    case 14: GO(128, 256, 2, 2, 2); break;
    case 15: GO(256, 256, 4, 2, 2); break;
    case 16: GO(128, 128, 4, 2, 3); break;
    case 17: GO(128, 192, 4, 2, 3); break;
    case 18: GO(256, 64, 4, 2, 4); break;
    case 19: GO(64, 64, 1, 2, 4); break;
    case 20: GO(128, 64, 4, 2, 4); break;
    case 21: GO(256, 192, 2, 4, 2); break;
    default: ufnd_set_error("gemm_bf16: unknown tile config %d", cfg); return UFND_ERR_INVALID;
  }
#undef GO
  return UFND_OK;
}

// Per-shape choice from the on-device sweep (tools/gemm_sweep.py, profiles/r01_gemm_sweep.md):
// a launch wants about one equal tile per CU (256) -- or per LDS slot at 2 blocks/CU -- and the
// largest tile that still gives that many, because L2->LDS traffic falls as 1/BM + 1/BN.
static int auto_cfg(int M, int N, int K) {
  auto tiles = [&](int cfg) { return (long long)ufnd_cdiv(M, kTiles[cfg].bm) * (N / kTiles[cfg].bn); };
  if (N % 192 == 0 && tiles(8) >= 160) return 8;      // 256x192, 8 waves: BERT QKV (192 tiles) / FFN1 (256)
  if (N >= 2048) {                                     // wide N, fewer rows (ViT QKV / FFN1): 8 waves, 32-row wave tiles
    if (N % 192 == 0 && N >= 3072) return 17;          //   128x192
    if (N % 128 == 0) return 16;                       //   128x128
  }
  if (K >= 2048) {                                     // narrow N, long K (FFN2 / patch embedding)
    if (N % 96 == 0 && tiles(12) >= 400) return 12;    //   64x96 4-stage, 2 blocks/CU
    return 20;                                         //   128x64 8 waves 4-stage
  }
  if (N % 128 == 0 && tiles(16) >= 150) return 16;     // out-proj at M=4096: 128x128 8 waves
  if (N % 96 == 0) return 12;                          // small out-proj: 64x96
  return 1;                                            // 128x64 3-stage, 2 blocks/CU
}

extern "C" int ufnd_gemm_bf16_ex(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                                 float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                                 int tile_cfg, void* stream_) {
  UFND_REQUIRE(A && W && (out_bf16 || out_f32), "gemm_bf16: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "gemm_bf16: M=%d N=%d K=%d (need N%%64==0, K%%64==0)", M, N, K);
  UFND_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ufnd_aligned(A, 16) && ufnd_aligned(W, 16),
               "gemm_bf16: A/W strides must be multiples of 8 and pointers 16-B aligned");
  UFND_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= N && ufnd_aligned(residual, 16)), "gemm_bf16: residual alignment");
  UFND_REQUIRE(!out_f32 || (ldf % 4 == 0 && ldf >= N && ufnd_aligned(out_f32, 16)), "gemm_bf16: out_f32 alignment");
  UFND_REQUIRE(!out_bf16 || (ldo % 8 == 0 && ldo >= N && ufnd_aligned(out_bf16, 16)), "gemm_bf16: out_bf16 alignment");
  UFND_REQUIRE(!bias || ufnd_aligned(bias, 4), "gemm_bf16: bias alignment");
  UFND_REQUIRE(act >= 0 && act <= 2, "gemm_bf16: act=%d", act);
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0, 1};
  // tile_cfg >= 0: explicit tile; +100 / +200 select the timing-only ablations (no MFMA / no in-loop DMA)
  // automatic choice uses the explicit software-pipelined schedule (5-12 % faster on every shape swept)
  int abl = 0, sch = tile_cfg < 0 ? 1 : 0, cfg = tile_cfg < 0 ? auto_cfg(M, N, K) : tile_cfg;
  if (cfg >= 1000) { sch = 1; cfg -= 1000; }       // +1000: explicit software-pipelined schedule
  if (cfg >= 200) { abl = 2; cfg -= 200; } else if (cfg >= 100) { abl = 1; cfg -= 100; }
  UFND_REQUIRE(cfg < kNumTiles, "gemm_bf16: unknown tile config %d", cfg);
  UFND_REQUIRE(N % kTiles[cfg].bn == 0, "gemm_bf16: tile config %d needs N %% %d == 0", cfg, kTiles[cfg].bn);
  int rc = launch_cfg(cfg, abl, sch, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_gemm_bf16_splitk(const void* A, const void* W, float* slabs, int M, int N, int K, int lda, int ldw, int ksplit,
                                     int tile_cfg, void* stream_) {
  UFND_REQUIRE(A && W && slabs, "gemm_bf16_splitk: null operand");
  UFND_REQUIRE(ksplit >= 1 && ksplit <= 8 && K % (64 * ksplit) == 0, "gemm_bf16_splitk: K=%d not divisible into %d slices of 64-steps", K, ksplit);
  UFND_REQUIRE(M >= 1 && N >= 64 && N % 64 == 0, "gemm_bf16_splitk: M=%d N=%d", M, N);
  UFND_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ufnd_aligned(A, 16) && ufnd_aligned(W, 16) && ufnd_aligned(slabs, 16),
               "gemm_bf16_splitk: alignment");
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, nullptr, nullptr, nullptr, slabs, M, N, K, lda, ldw, 0, 0, N, 0, 0, 0, ksplit};
  int cfg = tile_cfg < 0 ? 17 : tile_cfg;     // 128x192, 8 waves
  if (N % kTiles[cfg < kNumTiles ? cfg : 0].bn != 0) cfg = 1;
  UFND_REQUIRE(cfg < kNumTiles && N % kTiles[cfg].bn == 0, "gemm_bf16_splitk: tile config %d does not divide N=%d", cfg, N);
  int rc = launch_cfg(cfg, 0, 1, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_gemm_bf16(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                              float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                              void* stream_) {
  return ufnd_gemm_bf16_ex(A, W, bias, residual, out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, -1, stream_);
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
