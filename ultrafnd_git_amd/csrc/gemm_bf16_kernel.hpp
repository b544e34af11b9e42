// bf16 GEMM for the encoder Linears: out = act(A W^T + bias) + residual.  (Kernel, tile table and launcher; the
// product entry points are in gemm_bf16.hip, the diagnostics ones -- timing ablations, in-kernel stamps, explicit
// experimental tiles -- in diag/gemm_diag.hip, built with UFND_DIAG into libultrafnd_hip_diag.so only.)
//
// A (M,K) and W (N,K) are both K-contiguous, which is exactly the operand shape of
// v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15]).
// Tile BM x BN x 64, WM x WN waves, wave tile (BM/WM) x (BN/WN) of 16x16 MFMA tiles.
// Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane): the LDS image is
// lane-linear, so the bank-conflict swizzle (16-B chunk ^= (row>>1)&7, conflict-free for
// ds_read_b128 of 128-B rows) is applied to each lane's SOURCE address and to the read address.
// Multi-slot LDS rings with counted vmcnt and ONE raw s_barrier per K-step, placed between the two
// k-halves of the step: the DMA of later steps stays in flight across it, and every fragment read
// and DMA issue is overlapped with the MFMAs of the other k-half.
// Epilogue: accumulators -> LDS (per wave) -> whole rows: bias, activation, fp32 residual,
// 16-B bf16 and/or 32-B fp32 stores per lane (full cache lines per row).
// Grid: one block per tile, XCD-aware (bijective) remap so that the blocks sharing an A
// row-panel land on the same XCD's L2.
#pragma once
// Experiment switch (python -m ultrafnd_git_amd.build --defs=UFND_GEMM_OUT_NT=1): forward bf16 outputs with the non-temporal hint, so that a
// streaming output does not evict the operand panels an XCD's workgroups are about to re-read.  Measured in round 4 (one box, A/B/A,
// profiles/r04_nt_stores.txt): alone, FFN1 gains 5-9 % (104.5 -> 95.0 us on 256x256 tiles, 98.1 -> 92.2 persistent); inside the encoder
// pass nothing gains and the fused QKV + attention launch, whose input is the previous layer's nt-stored output, loses 5 us (84.8 -> 89.5):
// step 1.3746 -> 1.387-1.402 ms.  Off.
#ifndef UFND_GEMM_OUT_NT
#define UFND_GEMM_OUT_NT 0
#endif

#include <type_traits>

#include "common.hpp"
#include "attn_softmax.hpp"

// No implicit contraction in this file: the epilogues spell every fused multiply-add out (fmaf), so that
// all tile shapes emit the same floating-point operation sequence (rows are batch-invariant, bit for bit).
#pragma clang fp contract(off)

#ifndef UFND_GEMM_ILV
#define UFND_GEMM_ILV 1      // 1: fragment reads interleaved with the leading MFMAs of a k-half (0: issued as one burst)
#endif

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
  unsigned long long* stamps;   // DBG builds only
  // LayerNorm extras (LNX kernels only; every pointer optional)
  const float* a_stats;  // (M, a_parts, 2) partial {sum, sumsq} of the fp32 rows A was rounded from: LayerNorm of A folded in
  const float* colsum;   // (N) sum over k of the bf16 weight row (of W * gamma)
  const float* r_stats;  // (M, r_parts, 2): the residual is LayerNorm(residual) * r_gamma + r_beta
  const float* r_gamma;
  const float* r_beta;
  float* out_stats;      // (M, N / 32, 2): partial {sum, sumsq} of the fp32 output rows, one per aligned 32 columns
  int a_parts, r_parts;
  float a_eps, r_eps, inv_h;   // inv_h = 1 / (row width the statistics are over)
  // fused attention (ATT kernels only): the tile is one sample's Q | K | V of TWO heads; see the ATT notes at the kernel
  const int32_t* att_mask;   // (M) key mask, 1 = attend (NULL: every key)
  __bf16* att_ctx;           // (M, att_h) attention output
  int att_h;                 // heads * 64 (= rows of each of the three stacked weight blocks)
  int xcd_cols;              // 1 (0): an XCD takes whole row panels; 2: the two column halves go to XCDs 0-3 / 4-7
  float att_scale_log2e;     // 1 / sqrt(64) * log2(e)
  // bf16 residual stream (LNX kernels): the residual operand is the bf16 rounding the previous GEMM already wrote for its
  // consumer (16 B per 8 columns instead of 32), and no fp32 copy of the stream exists.  Exclusive with `residual`.
  const __bf16* residual_b;
  int ldrb;
  // backward forms (BWD kernels only; gemm_bf16_bwd.hip)
  int ksplit;                // > 1: the K range is cut into ksplit slices, grid = tiles x ksplit, slice s writes its fp32 partial
  size_t slab_stride;        //      products to out_f32 + s * slab_stride (a reduce pass adds the slabs: ufnd_gemm_bf16_wgrad)
  const __bf16* aux;         // act = UFND_ACT_GELU_BWD / UFND_ACT_QUICK_GELU_BWD: out = acc * act'(aux), aux (M, ldaux) = the
  int ldaux;                 //      pre-activations the forward kept (dgrad through an activation, fused)
  // fold guard (LNX kernels with a_stats): 1,024 floats; every workgroup leaves the largest |mean| * rstd among the rows it
  // normalises in slot blockIdx.x % 1024 (one atomicMax at its very end: non-negative floats order like ints)
  float* guard;
  int pp_rows;               // persistent form (gemm_bf16_pp.hpp): row panels per tile group of the walk order
};

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

// value barrier: keeps the backend from fusing the producer of x into a later add (zero instructions)
__device__ __forceinline__ float opaque_f(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ f32x2 opaque_f2(f32x2 x) {
  asm volatile("" : "+v"(x));
  return x;
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
// a K-step are two lists of 1-KiB DMA pieces, each dealt round-robin to the waves, so every wave has
// exactly PWA + PWB pieces per K-step in flight and a counted s_waitcnt vmcnt is exact.
// Two LDS rings, STA slots of the A tile and STB slots of the W tile (STA == STB or STA == STB + 1:
// where three whole stages do not fit in 160 KiB the A operand still runs one K-step further ahead).
// Every slot is filled before the loop; the loop's ONE s_barrier per K-step sits in the MIDDLE of the
// step (see body below), where it both certifies step t+1 and frees step t's slots for refilling.
// ABL (timing experiments only, results are garbage): 1 = no MFMA/LDS reads, 2 = no DMA in the loop.
// DBG = 1: lane 0 of every block also writes s_memtime / s_memrealtime stamps (entry, first stage
// landed, main loop done, end) to a.stamps[16 * block .. ] -- diagnostics builds only.
template <int V> struct IntC { static constexpr int value = V; };

// MI = 16: v_mfma_f32_16x16x32_bf16 (one fragment read feeds 32 k of a 16-row tile);
// MI = 32: v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = 8(l>>5)+j; half as many matrix
//          instructions per K-step, so the SIMD's vector issue is held half as long).
// LNX = 1: the LayerNorm-aware epilogue (row statistics in, folded normalisation, residual through a
//          LayerNorm, row statistics out) -- see ufnd_gemm_bf16_ln.
// ATT = 1 (BM = 128, BN = 384, MI = 16, LNX = 1): the fused-QKV projection of ONE sample (its 128 tokens are the tile's
//          rows) for TWO heads, followed by their attention, in one workgroup.  Tile column c is column
//          (c / 128) * att_h + 128 * pair + c % 128 of the stacked projection, i.e. [q_h q_h' | k_h k_h' | v_h v_h'] for
//          the head pair (h, h') = (2 pair, 2 pair + 1); grid = samples x heads / 2.  The epilogue rounds the projection
//          to bf16 exactly as the stand-alone GEMM does, but into LDS images (swizzled like attention.hip's), and the
//          workgroup then runs attention.hip's per-wave schedule on them (wave w: head w / 4, queries 32 (w % 4) ..):
//          S^T = K Q^T, online softmax over two 64-key blocks, O^T = V^T P^T.  Same operations in the same order as
//          ufnd_gemm_bf16[_ln] + ufnd_attention_bf16: bit-identical ctx, without the (tokens, 3H) round trip through
//          HBM, the second launch and its three dependent memory round trips.
// waves per SIMD the register allocation must leave room for: tiles whose LDS footprint lets two workgroups share a CU
// (<= 80 KiB) only do so if two workgroups' waves also fit the register file (8-wave blocks: 128 registers per lane)
constexpr int gemm_waves_per_simd(int BM, int BN, int WM, int WN, int STA, int STB, int MI, int LNX, int ATT) {
  const int ring = (STA * BM + STB * BN) * 128, cbytes = WM * WN * MI * (BN / WN + 4) * 4;
  const int smem = (ring > cbytes ? ring : cbytes) + (LNX ? BM * 8 + 64 : 0);
  return (!ATT && smem <= 80 * 1024 ? 2 : 1) * WM * WN / 4;
}
template <int BM, int BN, int WM, int WN, int STA, int STB, int MI, int ABL = 0, int DBG = 0, int LNX = 0, int ATT = 0, int BWD = 0>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(gemm_waves_per_simd(BM, BN, WM, WN, STA, STB, MI, LNX, ATT))))
void gemm_bf16_kernel(const GemmArgs a) {
  static_assert(!BWD || (!LNX && !ATT && !ABL && !DBG), "backward forms are plain kernels");
  static_assert(MI == 16 || MI == 32, "MFMA shape");
  static_assert(!ATT || (BM == 128 && BN == 384 && MI == 16 && WM * WN == 8 && LNX == 1), "fused attention tile");
  using acc_t = typename std::conditional<MI == 16, f32x4, f32x16>::type;
  constexpr int AR = MI * MI / 64;               // accumulator registers per MFMA tile
  constexpr int KQ = MI == 16 ? 1 : 2;           // MFMA k-steps per k-half (32 k)
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;      // wave tile
  constexpr int MT = TM / MI, NT = TN / MI;      // MFMA tiles per wave
  constexpr int ASLOT = BM * 128, BSLOT = BN * 128;
  constexpr int DA = STA - 1, DB = STB - 1;      // K-steps of lookahead per operand
  // DMA wave-instructions per wave per K-step.  Where the W pieces do not divide evenly over the waves (BN = 144:
  // 18 pieces, 8 waves) every wave still issues PWB pieces and the surplus ones re-load pieces 0.. (the same bytes to
  // the same LDS address): the counted vmcnt stays uniform at the price of a few redundant 1-KiB loads.
  constexpr int APIECES = BM / 8, BPIECES = BN / 8;
  constexpr int PWA = APIECES / NW, PWB = (BPIECES + NW - 1) / NW;
  constexpr int WAITN = PWA * (DA - 1 > 0 ? DA - 1 : 0) + PWB * (DB - 1 > 0 ? DB - 1 : 0);
  constexpr int CP = TN + 4;                     // fp32 C-staging pitch (floats), MI rows per wave
  constexpr int CBYTES = NW * MI * CP * 4;
  constexpr int RING = STA * ASLOT + STB * BSLOT;
  // ATT: six 16-KiB images (Q, K, V of two heads; 128 tokens x 128 B) + 128 key biases, behind the C staging patches
  // (they overlay the operand ring, which is dead by then)
  constexpr int ATT_OFF = (CBYTES + 1023) & ~1023;
  constexpr int ATT_END = ATT ? ATT_OFF + 6 * 16384 + 512 : 0;
  constexpr int STAT_OFF0 = (RING > CBYTES) ? RING : CBYTES;
  constexpr int STAT_OFF = STAT_OFF0 > ATT_END ? STAT_OFF0 : ATT_END;  // LNX: {mean, rstd} per tile row, behind the ring (and the images)
  constexpr int SMEM = STAT_OFF + (LNX ? BM * 8 + 64 : 0);      // (+ 16 floats: the waves' guard maxima)
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(APIECES % NW == 0 && TM % MI == 0 && TN % MI == 0, "tile split");
  static_assert(STB >= 2 && (STA == STB || STA == STB + 1), "ring depths");
  static_assert(WAITN <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  unsigned long long stamp[10];
  if constexpr (DBG) {
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
  }

  // XCD-aware bijective remap of the block id
  const int nblk = a.m_tiles * a.n_tiles;
  int bid, kslice = 0;
  {
    const int ngrid = BWD ? nblk * (a.ksplit > 1 ? a.ksplit : 1) : nblk;
    const int q = ngrid >> 3, r = ngrid & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if constexpr (BWD) {       // slice-major: an XCD's contiguous share of the grid is mostly one K slice (shared operand range)
      kslice = bid / nblk;
      bid -= kslice * nblk;
    }
  }
  // tile (tm, tn) of this workgroup.  Default: row-major, so an XCD's contiguous share of the grid is a set of whole row
  // panels (A fetched by one XCD, every weight row by all eight: fabric reads = A + 8 W).  xcd_cols = 2 (wide shapes:
  // W is the larger operand per launch): the grid is enumerated column-half by column-half, so XCDs 0-3 hold the left half
  // of the columns and XCDs 4-7 the right half (reads = 2 A + 4 W).
  int tm = bid / a.n_tiles, tn = bid % a.n_tiles;
  if (a.xcd_cols == 2) {
    const int nh = a.n_tiles >> 1, per_half = a.m_tiles * nh;
    const int half = bid / per_half, rem = bid - half * per_half;
    tm = rem / nh;
    tn = half * nh + (rem - tm * nh);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN, fr = lane & (MI - 1), g = lane / MI;   // fragment row, k-group

  acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

  int nk = a.K / BK, kbase = 0;               // K-steps (of this block's K slice)
  if constexpr (BWD) {
    if (a.ksplit > 1) {
      const int per = (nk + a.ksplit - 1) / a.ksplit;
      kbase = kslice * per * BK;
      nk = nk - kslice * per < per ? nk - kslice * per : per;
    }
  }
  // per-lane source rows of my pieces (fixed over the K loop): only the column offset moves
  const int prow = lane >> 3, ppos = lane & 7;
  auto issueA = [&](int t) {
    char* buf = smem + (t % STA) * ASLOT;
    const int k0 = kbase + t * BK;
#pragma unroll
    for (int ii = 0; ii < PWA; ++ii) {
      const int p = wave + NW * ii;                  // piece id: rows 8p..8p+7 of the A tile
      const int r = 8 * p + prow;
      const int c = ppos ^ ((r >> 1) & 7);
      int gr = m0 + r;
      gr = gr < a.M ? gr : a.M - 1;
      dma16(a.A + (size_t)gr * a.lda + k0 + c * 8, buf + p * 1024);
    }
  };
  auto issueB = [&](int t) {
    char* buf = smem + STA * ASLOT + (t % STB) * BSLOT;
    const int k0 = kbase + t * BK;
#pragma unroll
    for (int ii = 0; ii < PWB; ++ii) {
      int p = wave + NW * ii;
      if constexpr (BPIECES % NW != 0) p = p < BPIECES ? p : p - BPIECES;
      const int r = 8 * p + prow;
      const int c = ppos ^ ((r >> 1) & 7);
      int gr = n0 + r;
      gr = gr < a.N ? gr : a.N - 1;
      if constexpr (ATT) gr = (r >> 7) * a.att_h + tn * 128 + (r & 127);     // [q | k | v] rows of this head pair
      dma16(a.W + (size_t)gr * a.ldw + k0 + c * 8, buf + p * 1024);
    }
  };
  // LNX: the producers' partial {sum, sumsq} of my tile's rows become {mean, rstd} in LDS.  The loads are
  // the kernel's FIRST memory operations (older than every DMA piece, so the counted vmcnt waits of the
  // K loop are not disturbed) and are consumed while the first stages fly -- their latency hides behind
  // the wait for stage 0, which every block pays anyway.  The summation order is CANONICAL -- 16-B chunk c (two
  // partials) belongs to group c % 4, a group adds its chunks in ascending order, the total is
  // (g0 + g1) + (g2 + g3) -- whatever the tile shape, so a row's statistics (hence its outputs) do
  // not depend on the batch it is computed in.  No atomics.
  constexpr int TPR = NW * 64 / BM;                 // threads per tile row: 2 or 4
  static_assert(!LNX || TPR == 2 || TPR == 4, "threads per tile row");
  constexpr int GPT = 4 / (TPR < 4 ? TPR : 4);      // groups per thread
  f32x4 sv[GPT][3];
  f32x2* st_lds = reinterpret_cast<f32x2*>(smem + STAT_OFF);
  const float* sp = nullptr;
  if constexpr (LNX) {
    sp = a.a_stats ? a.a_stats : a.r_stats;
    if (sp) {
      const int parts = a.a_stats ? a.a_parts : a.r_parts;
      int row = m0 + (int)threadIdx.x / TPR;
      row = row < a.M ? row : a.M - 1;
      const f32x4* base = reinterpret_cast<const f32x4*>(sp + (size_t)row * parts * 2);
      const int nq = parts >> 1, sub = threadIdx.x % TPR;
#pragma unroll
      for (int gi = 0; gi < GPT; ++gi)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int q = (sub + gi * TPR) + 4 * u;     // chunks of group sub + gi*TPR
          // unconditional (clamped) loads, issued back to back and masked at use.  Inline asm: the compiler's
          // own vmcnt bookkeeping would put s_waitcnt vmcnt(0) in front of their first use, i.e. also wait for
          // every prologue DMA stage; stats_wait() below waits for exactly these loads.
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sv[gi][u]) : "v"(base + (q < nq ? q : 0)) : "memory");
        }
    }
  }
  int att_mk = 1;
  if constexpr (ATT) {      // this sample's key mask: requested now, used after the K loop (older than every DMA piece)
    if (a.att_mask && threadIdx.x < BM) att_mk = a.att_mask[m0 + threadIdx.x];
  }
  // prologue: every ring slot is filled (W(s) before A(s), step by step, so that a counted vmcnt
  // separates "steps <= t+1" from the later ones)
#pragma unroll
  for (int s = 0; s < STA; ++s) {
    if (s < STB && s < nk) issueB(s);
    if (s < nk) issueA(s);
  }

  if constexpr (LNX) {
    if (sp) {
      // the statistics loads are older than every prologue DMA piece: wait until at most those pieces are
      // outstanding (the loaded registers are operands, so nothing that reads them can move above the wait)
      constexpr int NPRO = STA * PWA + STB * PWB;
      static_assert(NPRO <= 63, "vmcnt range");
      if (STA <= nk) {
        if constexpr (GPT == 1)
          asm volatile("s_waitcnt vmcnt(%3)" : "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]) : "n"(NPRO) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%6)" : "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]), "+v"(sv[GPT - 1][0]), "+v"(sv[GPT - 1][1]),
                       "+v"(sv[GPT - 1][2]) : "n"(NPRO) : "memory");
      } else {
        if constexpr (GPT == 1)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]) : : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]), "+v"(sv[GPT - 1][0]), "+v"(sv[GPT - 1][1]),
                       "+v"(sv[GPT - 1][2]) : : "memory");
      }
      const int nq_ = (a.a_stats ? a.a_parts : a.r_parts) >> 1, sub_ = threadIdx.x % TPR;
      float gs[GPT], gq[GPT];
#pragma unroll
      for (int gi = 0; gi < GPT; ++gi) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if ((sub_ + gi * TPR) + 4 * u >= nq_) sv[gi][u] = f32x4{0.f, 0.f, 0.f, 0.f};   // chunk beyond the partial count
        }
        gs[gi] = sv[gi][0][0] + sv[gi][0][2];
        gq[gi] = sv[gi][0][1] + sv[gi][0][3];
#pragma unroll
        for (int u = 1; u < 3; ++u) { gs[gi] += sv[gi][u][0] + sv[gi][u][2]; gq[gi] += sv[gi][u][1] + sv[gi][u][3]; }
      }
      float sm, sq;
      if constexpr (TPR == 4) {        // lane sub holds group sub
        sm = gs[0] + quad_xor1(gs[0]);
        sq = gq[0] + quad_xor1(gq[0]);
        sm += quad_xor2(sm);
        sq += quad_xor2(sq);
      } else {                         // lane sub holds groups sub and sub + 2
        const float s01 = gs[0] + quad_xor1(gs[0]), s23 = gs[1] + quad_xor1(gs[1]);
        const float q01 = gq[0] + quad_xor1(gq[0]), q23 = gq[1] + quad_xor1(gq[1]);
        sm = s01 + s23;
        sq = q01 + q23;
      }
      const float mean = __fmul_rn(sm, a.inv_h);
      const float var = fmaxf(__fmaf_rn(-mean, mean, __fmul_rn(sq, a.inv_h)), 0.f);
      const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(var, a.a_stats ? a.a_eps : a.r_eps));   // v_rsq_f32 (1 ulp, the same instruction in every tile shape)
      if (threadIdx.x % TPR == 0) st_lds[threadIdx.x / TPR] = f32x2{mean, rstd};
      if (a.guard && a.a_stats) {      // fold guard: the largest |mean| / std among the rows this workgroup folds (reported at the kernel's end)
        float ratio = __fmul_rn(fabsf(mean), rstd);
        ratio = ratio == ratio ? ratio : INFINITY;      // (a NaN statistic must trip the guard: fmaxf would drop it)
        const float worst = wave_max(threadIdx.x % TPR == 0 ? ratio : 0.0f);
        if (lane == 0) reinterpret_cast<float*>(smem + STAT_OFF + BM * 8)[wave] = worst;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  bf16x8 af0[KQ][MT], bf0[KQ][NT], af1[KQ][MT], bf1[KQ][NT];      // fragments of k-half 0 / k-half 1
  auto read_half = [&](int t, auto kk_, bf16x8 (&af)[KQ][MT], bf16x8 (&bfr)[KQ][NT]) {
    constexpr int kk = decltype(kk_)::value;
    const char* At = smem + (t % STA) * ASLOT;
    const char* Bt = smem + STA * ASLOT + (t % STB) * BSLOT;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int chunk = MI == 16 ? g + 4 * kk : g + 2 * q + 4 * kk;     // 16-B chunk of the 128-B row
#pragma unroll
      for (int i = 0; i < MT; ++i) af[q][i] = lds_frag(At, wm * TM + i * MI + fr, chunk);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[q][j] = lds_frag(Bt, wn * TN + j * MI + fr, chunk);
    }
  };
  auto mma_half = [&](const bf16x8 (&af)[KQ][MT], const bf16x8 (&bfr)[KQ][NT]) {
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (MI == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q][i], bfr[q][j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][i], bfr[q][j], acc[i][j], 0, 0, 0);
        }
  };
  // step t (its k-half-0 fragments are already in registers):
  //   k-half-1 fragment reads || MFMAs of k-half 0
  //   wait: my pieces of step t+1 landed, my reads of step t retired  ->  ONE s_barrier (step t+1 is
  //   complete in LDS; nobody reads step t's slots any more)
  //   refill step t's slots with W(t+STB), A(t+STA); k-half-0 reads of step t+1 || MFMAs of k-half 1
  // so every LDS read and every DMA issue sits beside MFMAs of the other half, and the barrier has
  // half a step of queued matrix work on either side.  One MFMA leads each half so that the wait the
  // compiler places in front of it covers only reads issued half a step earlier.
  constexpr int NMF = KQ * MT * NT, NRD = KQ * (MT + NT);
  constexpr bool ILV = UFND_GEMM_ILV != 0;
  auto body = [&](int t, auto ia_, auto ib_, auto next_) {
    constexpr bool IA = decltype(ia_)::value != 0, IB = decltype(ib_)::value != 0, NEXT = decltype(next_)::value != 0;
    if constexpr (ABL != 1) {
      read_half(t, IntC<1>{}, af1, bf1);
      mma_half(af0, bf0);
      if constexpr (ILV) {            // one fragment read behind each of the first MFMAs: the matrix pipe is never left
#pragma unroll                        // waiting behind a burst of NRD LDS instructions of both waves of the SIMD
        for (int q = 0; q < (NRD < NMF ? NRD : NMF - 1); ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NRD - (NRD < NMF ? NRD : NMF - 1), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - (NRD < NMF ? NRD : NMF - 1), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
      }
    }
    if constexpr (NEXT) {
      __builtin_amdgcn_sched_barrier(0);
      if (t + STA - 1 < nk) wait_vmcnt<WAITN>();    // steps t+2 .. may stay in flight (all exist while t+STA-1 < nk)
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // (reads first in program order: the DMA writes other slots, but the compiler cannot know and
      //  would keep every LDS read behind the last DMA)
      if constexpr (ABL != 1) read_half(t + 1, IntC<0>{}, af0, bf0);
      if constexpr (ABL != 2) {
        if constexpr (IB) issueB(t + STB);
        if constexpr (IA) issueA(t + STA);
      }
    }
    if constexpr (ABL != 1) {
      mma_half(af1, bf1);
      constexpr int PW = (ABL == 2 || !NEXT) ? 0 : (IA ? PWA : 0) + (IB ? PWB : 0);
      constexpr int GRP = (NMF - 1) / (PW + 1) > 0 ? (NMF - 1) / (PW + 1) : 1;
      if constexpr (ILV && NEXT && NRD + PW < NMF) {
#pragma unroll
        for (int q = 0; q < NRD; ++q) {           // {MFMA, fragment read} x NRD
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        constexpr int REST = NMF - NRD, G2 = REST / (PW + 1) > 0 ? REST / (PW + 1) : 1;
#pragma unroll
        for (int q = 0; q < PW; ++q) {            // {G2 MFMA, DMA piece} x PW
          __builtin_amdgcn_sched_group_barrier(0x008, G2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, REST - G2 * PW > 0 ? REST - G2 * PW : 0, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
        for (int q = 0; q < PW; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, GRP, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1 - GRP * PW > 0 ? NMF - 1 - GRP * PW : 0, 0);
      }
    }
  };

  // stage 0 landed (everything issued after it may stay in flight when it all exists)
  if (STA <= nk) wait_vmcnt<PWA * (STA - 1) + PWB * (STB - 1)>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG) {
    stamp[2] = __builtin_amdgcn_s_memtime();
    stamp[3] = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (ABL != 1) read_half(0, IntC<0>{}, af0, bf0);
  int t = 0;
  for (; t + STA < nk; ++t) body(t, IntC<1>{}, IntC<1>{}, IntC<1>{});     // steady state
  for (; t + STB < nk; ++t) body(t, IntC<0>{}, IntC<1>{}, IntC<1>{});     // (STA == STB + 1) only W left to fetch
  for (; t + 1 < nk; ++t) body(t, IntC<0>{}, IntC<0>{}, IntC<1>{});       // drain
  body(t, IntC<0>{}, IntC<0>{}, IntC<0>{});                               // last step: no successor
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();   // everyone is done reading the ring: reuse it for the epilogue
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG) {
    stamp[4] = __builtin_amdgcn_s_memtime();
    stamp[5] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- epilogue.  Bias and activation are elementwise, so they run on the accumulators where they
  // lie (straight-line VALU, no LDS dependency); only the layout change for whole-row stores goes
  // through a wave-private LDS patch, 16 rows at a time, where the fp32 residual is added.
  // per-column vectors of this lane's accumulator columns (col_j = n0 + wn*TN + j*MI + fr), loaded once
  float bj[NT], csj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { bj[j] = 0.f; csj[j] = 0.f; }
  // global column of tile column c
  auto gcol = [&](int c) { return ATT ? (c >> 7) * a.att_h + tn * 128 + (c & 127) : n0 + c; };
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j) bj[j] = a.bias[gcol(wn * TN + j * MI + fr)];
  }
  bool fold = false, rln = false;
  if constexpr (LNX) {
    fold = a.a_stats != nullptr;
    rln = a.r_stats != nullptr;
    if (fold) {
#pragma unroll
      for (int j = 0; j < NT; ++j) csj[j] = a.colsum[gcol(wn * TN + j * MI + fr)];
    }
  }
  float* cst = reinterpret_cast<float*>(smem) + wave * MI * CP;
  float* const outf = (BWD && a.out_f32) ? a.out_f32 + (size_t)kslice * a.slab_stride : a.out_f32;
  constexpr int CPR = TN / 8;                    // 8-column chunks per row
  constexpr int CHUNKS = MI * CPR;               // chunks per MI-row patch
  constexpr int NIT = (CHUNKS + 63) / 64;        // row-phase iterations per patch
  constexpr bool FIXCOL = (64 % CPR) == 0;       // a lane keeps its 8 columns across the iterations
  auto ld8 = [&](const float* ptr, float (&o)[8]) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(ptr), x1 = *reinterpret_cast<const f32x4*>(ptr + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { o[q] = x0[q]; o[4 + q] = x1[q]; }
  };
  float g8[8], b8[8];                            // LNX, FIXCOL: gamma / beta of the residual's LayerNorm for my columns
  if constexpr (LNX && FIXCOL) {
    if (rln) {
      const int col = n0 + wn * TN + (lane % CPR) * 8;
      ld8(a.r_gamma + col, g8);
      ld8(a.r_beta + col, b8);
    }
  }
  // MODE (LNX kernels): 0 plain, 1 LayerNorm of the A operand folded in, 2 residual through a LayerNorm
  auto epilogue = [&](auto act_, auto mode_) {
    constexpr int ACT = decltype(act_)::value;
    constexpr bool FOLD = LNX && decltype(mode_)::value == 1, RLN = LNX && decltype(mode_)::value == 2;
    // A folded-LayerNorm call (QKV, FFN1) carries no residual and produces no row statistics (the host entry refuses both):
    // its epilogue is compiled without them -- with the 128 accumulators of a 256x256 tile alive, the residual registers of
    // a path never taken were what pushed that kernel into scratch (56 VGPRs spilled, +44 MB read and written per FFN1 launch).
    constexpr bool RES = !FOLD;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // the residual rows of this patch are requested first: they land while the patch is staged.  (Requesting them one
      // patch ahead into a second register set changes nothing -- 15.9 us either way for a 256x192 tile at 16,384 rows:
      // that epilogue moves 490 KB per CU on all 256 CUs at once = 7.8 TB/s, it is bound by the fabric, not by latency.)
      float rr8[RES ? NIT : 1][8];
      if constexpr (RES) if (a.residual && !(BWD && a.aux)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int id = lane + 64 * it;
          if (id < CHUNKS) {
            int row = m0 + wm * TM + i * MI + id / CPR;
            row = row < a.M ? row : a.M - 1;
            ld8(a.residual + (size_t)row * a.ldr + n0 + wn * TN + (id % CPR) * 8, rr8[it]);
          }
        }
      }
      if constexpr (BWD) {      // pre-activations of the fused activation backward: requested like a residual, used as a factor
        if (a.aux) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int id = lane + 64 * it;
            if (id < CHUNKS) {
              int row = m0 + wm * TM + i * MI + id / CPR;
              row = row < a.M ? row : a.M - 1;
              const bf16x8 xb = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)row * a.ldaux + n0 + wn * TN + (id % CPR) * 8);
#pragma unroll
              for (int q = 0; q < 8; ++q) rr8[it][q] = (float)xb[q];
            }
          }
        }
      }
      if constexpr (LNX && RES) {
        if (a.residual_b) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int id = lane + 64 * it;
            if (id < CHUNKS) {
              int row = m0 + wm * TM + i * MI + id / CPR;
              row = row < a.M ? row : a.M - 1;
              const bf16x8 rb = *reinterpret_cast<const bf16x8*>(a.residual_b + (size_t)row * a.ldrb + n0 + wn * TN + (id % CPR) * 8);
#pragma unroll
              for (int q = 0; q < 8; ++q) rr8[it][q] = (float)rb[q];
            }
          }
        }
      }
      // register phase: [folded LayerNorm of the A operand,] bias, activation on the accumulators where they lie, two rows
      // (adjacent accumulator registers) per packed fp32 operation
#pragma unroll
      for (int r = 0; r < AR; r += 2) {
        // accumulator register r of lane (fr, g): 16x16 -> row 4g + r; 32x32 -> row 8(r>>2) + 4g + (r&3); r + 1 is the next row
        const int prow_ = MI == 16 ? 4 * g + r : 8 * (r >> 2) + 4 * g + (r & 3);
        f32x2 mean2 = {0.f, 0.f}, rstd2 = {1.f, 1.f};
        if constexpr (FOLD) {      // {mean, rstd} of the two rows
          const f32x2 a0 = st_lds[wm * TM + i * MI + prow_], a1 = st_lds[wm * TM + i * MI + prow_ + 1];
          mean2 = f32x2{a0[0], a1[0]};
          rstd2 = f32x2{a0[1], a1[1]};
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x2 v = {acc[i][j][r], acc[i][j][r + 1]};
          // LayerNorm(A) W'^T = rstd (A W'^T - mean colsum(W')); explicit roundings: every tile shape must
          // emit the same operation sequence (rows stay batch-invariant)
          // (opaque(): the product must not be re-fused with the bias add below -- the backend did so
          //  for a few elements of some tile shapes even with contraction switched off in the source)
          if constexpr (FOLD) v = opaque_f2(rstd2 * __builtin_elementwise_fma(-mean2, f32x2{csj[j], csj[j]}, v));
          v = v + f32x2{bj[j], bj[j]};
          if constexpr (ACT == UFND_ACT_GELU) v = gelu_fast_f2(v);
          else if constexpr (ACT == UFND_ACT_QUICK_GELU) v = quick_gelu_fast_f2(v);
          cst[prow_ * CP + j * MI + fr] = v.x;
          cst[(prow_ + 1) * CP + j * MI + fr] = v.y;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // row phase: whole rows out of the wave's patch: residual, statistics, 16-B / 32-B stores
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = lane + 64 * it;
        if (id >= CHUNKS) break;
        const int rr = id / CPR, cl = (id % CPR) * 8;
        const int row = m0 + wm * TM + i * MI + rr;
        const int col = n0 + wn * TN + cl;
        const bool live = row < a.M;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(cst + rr * CP + cl + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if constexpr (BWD) {
          if (a.aux) {
            if (a.act == UFND_ACT_GELU_BWD) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = v[q] * gelu_grad_f(rr8[it][q]);
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = v[q] * quick_gelu_grad_f(rr8[it][q]);
            }
          }
        }
        if constexpr (RES) if ((a.residual && !(BWD && a.aux)) || (LNX && a.residual_b)) {
          if constexpr (RLN) {      // the residual stream is LayerNorm(residual) * gamma + beta, never materialised
            const f32x2 ms = st_lds[wm * TM + i * MI + rr];
            if constexpr (!FIXCOL) {
              ld8(a.r_gamma + col, g8);
              ld8(a.r_beta + col, b8);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) rr8[it][q] = __fmaf_rn(__fmul_rn(__fsub_rn(rr8[it][q], ms[0]), ms[1]), g8[q], b8[q]);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = __fadd_rn(v[q], rr8[it][q]);
        }
        if constexpr (LNX && RES) {
          if (a.out_stats) {      // partial {sum, sumsq} of the fp32 output row over each aligned 32-column group
            if constexpr (CPR % 4 == 0) {   // (canonical: the partial of columns [32p, 32p+32) never depends on the tile shape)
              float sm = 0.f, sq = 0.f;
#pragma unroll
              for (int q = 0; q < 8; ++q) { sm = __fadd_rn(sm, v[q]); sq = __fmaf_rn(v[q], v[q], sq); }
              sm += quad_xor1(sm);
              sq += quad_xor1(sq);
              sm += quad_xor2(sm);
              sq += quad_xor2(sq);
              if (live && (id & 3) == 0)
                *reinterpret_cast<f32x2*>(a.out_stats + ((size_t)row * (a.N >> 5) + (col >> 5)) * 2) = f32x2{sm, sq};
            }
          }
        }
        if constexpr (ATT) {        // bf16 rounding of the projection (as the stand-alone GEMM stores it) into the LDS image of
          const int ct = wn * TN + cl;                     // its (q|k|v, head): 128-B rows, 16-B chunks swizzled as attention.hip reads them
          const int trow = wm * TM + i * MI + rr, img = ct >> 6, ch = (ct & 63) >> 3;
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (__bf16)v[q];
          const int sw = img >= 4 ? (ch ^ (((trow >> 1) & 3) << 1)) : (ch ^ ((trow >> 1) & 7));      // V images: the transposed-read swizzle
          *reinterpret_cast<bf16x8*>(smem + ATT_OFF + img * 16384 + trow * 128 + (sw << 4)) = o;
          continue;
        }
        if (!live) continue;
        if (a.out_f32) {
          float* op = outf + (size_t)row * a.ldf + col;
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (a.out_bf16) {
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (__bf16)v[q];
          if constexpr (UFND_GEMM_OUT_NT && !BWD) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(a.out_bf16 + (size_t)row * a.ldo + col));
          else *reinterpret_cast<bf16x8*>(a.out_bf16 + (size_t)row * a.ldo + col) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before the next row-tile overwrites it
    }
  };
  if constexpr (LNX) {
    if (fold) {
      if (a.act == UFND_ACT_GELU) epilogue(IntC<UFND_ACT_GELU>{}, IntC<1>{});
      else if (a.act == UFND_ACT_QUICK_GELU) epilogue(IntC<UFND_ACT_QUICK_GELU>{}, IntC<1>{});
      else epilogue(IntC<UFND_ACT_NONE>{}, IntC<1>{});
    } else if (rln) {
      epilogue(IntC<UFND_ACT_NONE>{}, IntC<2>{});      // (host side: residual-through-LayerNorm calls carry no activation)
    } else {
      epilogue(IntC<UFND_ACT_NONE>{}, IntC<0>{});      // (and neither do plain calls of this entry point)
    }
  } else {
    if (a.act == UFND_ACT_GELU) epilogue(IntC<UFND_ACT_GELU>{}, IntC<0>{});
    else if (a.act == UFND_ACT_QUICK_GELU) epilogue(IntC<UFND_ACT_QUICK_GELU>{}, IntC<0>{});
    else epilogue(IntC<UFND_ACT_NONE>{}, IntC<0>{});
  }
  if constexpr (ATT && DBG) {
    stamp[8] = __builtin_amdgcn_s_memtime();       // projection epilogue done (images written), attention starts
    stamp[9] = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (ATT) {
    // ---- attention of the tile's two heads on the LDS images (attention.hip's per-wave schedule, L = 128 = two key blocks)
    constexpr float NEG_MASK = -3.0e38f;
    float* kbias = reinterpret_cast<float*>(smem + ATT_OFF + 6 * 16384);
    if (threadIdx.x < BM) kbias[threadIdx.x] = att_mk != 0 ? 0.0f : NEG_MASK;
    __syncthreads();
    const int hh = wave >> 2, wq = wave & 3, g4 = lane >> 4, f16 = lane & 15;
    const char* qimg = smem + ATT_OFF + hh * 16384;
    const char* kimg = smem + ATT_OFF + (2 + hh) * 16384;
    const char* vimg = smem + ATT_OFF + (4 + hh) * 16384;
    bf16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = wq * 32 + qt * 16 + f16;
      qf[qt][0] = lds_frag(qimg, q, g4);
      qf[qt][1] = lds_frag(qimg, q, g4 + 4);
    }
    f32x4 o[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    constexpr int KB = 64, KT = KB / 16;
#pragma unroll
    for (int kb0 = 0; kb0 < BM; kb0 += KB) {
      const char* ks = kimg + kb0 * 128;
      const char* vs = vimg + kb0 * 128;
      if (masked_block_is_noop(kbias + kb0, lane, m_run)) continue;      // (as attention.hip: bit-identical, see attn_softmax.hpp)
      f32x4 sc[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const bf16x8 kf = lds_frag(ks, kt * 16 + f16, g4 + 4 * kk);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][kk], sc[kt][qt], 0, 0, 0);
        }
      const bool any_masked = __any(kbias[kb0 + lane] != 0.0f);      // (64 bias words of this key block: one per lane; wave-uniform)
      bf16x8 pf[KT / 2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float alpha = online_softmax_block<KT>(sc, qt, kbias + kb0, any_masked, g4, a.att_scale_log2e, m_run[qt], l_run[qt], pf);
        if (!__all(alpha == 1.0f)) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
        }
      }
#pragma unroll
      for (int ksd = 0; ksd < KT / 2; ++ksd) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int qq = f16 >> 2, pp = f16 & 3;
          const int key0 = 32 * ksd + 4 * g4 + qq;
          const int ch = 2 * dt + (pp >> 1);
          const int off0 = key0 * 128 + ((ch ^ (((key0 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
          const int key1 = key0 + 16;
          const int off1 = key1 * 128 + ((ch ^ (((key1 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
          const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs + off0));
          const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vs + off1));
          union { s16x4 s2[2]; bf16x8 v; } u;
          u.s2[0] = t0;
          u.s2[1] = t1;
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[ksd][qt], o[dt][qt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      const int q = wq * 32 + qt * 16 + f16;
      __bf16* dst = a.att_ctx + (size_t)(m0 + q) * a.att_h + (tn * 2 + hh) * 64 + 4 * g4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(__bf16)(o[dt][qt][0] * inv), (__bf16)(o[dt][qt][1] * inv), (__bf16)(o[dt][qt][2] * inv), (__bf16)(o[dt][qt][3] * inv)};
        *reinterpret_cast<bf16x4*>(dst + dt * 16) = ov;
      }
    }
  }
  if constexpr (LNX) {
    if (a.guard && a.a_stats && threadIdx.x == 0) {      // (the waves' maxima were written in front of the K loop's first barrier)
      const float* gw = reinterpret_cast<const float*>(smem + STAT_OFF + BM * 8);
      float worst = gw[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) worst = fmaxf(worst, gw[w]);
      atomicMax(reinterpret_cast<int*>(a.guard + (blockIdx.x & 1023)), __float_as_int(worst));
    }
  }
  if constexpr (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[6] = __builtin_amdgcn_s_memtime();
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      constexpr int NS = ATT ? 10 : 8;
#pragma unroll
      for (int i = 0; i < NS; ++i) a.stamps[(size_t)blockIdx.x * NS + i] = stamp[i];
    }
  }
}

}  // namespace

// tile configurations (BM x BN, waves WM x WN, LDS ring slots of A / W).  Exact-fit widths exist
// because a 256-CU chip wants ~256 (or 512 at 2 blocks/CU) equal tiles per launch, not "as many as
// it takes".  X(id, BM, BN, WM, WN, STA, STB, MI, LN, PROD)
//   LN   = 1: the LayerNorm-aware kernel exists too;
//   PROD = 1: the tile is part of libultrafnd_hip.so (every such tile is selectable through ufnd_gemm_bf16_ex and
//             covered by tests/test_gpu_tier_b.py::test_gemm_bf16_every_exported_tile); PROD = 0 tiles are sweep
//             material and exist in the diagnostics library only.
#define UFND_GEMM_TILES(X)                                                                             \
  X(0, 128, 128, 2, 2, 3, 3, 16, 0, 0)   /*  96 KiB */                                                           \
  X(1, 128, 64, 2, 2, 3, 3, 16, 0, 0)    /*  72 KiB (2 blocks/CU) */                                             \
  X(2, 256, 128, 4, 2, 3, 3, 16, 1, 1)   /* 144 KiB */                                                           \
  X(3, 128, 128, 2, 2, 2, 2, 16, 1, 0)   /*  64 KiB (2 blocks/CU) */                                             \
  X(4, 128, 64, 2, 2, 4, 4, 16, 0, 0)    /*  96 KiB */                                                           \
  X(5, 256, 64, 4, 2, 3, 3, 16, 0, 0)    /* 120 KiB */                                                           \
  X(6, 128, 288, 2, 2, 3, 3, 16, 0, 0)   /* 156 KiB (N=2304 -> 8 column tiles) */                                \
  X(7, 128, 96, 2, 2, 3, 3, 16, 0, 0)    /*  84 KiB (N=768 -> 8 column tiles) */                                 \
  X(8, 256, 192, 4, 2, 2, 2, 16, 1, 1)   /* 112 KiB (N=3072 -> 16 column tiles) */                               \
  X(9, 128, 384, 2, 2, 2, 2, 16, 0, 0)   /* 128 KiB */                                                           \
  X(10, 128, 192, 2, 2, 3, 3, 16, 0, 0)  /* 120 KiB */                                                           \
  X(11, 128, 96, 2, 2, 4, 4, 16, 0, 0)   /* 112 KiB */                                                           \
  X(12, 64, 96, 1, 2, 4, 4, 16, 0, 0)    /*  80 KiB (2 blocks/CU; ViT M=1600 -> 25 row tiles) */                 \
  X(13, 64, 192, 1, 2, 3, 3, 16, 0, 0)   /*  96 KiB */                                                           \
  X(14, 128, 256, 2, 2, 2, 2, 16, 0, 0)  /*  96 KiB */                                                           \
  X(15, 256, 256, 4, 2, 2, 2, 16, 1, 1)  /* 128 KiB */                                                           \
  X(16, 128, 128, 4, 2, 3, 3, 16, 1, 1)  /*  96 KiB (8 waves, wave tile 32x64) */                                \
  X(17, 128, 192, 4, 2, 3, 3, 16, 1, 1)  /* 120 KiB (8 waves, wave tile 32x96) */                                \
  X(18, 256, 64, 4, 2, 4, 4, 16, 0, 0)   /* 160 KiB */                                                           \
  X(19, 64, 64, 1, 2, 4, 4, 16, 0, 0)    /*  64 KiB (2 blocks/CU) */                                             \
  X(20, 128, 64, 4, 2, 4, 4, 16, 1, 1)   /*  96 KiB (8 waves, wave tile 32x32) */                                \
  X(21, 256, 192, 2, 4, 2, 2, 16, 0, 0)  /* 112 KiB (wave tile 128x48) */                                        \
  X(22, 256, 192, 4, 2, 3, 2, 16, 1, 1)  /* 144 KiB: A two K-steps ahead, W one */                               \
  X(23, 256, 256, 4, 2, 3, 2, 16, 0, 0)  /* 160 KiB */                                                           \
  X(24, 128, 192, 4, 2, 4, 4, 16, 0, 0)  /* 160 KiB */                                                           \
  X(25, 128, 256, 4, 2, 3, 3, 16, 0, 0)  /* 144 KiB (8 waves, wave tile 32x128) */                               \
  X(26, 256, 192, 4, 2, 2, 2, 32, 0, 0)  /* 32x32x16 MFMA form (measured 2-4 % slower than 16x16x32 on every shape) */ \
  X(27, 128, 128, 4, 2, 3, 3, 32, 0, 0)                                                                    \
  X(28, 256, 144, 8, 1, 2, 2, 16, 1, 1)  /* 100 KiB: N=2304 -> 16 x 16 = 256 tiles (wave tile 32x144; W pieces dealt unevenly) */ \
  X(29, 128, 128, 4, 2, 2, 2, 16, 1, 0)  /*  64 KiB, 8 waves (wave tile 32x64, 110 registers): 2 blocks/CU (measured: no gain) */

#ifdef UFND_DIAG
#define UFND_TILE_BUILT(PROD_) 1
#else
#define UFND_TILE_BUILT(PROD_) (PROD_)
#endif

struct TileCfg { int bm, bn, threads, wn, lnx, built; };
static const TileCfg kTiles[] = {
#define X(id, BM_, BN_, WM_, WN_, SA_, SB_, MI_, LN_, PROD_) {BM_, BN_, WM_ * WN_ * 64, WN_, LN_, UFND_TILE_BUILT(PROD_)},
    UFND_GEMM_TILES(X)
#undef X
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

// Column split of the grid over the XCDs (see the kernel): worth it where the weight operand (N x K) outweighs the
// activation operand (M x K) per launch AND both halves still give every XCD whole row panels.
static int xcd_cols_for(int M, int N, int K, int m_tiles, int n_tiles) {
#ifdef UFND_DIAG
  static const int forced = [] { const char* e = getenv("UFND_GEMM_XCD_COLS"); return e ? atoi(e) : 0; }();
  if (forced == 1) return 1;
  if (forced == 2) return (n_tiles % 2 == 0) ? 2 : 1;
#endif
  (void)K;
  // reads: A + 8 W (row panels) against 2 A + 4 W (column halves): the split pays when A < 4 W, i.e. M < 4 N
  return (4LL * N > (long long)M && n_tiles % 2 == 0 && m_tiles >= 4) ? 2 : 1;
}

// mode: 0 plain kernel, 4 LayerNorm-aware kernel (tiles with LN = 1).  Diagnostics build only: 1 / 2 timing
// ablations (no MFMA / no in-loop DMA; results are garbage), 3 stamps build, 5 LayerNorm-aware stamps build.
static int launch_cfg(int cfg, int mode, GemmArgs& a, hipStream_t stream) {
  if (cfg < 0 || cfg >= kNumTiles || !kTiles[cfg].built) {
    ufnd_set_error("gemm_bf16: tile config %d is not part of this library", cfg);
    return UFND_ERR_INVALID;
  }
  const TileCfg& t = kTiles[cfg];
  a.m_tiles = ufnd_cdiv(a.M, t.bm);
  a.n_tiles = a.N / t.bn;
  a.xcd_cols = xcd_cols_for(a.M, a.N, a.K, a.m_tiles, a.n_tiles);
  const dim3 grid(a.m_tiles * a.n_tiles * (mode == 6 && a.ksplit > 1 ? a.ksplit : 1)), block(t.threads);
#ifdef UFND_GEMM_ONLY_BWD
  // gemm_bf16_bwd.hip: only the backward kernels (mode 6) of the tiles the automatic choice can return
#define X(id, BM_, BN_, WM_, WN_, SA_, SB_, MI_, LN_, PROD_)                                                                      \
  case id:                                                                                                                        \
    if constexpr (id == 15 || id == 16 || id == 17 || id == 20 || id == 22) {                                                     \
      hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 0, 0, 0, 0, 1>), grid, block, 0, stream, a);        \
    } else {                                                                                                                      \
      ufnd_set_error("gemm_bf16 backward: tile %d has no backward kernel", id);                                                   \
      return UFND_ERR_INVALID;                                                                                                    \
    }                                                                                                                             \
    break;
  (void)mode;
  switch (cfg) {
    UFND_GEMM_TILES(X)
    default: ufnd_set_error("gemm_bf16: unknown tile config %d", cfg); return UFND_ERR_INVALID;
  }
#undef X
  return UFND_OK;
#else
#ifdef UFND_DIAG
#define UFND_DIAG_LAUNCH(BM_, BN_, WM_, WN_, SA_, SB_, MI_)                                                                             \
  else if (mode == 5) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 0, 1, 1>), grid, block, 0, stream, a);   \
  else if (mode == 1) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 1, 0>), grid, block, 0, stream, a);      \
  else if (mode == 2) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 2, 0>), grid, block, 0, stream, a);      \
  else if (mode == 3) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 0, 1>), grid, block, 0, stream, a);
#else
#define UFND_DIAG_LAUNCH(BM_, BN_, WM_, WN_, SA_, SB_, MI_)
#endif
#define X(id, BM_, BN_, WM_, WN_, SA_, SB_, MI_, LN_, PROD_)                                                        \
  case id:                                                                                                          \
    if constexpr (UFND_TILE_BUILT(PROD_) != 0) {                                                                    \
      if (mode == 0) {                                                                                              \
        hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 0, 0>), grid, block, 0, stream, a); \
      } else if constexpr (LN_ != 0) { /* LayerNorm-aware (and diagnostics) builds exist for these tiles only */   \
        if (mode == 4) hipLaunchKernelGGL((gemm_bf16_kernel<BM_, BN_, WM_, WN_, SA_, SB_, MI_, 0, 0, 1>), grid, block, 0, stream, a); \
        UFND_DIAG_LAUNCH(BM_, BN_, WM_, WN_, SA_, SB_, MI_)                                                         \
        else {                                                                                                      \
          ufnd_set_error("gemm_bf16: tile %d has no kernel for mode %d in this library", id, mode);                 \
          return UFND_ERR_INVALID;                                                                                  \
        }                                                                                                           \
      } else {                                                                                                      \
        ufnd_set_error("gemm_bf16: tile %d has no diagnostics / LayerNorm-aware build", id);                        \
        return UFND_ERR_INVALID;                                                                                    \
      }                                                                                                             \
    }                                                                                                               \
    break;
  switch (cfg) {
    UFND_GEMM_TILES(X)
    default: ufnd_set_error("gemm_bf16: unknown tile config %d", cfg); return UFND_ERR_INVALID;
  }
#undef X
#undef UFND_DIAG_LAUNCH
  return UFND_OK;
#endif
}

// Per-shape choice from the on-device sweep (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt):
// a launch wants about one equal tile per CU (256), and the largest tile that still gives that
// many, because L2->LDS traffic falls as 1/BM + 1/BN; 8-wave workgroups throughout (two waves per
// SIMD keep the matrix pipe fed across the mid-step barrier).
static int auto_cfg(int M, int N, int K) {
#ifdef UFND_DIAG      // experiments only: the product library has no environment override
  static const int forced = [] { const char* e = getenv("UFND_GEMM_FORCE_CFG"); return e ? atoi(e) : -1; }();
  if (forced >= 0 && forced < kNumTiles && N % kTiles[forced].bn == 0) return forced;
#endif
  auto tiles = [&](int cfg) { return (long long)ufnd_cdiv(M, kTiles[cfg].bm) * (N / kTiles[cfg].bn); };
  // (256x144 gives BERT QKV 256 tiles instead of 192 and is 8 % faster alone -- 17.2 vs 18.8 us -- but the step got
  //  4 % SLOWER with it: the 64 CUs the 192-tile launch leaves free are where the other encoder's and the head's
  //  kernels run meanwhile.  Tile 28 stays in the table for the sweep; it is not selected.)
  // 256x192: BERT QKV (192 tiles) / FFN1 (256).  The A ring is 3 slots deep (tile 22, not 8): inside the encoder the
  // activation operand was just written by the previous kernel and comes from the Infinity Cache / HBM, not from L2
  // (gemm_sweep --rotate=40: 21.6 -> 19.4 us on QKV, 26.8 -> 24.7 on FFN1; cold WEIGHTS cost under 1 us either way)
  // 256x256 where a wide-N launch has at least two full rounds of it (FFN1 at 16,384 rows = encoder lookahead 4: 768 tiles
  // instead of 1024 of 256x192 -- the only tile whose K loop is not bound by the CU's operand ingest; 114 -> 108 us per launch,
  // step +1.1 % in three interleaved pairs of runs.  Not for N = 2304: 9 column tiles, 64 vs 57 us alone.)
  if (N % 256 == 0 && N >= 3072 && tiles(15) >= 512) return 15;
  // ... and where it fills ONE round (ViT QKV at 6,400 rows: 225 tiles instead of 300 of 256x192 in 1.17 rounds: 40 -> 29.5 us
  // per launch, step +0.8 % in two interleaved pairs)
  if (N % 256 == 0 && N >= 2048 && tiles(15) >= 190 && tiles(15) <= 256) return 15;
  if (N % 192 == 0 && tiles(22) >= 160) return 22;
  if (N >= 2048) {                                     // wide N, fewer rows (ViT QKV / FFN1): 32-row wave tiles
    if (N % 192 == 0 && N >= 3072) return 17;          //   128x192
    if (N % 128 == 0) return 16;                       //   128x128
  }
  // narrow N with 128x192 tiles filling ONE round (ViT out-proj / FFN2 at 6,400 rows = encoder lookahead 4: 200 tiles instead of
  // 300 tiles of 128x128 in 1.17 rounds -- FFN2 55 -> 44 us, out-proj 22 -> 18.5 us, step +0.9 % in three interleaved pairs of runs)
  if (N < 2048 && N % 192 == 0 && tiles(17) >= 150 && tiles(17) <= 256) return 17;
  if (N % 128 == 0 && tiles(16) >= 150) return 16;     // narrow N at M=4096 (attention out-proj, output.dense): 128x128
  return 20;                                           // small problems (ViT out-proj / FFN2 / patch embedding): 128x64, 4-slot ring
}

