// Persistent, software-pipelined form of the bf16 GEMM (round 4): out = act(LN?(A) W^T + bias) [+ residual], K = 768.
//
// Why it exists (profiles/r03_gemm_stamps_g4.txt): on the K = 768 Linears a one-tile-per-workgroup kernel spends 40-48 % of
// a workgroup's life outside its K loop -- 3 us of prologue (statistics + the first ring stages) and 7-9 us of epilogue
// (GELU / residual / whole-row stores) around 13-18 us of matrix work -- and nothing else can run on that CU meanwhile: the
// 256x192 / 256x256 tiles own all of its LDS and registers.
//
// Structure here: grid = one workgroup per CU; a workgroup walks a list of 256 x 128 tiles.  Its eight waves keep TWO
// accumulator sets (wave tile 64 x 64: 64 + 64 registers): while the MFMAs of tile i+1 fill one set, the epilogue of tile
// i drains the other, cut into slices that are dealt over the twelve K-steps of tile i+1 (one 8-row x 64-column patch per
// wave per K-step: register phase -> wave-private LDS patch -> whole-row phase -> 16-B stores).  The LDS-DMA operand stream
// never drains: the ring stages of tile i+1 are requested during the last K-steps of tile i, so a prologue is paid once per
// workgroup and an un-overlapped epilogue once (the last tile's).  The C staging has its own LDS (17 KiB) instead of
// overlaying the ring.
//
// vmcnt discipline: LDS-DMA pieces, the epilogue's loads (statistics, residual rows) and its stores all count in ONE
// in-order counter.  Every vector-memory operation inside the K loop is therefore issued unconditionally by every wave, in a
// program order pinned by `asm volatile` + "memory", and every wait is an exact count derived from the static schedule
// (n_fh / n_st below).  Epilogue loads and stores are inline asm: hipcc would otherwise drain the whole queue
// (s_waitcnt vmcnt(0)) in front of their first use (cdna_hip_programming.md section 5, trap (b)).
//
// Arithmetic: the same operations in the same order as gemm_bf16_kernel (k ascending in steps of 32 per MFMA; the epilogue
// code is the same expression sequence), so a row's results do not depend on which kernel -- or which batch -- computed it:
// tests/test_gpu_gemm_pp.py holds the two kernels bit-identical.
#pragma once
#include "gemm_bf16_kernel.hpp"

#ifndef UFND_PP_WPE
#define UFND_PP_WPE 2
#endif
#define UFND_GEMM_TILE_PP UFND_GEMM_TILE_PERSISTENT      // the id this form answers to in ufnd_gemm_ln.tile_cfg / ufnd_gemm_bf16_ex (include/ultrafnd_hip.h; not a table entry)

namespace {
namespace pp {
constexpr int BM = 256, BN = 128, NW = 8, WN = 2, TM = 64, TN = 64, MT = 4, NT = 4;
constexpr int STA = 3, STB = 2, ASLOT = BM * 128, BSLOT = BN * 128, RING = STA * ASLOT + STB * BSLOT;
constexpr int A_OFF = STB * BSLOT;                       // LDS: [W ring | A ring | C patches | statistics | vectors]
constexpr int PWA = BM / 8 / NW, PWB = BN / 8 / NW;      // DMA pieces per wave per K-step: 4 + 2
constexpr int DMA = PWA + PWB;
constexpr int CP = TN + 4, PROWS = 8, CBYTES = NW * PROWS * CP * 4;
constexpr int ST_OFF = RING + CBYTES;                    // {mean, rstd} of the 256 rows of the tile whose epilogue is (about to be) in flight
constexpr int VEC_OFF = ST_OFF + BM * 8;                 // bias | colsum | gamma | beta of that tile's 128 columns
constexpr int GRD_OFF = VEC_OFF + 4 * BN * 4;            // the waves' guard maxima
constexpr int SMEM = GRD_OFF + 64;
constexpr int NK = 12;                                   // K-steps per tile (K = 768)
constexpr int MIN_TILES = 512;                           // automatic choice: at least two tiles per workgroup (folded LayerNorm + activation)
constexpr int NPATCH = 8;                                // 8-row x 64-column patches per wave tile: the patch of K-step s is patch s
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(NK % STA == 0 && NK % STB == 0, "a tile must start on slot 0 of both rings");

enum { PLAIN = 0, FOLD = 1, RES_LN = 2, RES = 3 };       // epilogue modes

// The static schedule of the epilogue's vector-memory operations, per wave and K-step s of a tile's K loop:
//   s = 10   first half: the tile's OWN row statistics (6 loads) + column vectors (1)            [every tile]
//   s = 11   first half: wait for them, {mean, rstd} and vectors -> LDS; residual rows of patch 0  [every tile]
//   s = 0..7 (the tile BEFORE's accumulators): first half: register phase of patch s, residual rows of patch s + 1;
//            second half: row phase of patch s, its stores                                      [tiles with a predecessor]
template <int MODE> constexpr int n_sl() { return (MODE == FOLD || MODE == RES_LN) ? 7 : 1; }
template <int MODE> constexpr int n_lr(int s, bool epi) { return MODE >= RES_LN && (s == 11 || (epi && s >= 0 && s <= NPATCH - 2)) ? 1 : 0; }
template <int MODE> constexpr int n_fh(int s, bool epi) { return s == 10 ? n_sl<MODE>() : n_lr<MODE>(s, epi); }      // issued in a step's first half
template <int MODE, int DBG = 0> constexpr int n_st(int s, bool epi) { return (!(DBG & (4 | 16)) && epi && s >= 0 && s <= NPATCH - 1) ? (MODE >= RES_LN ? 2 : 1) : 0; }      // (DBG & 4: timing-only build without the stores)

__device__ __forceinline__ void ld16(f32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld4(float& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void st16(void* p, f32x4 v) {      // (s_nop: the data registers may be rewritten right behind an asm store)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st8(void* p, f32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int KT>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voff, int soff) {      // 1 KiB: 64 lanes x 16 B, LDS-linear
  // (the K-step goes into the scalar offset: the instruction's immediate offset is added to the LDS address as well)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)lds_dst, 16, voff, soff + KT * BK * 2, 0, 0);
}
}  // namespace pp

template <int MODE, int ACT, int DBG>
// (built without packed fp32 instructions: see gemm_bf16_pp.hip)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(UFND_PP_WPE)))
void gemm_pp_kernel(const GemmArgs a) {
  using namespace pp;
  static_assert(MODE == FOLD || ACT == UFND_ACT_NONE || MODE == PLAIN, "activations: plain and folded calls");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  unsigned long long stamp[8];
  if constexpr (DBG & 1) {
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, g = lane >> 4;

  // ---- the tiles of this workgroup.  Logical workgroup ids are XCD-contiguous; XCD x owns the contiguous range
  // [x T / 8, (x + 1) T / 8) of the tile order below and deals it round-robin to its workgroups, so the workgroups of an XCD
  // work on neighbouring tiles at any time: pp_rows row panels x (workgroups per XCD / pp_rows) column tiles, whose A panels
  // stay in that XCD's L2 while it walks the columns.
  const int G = gridDim.x, T = a.m_tiles * a.n_tiles;
  const int gpx = G >> 3;
  const int L = xcd_contiguous_id(blockIdx.x, G);
  const int xcd = L / gpx;
  const int u_end = (int)(((long long)(xcd + 1) * T) >> 3);
  int u = (int)(((long long)xcd * T) >> 3) + (L - xcd * gpx);
  if (u >= u_end) return;
  auto coords = [&](int uu, int& m0, int& n0) {      // tile order: groups of pp_rows row panels, column-major inside a group
    const int R = a.pp_rows, per = R * a.n_tiles;
    const int rg = uu / per, rem = uu - rg * per;
    int rows = a.m_tiles - rg * R;
    rows = rows < R ? rows : R;
    const int tn = rem / rows;
    m0 = (rg * R + (rem - tn * rows)) * BM;
    n0 = tn * BN;
  };

  // ---- operand stream: buffer_load ... lds, ONE offset register per operand.  Piece p = wave + 8 ii holds tile rows 8p .. 8p+7
  // (lane: row 8p + lane / 8, 16-B chunk (lane % 8) ^ ((row >> 1) & 7) -- the swizzle does not depend on ii), so the pieces of a
  // wave differ by a scalar: soffset = (tile row 0 + 64 ii) * ld * 2 + K-step * 128.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 0x7fffffff, 0x00020000);
  int voffA, voffW;
  {
    const int r = 8 * wave + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    voffA = (r * a.lda + c * 8) * 2;
    voffW = (r * a.ldw + c * 8) * 2;
  }
  const int strideA = 64 * a.lda * 2, strideW = 64 * a.ldw * 2;      // bytes between a wave's consecutive pieces
  auto issueA2 = [&](auto kt_, int sA, int slot, int half) {          // sA = m0 * lda * 2; pieces 2 half, 2 half + 1 of this wave
    constexpr int KT = decltype(kt_)::value;
    char* buf = smem + A_OFF + slot * ASLOT + wave * 1024;
#pragma unroll
    for (int ii = 2 * half; ii < 2 * half + 2; ++ii) dma_piece<KT>(rsA, buf + ii * NW * 1024, voffA, sA + ii * strideA);
  };
  auto issueA = [&](auto kt_, int sA, int slot) { issueA2(kt_, sA, slot, 0); issueA2(kt_, sA, slot, 1); };
  auto issueB = [&](auto kt_, int sW, int slot) {
    constexpr int KT = decltype(kt_)::value;
    char* buf = smem + slot * BSLOT + wave * 1024;
#pragma unroll
    for (int ii = 0; ii < PWB; ++ii) dma_piece<KT>(rsW, buf + ii * NW * 1024, voffW, sW + ii * strideW);
  };

  // two accumulator sets in ping-pong: a tile accumulates into one while the epilogue of the tile before drains the other
  // (no copy at the seam: a copy makes the register allocator hold three sets there and spill)
  using acc_t = f32x4[MT][NT];
  acc_t accX, accY;
  bf16x8 af0[MT], bf0[NT], af1[MT], bf1[NT];
  // Fragment addresses: the swizzle of a row depends on (row >> 1) & 7 = (fr >> 1) & 7 only, so a fragment's LDS address is
  // [lane part of its k-half] + slot * SLOT + tile * 2048.  Six address registers (W: one per k-half; A: two per k-half, because
  // the 16-bit ds_read offset does not reach the third A slot) and immediates -- left to itself the compiler materialises one
  // address register per (slot, k-half, operand) and spills the parked accumulators to hold them.
  unsigned fa_lo[2], fb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const unsigned sw = (unsigned)(((g + 4 * kk) ^ ((fr >> 1) & 7)) << 4);
    fb[kk] = (unsigned)(wn * TN + fr) * 128u + sw;
    fa_lo[kk] = (unsigned)A_OFF + (unsigned)(wm * TM + fr) * 128u + sw;
    asm volatile("" : "+v"(fb[kk]), "+v"(fa_lo[kk]));
  }
  using lds_frag_ptr = const bf16x8 __attribute__((address_space(3)))*;
  const unsigned lds0 = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)smem;
  auto ldA = [&](int slot, int kk, int i) {      // slot, kk, i are compile-time at every call site
    const unsigned base = slot == 0 ? fa_lo[kk] : fa_lo[kk] + (unsigned)ASLOT;
    return *(lds_frag_ptr)(size_t)(lds0 + base + (unsigned)((slot == 0 ? 0 : slot - 1) * ASLOT + i * 2048));
  };
  auto ldB = [&](int slot, int kk, int j) { return *(lds_frag_ptr)(size_t)(lds0 + fb[kk] + (unsigned)(slot * BSLOT + j * 2048)); };
  auto read_half = [&](int sa, int sb, int kk, bf16x8 (&af)[MT], bf16x8 (&bfr)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = ldA(sa, kk, i);
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[j] = ldB(sb, kk, j);
  };
  // the MFMAs of row tile i of a k-half
  auto mma_row = [&](acc_t& acc, auto i_, const bf16x8 (&af)[MT], const bf16x8 (&bfr)[NT]) {
    constexpr int i = decltype(i_)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  };
  // group q of a half-step reads two fragments of the NEXT k-half: q = 0: W 0, 1; q = 1: W 2, 3; q = 2: A 0, 1; q = 3: A 2, 3 --
  // every fragment is in registers at least one group (four MFMAs) before its first use
  auto read_q = [&](auto q_, int sa, int sb, int kk, bf16x8 (&af)[MT], bf16x8 (&bfr)[NT]) {
    constexpr int q = decltype(q_)::value;
    if constexpr (q < 2) { bfr[2 * q] = ldB(sb, kk, 2 * q); bfr[2 * q + 1] = ldB(sb, kk, 2 * q + 1); }
    else { af[2 * (q - 2)] = ldA(sa, kk, 2 * (q - 2)); af[2 * (q - 2) + 1] = ldA(sa, kk, 2 * (q - 2) + 1); }
  };

  // The slices derive every lane-dependent index from an OPAQUE copy of the lane / thread id at the point of use: a plain
  // expression of threadIdx.x is loop-invariant, gets hoisted out of the tile loop, lives across all K loops and is spilled
  // (each reload of such a value drains the operand stream: scratch loads count in vmcnt).
  auto olane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto otid = [&]() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; };

  // ---- epilogue state
  int m0c = 0, n0c = 0;           // the tile in acc[][]
  int m0p = 0, n0p = 0;           // the tile in prv[][]
  f32x4 sv[2][3];                 // step 10 -> 11: partial row statistics
  float vecv = 0.f;               //                one element of the column vectors
  f32x4 rres0, rres1;             // residual rows (bf16 x 8) of the even / odd patches
  float guard_max = 0.f;
  f32x2* st_lds = reinterpret_cast<f32x2*>(smem + ST_OFF);
  float* vec_lds = reinterpret_cast<float*>(smem + VEC_OFF);
  float* cst = reinterpret_cast<float*>(smem + RING) + wave * PROWS * CP;
  const float* stat_src = MODE == FOLD ? a.a_stats : a.r_stats;
  const int stat_parts = MODE == FOLD ? a.a_parts : a.r_parts;
  const float stat_eps = MODE == FOLD ? a.a_eps : a.r_eps;

  // tile row of patch P = 2 i + h, patch-local row pr (0..7): accumulator registers 2h, 2h+1 of the four lane groups of row tile i
  //   = (wm * TM + 4 (pr >> 1) + (pr & 1))  [lane part, pr = lane / 8 in the row phase]  +  (P >> 1) * 16 + 2 (P & 1)  [static]
  // The slices are fillers between MFMAs, and there a vector instruction beyond the two a 16x16x32 MFMA covers costs its full
  // four issue cycles (measured: the slices' instruction count times four IS what they add to a K-step).  So no address arithmetic
  // in them: everything lane-dependent is computed ONCE here, made opaque (the compiler must neither re-derive it per use nor
  // expand it into one register per patch), and the per-patch part of every global address is a scalar offset of a buffer
  // instruction.
  auto prow_of = [&](int P) { return (P >> 1) * 16 + 2 * (P & 1); };
  const int rp_row = wm * TM + 4 * (lane >> 4) + ((lane >> 3) & 1);      // row part of the lane in the row phase
  const int rp_col = wn * TN + (lane & 7) * 8;                           // its first column inside the tile
  unsigned cw_a = lds0 + RING + (unsigned)(wave * PROWS * CP + 2 * g * CP + fr) * 4u;                     // patch, register phase: row 2g, column fr
  unsigned crd_a = lds0 + RING + (unsigned)(wave * PROWS * CP + (lane >> 3) * CP + (lane & 7) * 8) * 4u;  // patch, row phase: row lane / 8
  unsigned vb_a = lds0 + VEC_OFF + (unsigned)(wn * TN + fr) * 4u;                                         // bias (+ BN * 4: column sum) of column fr
  unsigned st_a = lds0 + ST_OFF + (unsigned)(wm * TM + 4 * g) * 8u;                                       // {mean, rstd} of row 4g (register phase)
  unsigned rs_a = lds0 + ST_OFF + (unsigned)rp_row * 8u;                                                  // {mean, rstd} of my row (row phase)
  unsigned gb_a = lds0 + VEC_OFF + (unsigned)(2 * BN + rp_col) * 4u;                                      // gamma (+ BN * 4: beta) of my 8 columns
  unsigned vo_out = (unsigned)(rp_row * a.ldo + rp_col) * 2u;                                             // byte offsets of my row-phase chunk
  unsigned vo_res = (unsigned)(rp_row * a.ldrb + rp_col) * 2u;
  unsigned vo_stat = (unsigned)(rp_row * (a.N >> 5) + (rp_col >> 5)) * 8u;
  asm volatile("" : "+v"(cw_a), "+v"(crd_a), "+v"(vb_a), "+v"(vo_out));
  if constexpr (MODE == FOLD) asm volatile("" : "+v"(st_a));
  if constexpr (MODE == RES_LN) asm volatile("" : "+v"(rs_a), "+v"(gb_a));
  if constexpr (MODE >= RES_LN) asm volatile("" : "+v"(vo_res), "+v"(vo_stat));
  using lds_f1 = const float __attribute__((address_space(3)))*;
  using lds_f2 = const f32x2 __attribute__((address_space(3)))*;
  using lds_f4 = const f32x4 __attribute__((address_space(3)))*;
  using lds_w1 = float __attribute__((address_space(3)))*;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsOut = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_bf16, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsRes = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE >= RES_LN ? (const void*)a.residual_b : (const void*)a.A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsStat = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE >= RES_LN ? (void*)a.out_stats : (void*)a.out_bf16), 0, 0x7fffffff, 0x00020000);

  // step 10, first half: request the row statistics and column vectors of the tile being accumulated (m0c, n0c)
  auto slice_loads = [&]() {
    if constexpr (MODE == FOLD || MODE == RES_LN) {
      const int tid = otid();
      const int row = m0c + (tid >> 1), sub = tid & 1, nq = stat_parts >> 1;
      const f32x4* base = reinterpret_cast<const f32x4*>(stat_src + (unsigned)row * (unsigned)stat_parts * 2u);
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
          const int q = (sub + gi * 2) + 4 * q3;
          ld16(sv[gi][q3], base + (q < nq ? q : 0));
        }
    }
    {
      // (which vector: wave-uniform -- two waves per vector -- so the pointer is picked on the scalar unit; picked per lane, the
      //  compiler fetches it from the kernel-argument block with a VECTOR load and drains the operand stream for it)
      const int which = wave >> 1, c = otid() & 127;
      const float* src = which == 0 ? a.bias : which == 1 ? (MODE == FOLD ? a.colsum : nullptr) : (MODE == RES_LN ? (which == 2 ? a.r_gamma : a.r_beta) : nullptr);
      const void* ptr = src ? (const void*)(src + n0c + c) : (const void*)a.W;      // (an absent vector: a load that is counted and ignored)
      ld4(vecv, ptr);
    }
  };
  // step 11, first half (behind the wait for slice_loads): {mean, rstd} and the vectors into LDS.  The summation order of the
  // partials is the canonical one of gemm_bf16_kernel (two threads per row: groups sub, sub + 2).
  auto slice_reduce = [&]() {
    const int tid = otid();
    if constexpr (MODE == FOLD || MODE == RES_LN) {
      const int sub = tid & 1, nq = stat_parts >> 1;
      float gs[2], gq[2];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
          if ((sub + gi * 2) + 4 * q3 >= nq) sv[gi][q3] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        gs[gi] = sv[gi][0][0] + sv[gi][0][2];
        gq[gi] = sv[gi][0][1] + sv[gi][0][3];
#pragma unroll
        for (int q3 = 1; q3 < 3; ++q3) { gs[gi] += sv[gi][q3][0] + sv[gi][q3][2]; gq[gi] += sv[gi][q3][1] + sv[gi][q3][3]; }
      }
      const float s01 = gs[0] + quad_xor1(gs[0]), s23 = gs[1] + quad_xor1(gs[1]);
      const float q01 = gq[0] + quad_xor1(gq[0]), q23 = gq[1] + quad_xor1(gq[1]);
      const float sm = s01 + s23, sq = q01 + q23;
      const float mean = __fmul_rn(sm, a.inv_h);
      const float var = fmaxf(__fmaf_rn(-mean, mean, __fmul_rn(sq, a.inv_h)), 0.f);
      const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(var, stat_eps));
      if (sub == 0) st_lds[tid >> 1] = f32x2{mean, rstd};
      if constexpr (MODE == FOLD) {
        float ratio = __fmul_rn(fabsf(mean), rstd);
        ratio = ratio == ratio ? ratio : INFINITY;      // (a NaN statistic must trip the guard: fmaxf would drop it)
        guard_max = fmaxf(guard_max, sub == 0 ? ratio : 0.0f);
      }
    }
    {
      const int which = wave >> 1;
      const bool present = which == 0 ? a.bias != nullptr : which == 1 ? (MODE == FOLD) : (MODE == RES_LN);
      vec_lds[tid] = present ? vecv : 0.0f;
    }
  };
  // residual rows of patch P of the tile at (m0, n0): requested one K-step before the patch's row phase
  auto slice_res_load = [&](int P, int m0, int n0, f32x4& dst) {
    const int soff = ((m0 + prow_of(P)) * a.ldrb + n0) * 2;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo_res), "s"(rsRes), "s"(soff) : "memory");
  };
  // register phase of patch P, column tile j: [folded LayerNorm,] bias, activation on the accumulators where they lie -> the
  // wave's LDS patch.  Everything the phase reads besides the accumulators is in registers BEFORE its K-step begins: the bias /
  // column sums of this lane's four accumulator columns (bjv, csjv: fetched once per tile, in step 11 of its own K loop, right
  // behind the barrier that publishes the staged vectors) and the two rows' {mean, rstd} (fetched one half-step ahead) -- an LDS
  // read inside a slice is a full LDS round trip of latency on the wave's critical path, eight times per tile.
  f32x2 mean2 = {0.f, 0.f}, rstd2 = {1.f, 1.f};
  float bjv[NT], csjv[NT];      // bias / column sums of this lane's four accumulator columns, per tile
  auto slice_vectors = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bjv[j] = *(lds_f1)(size_t)(vb_a + (unsigned)(j * 64));
      if constexpr (MODE == FOLD) csjv[j] = *(lds_f1)(size_t)(vb_a + (unsigned)(BN * 4 + j * 64));
      else csjv[j] = 0.f;
    }
  };
  auto slice_stats_prefetch = [&](int P) {      // {mean, rstd} of the two rows patch P's register phase works on
    if constexpr (MODE == FOLD) {
      const f32x4 ms = *(lds_f4)(size_t)(st_a + (unsigned)(((P >> 1) * 16 + 2 * (P & 1)) * 8));      // rows 4g + r0, 4g + r0 + 1
      mean2 = f32x2{ms[0], ms[2]};
      rstd2 = f32x2{ms[1], ms[3]};
    }
  };
  auto slice_regs_j = [&](acc_t& prv, auto P_, auto j_) {
    constexpr int P = decltype(P_)::value, i = P >> 1, r0 = 2 * (P & 1), j = decltype(j_)::value;
    f32x2 v = {prv[i][j][r0], prv[i][j][r0 + 1]};
    const float bj = bjv[j], csj = csjv[j];
    if constexpr (MODE == FOLD) v = opaque_f2(rstd2 * __builtin_elementwise_fma(-mean2, f32x2{csj, csj}, v));
    v = v + f32x2{bj, bj};
    if constexpr (ACT == UFND_ACT_GELU) v = gelu_fast_f2(v);
    else if constexpr (ACT == UFND_ACT_QUICK_GELU) v = quick_gelu_fast_f2(v);
    if constexpr (DBG & 8) { asm volatile("" : : "v"(v.x), "v"(v.y)); return; }      // (timing-only: no patch writes)
    *(lds_w1)(size_t)(cw_a + (unsigned)(j * 64)) = v.x;
    *(lds_w1)(size_t)(cw_a + (unsigned)(CP * 4 + j * 64)) = v.y;
  };
  auto slice_regs = [&](acc_t& prv, auto P_) {
    slice_regs_j(prv, P_, IntC<0>{}); slice_regs_j(prv, P_, IntC<1>{}); slice_regs_j(prv, P_, IntC<2>{}); slice_regs_j(prv, P_, IntC<3>{});
  };
  // row phase of patch P: whole rows out of the patch: residual [through its LayerNorm], statistics, 16-B stores -- in three
  // pieces (a: the patch row out of LDS; b: residual and statistics; c: rounding and stores); v8 carries the row between them
  float v8[8];
  auto slice_rows_a = [&](auto P_) {
    const f32x4 v0 = *(lds_f4)(size_t)(crd_a);
    const f32x4 v1 = *(lds_f4)(size_t)(crd_a + 16u);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v8[q] = v0[q]; v8[4 + q] = v1[q]; }
  };
  auto slice_rows_b = [&](auto P_, const f32x4& rres) {
    constexpr int P = decltype(P_)::value;
    if constexpr (MODE >= RES_LN) {
      const bf16x8 rb = __builtin_bit_cast(bf16x8, rres);
      float r8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r8[q] = (float)rb[q];
      if constexpr (MODE == RES_LN) {
        const f32x2 ms = *(lds_f2)(size_t)(rs_a + (unsigned)(prow_of(P) * 8));
        const f32x4 g0 = *(lds_f4)(size_t)(gb_a), g1 = *(lds_f4)(size_t)(gb_a + 16u);
        const f32x4 b0 = *(lds_f4)(size_t)(gb_a + (unsigned)(BN * 4)), b1 = *(lds_f4)(size_t)(gb_a + (unsigned)(BN * 4 + 16));
        const float g8[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float b8[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) r8[q] = __fmaf_rn(__fmul_rn(__fsub_rn(r8[q], ms[0]), ms[1]), g8[q], b8[q]);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v8[q] = __fadd_rn(v8[q], r8[q]);
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { sm = __fadd_rn(sm, v8[q]); sq = __fmaf_rn(v8[q], v8[q], sq); }
      sm += quad_xor1(sm);
      sq += quad_xor1(sq);
      sm += quad_xor2(sm);
      sq += quad_xor2(sq);
      // (every lane stores: the four lanes of a 32-column group hold the same pair and write the same 8 bytes -- an
      //  unpredicated instruction keeps the vmcnt schedule exact)
      if constexpr (DBG & 4) asm volatile("" : : "v"(sm), "v"(sq));
      else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, f32x2{sm, sq}), rsStat, vo_stat,
                                                 ((m0p + prow_of(P)) * (a.N >> 5) + (n0p >> 5)) * 8, 0);
    }
  };
  auto slice_rows_c = [&](auto P_) {
    constexpr int P = decltype(P_)::value;
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (__bf16)v8[q];
    if constexpr (DBG & (4 | 16)) asm volatile("" : : "v"(o));
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsOut, vo_out, ((m0p + prow_of(P)) * a.ldo + n0p) * 2, UFND_GEMM_OUT_NT ? 2 : 0);      // (aux bit 1 = nt)
  };
  auto slice_rows = [&](auto P_, const f32x4& rres) { slice_rows_a(P_); slice_rows_b(P_, rres); slice_rows_c(P_); };

  // ---- one K-step.  S: step inside the tile (static: ring slots, slices and wait counts are compile-time);
  // EPI: the epilogue slices of the tile before (prv) ride along.  sAc / sWc: this tile's operand offsets, sAn / sWn: the next tile's.
  auto kstep = [&](acc_t& acc, acc_t& prv, auto S_, auto EPI_, bool has_next, int sAc, int sWc, int sAn, int sWn) {
    constexpr int S = decltype(S_)::value;
    constexpr bool EPI = decltype(EPI_)::value != 0 && !(DBG & 2);      // (DBG & 2: timing-only build without the epilogue slices; results are garbage)
    constexpr bool PATCH = EPI && S < NPATCH;
    constexpr bool PATCH_REGS = PATCH && !(DBG & 32), PATCH_ROWS = PATCH && !(DBG & 16);      // (timing-only builds without one phase)
    constexpr int SA = S % STA, SB = S % STB, SA1 = (S + 1) % STA, SB1 = (S + 1) % STB;
    // ---- first half: the MFMAs of k-half 0 while the fragments of k-half 1 arrive.  Four groups of {2 fragment reads, 4 MFMAs,
    // a quarter of the register phase of patch S}, each fenced: the compiler schedules inside a group, never across one (left
    // alone it hoists the fragment reads of a whole half-step and spills the parked accumulators)
    // Inside a group the MFMAs and the slice's vector instructions ALTERNATE (one MFMA, then VPM of the slice's instructions):
    // the two waves of a SIMD run this code in lockstep between barriers, so a group issued as {4 MFMAs, then the slice} has both
    // waves queueing on the matrix pipe and then both on the vector ALU with the matrix pipe idle -- measured: the slices then add
    // their full issue time to the K-step.  Alternating, each wave's vector work issues while the other wave's MFMA executes.
    constexpr int VPM = !PATCH ? 0 : (ACT == UFND_ACT_GELU ? 16 : ACT == UFND_ACT_QUICK_GELU ? 11 : MODE == FOLD ? 8 : 5);
    auto interleave = [&]() {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // fragment reads
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      }
      if constexpr (PATCH) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    };
    read_q(IntC<0>{}, SA, SB, 1, af1, bf1);
    mma_row(acc, IntC<0>{}, af0, bf0);
    if constexpr (PATCH_REGS) slice_regs_j(prv, IntC<S>{}, IntC<0>{});
    interleave();
    if constexpr (S == 10) slice_loads();
    __builtin_amdgcn_sched_barrier(0);
    read_q(IntC<1>{}, SA, SB, 1, af1, bf1);
    mma_row(acc, IntC<1>{}, af0, bf0);
    if constexpr (PATCH_REGS) slice_regs_j(prv, IntC<S>{}, IntC<1>{});
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    read_q(IntC<2>{}, SA, SB, 1, af1, bf1);
    mma_row(acc, IntC<2>{}, af0, bf0);
    if constexpr (PATCH_REGS) slice_regs_j(prv, IntC<S>{}, IntC<2>{});
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    read_q(IntC<3>{}, SA, SB, 1, af1, bf1);
    mma_row(acc, IntC<3>{}, af0, bf0);
    if constexpr (PATCH_REGS) slice_regs_j(prv, IntC<S>{}, IntC<3>{});
    interleave();
    if constexpr (PATCH) {
      if constexpr (MODE >= RES_LN && S + 1 < NPATCH) slice_res_load(S + 1, m0p, n0p, ((S + 1) & 1) ? rres1 : rres0);
    }
    if constexpr (S == 11) {
      // the statistics are older than step 10's DMA pieces (issued only when a next tile exists); nothing else was issued since
      // ONE asm statement for both cases (the count is picked by a scalar branch inside it): two statements naming the same
      // registers make the compiler copy them in front of the branch -- i.e. read the loads' destinations BEFORE the wait
      // (found as garbage statistics on a workgroup's last tile, some launches only)
      const int hn = __builtin_amdgcn_readfirstlane(has_next ? 1 : 0);      // (an "s" operand must be provably uniform)
      if constexpr (MODE == FOLD || MODE == RES_LN) {
        asm volatile("s_cmp_lg_u32 %8, 0\n\ts_cbranch_scc1 .Lppw%=\n\ts_waitcnt vmcnt(0)\n\ts_branch .Lppx%=\n.Lppw%=:\n\ts_waitcnt vmcnt(%7)\n.Lppx%=: ; PPRETIRE %0 %1 %2 %3 %4 %5 %6"
                     : "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]), "+v"(sv[1][0]), "+v"(sv[1][1]), "+v"(sv[1][2]), "+v"(vecv)
                     : "n"(DMA), "s"(hn) : "memory", "scc");
      } else {
        asm volatile("s_cmp_lg_u32 %2, 0\n\ts_cbranch_scc1 .Lppw%=\n\ts_waitcnt vmcnt(0)\n\ts_branch .Lppx%=\n.Lppw%=:\n\ts_waitcnt vmcnt(%1)\n.Lppx%=: ; PPRETIRE %0"
                     : "+v"(vecv) : "n"(DMA), "s"(hn) : "memory", "scc");
      }
      slice_reduce();
      if constexpr (MODE >= RES_LN) slice_res_load(0, m0c, n0c, rres0);      // patch 0 of THIS tile: its row phase is step 0 of the next K loop
    }
    __builtin_amdgcn_sched_barrier(0);
    // step S + 1 is complete in LDS once my pieces W(S+1) [requested in step S-1] and A(S+1) [step S-2] have landed; younger
    // and allowed to fly: A(S+2) [step S-1; behind the seam only with a next tile], the stores of step S-1, this step's first-half loads
    if constexpr (S + 2 < NK) {
      wait_vmcnt<PWA + n_st<MODE, DBG>(S - 1, EPI) + n_fh<MODE>(S, EPI)>();
    } else {      // S = 10, 11: A(S+2) belongs to the next tile
      if (has_next) wait_vmcnt<PWA + n_st<MODE, DBG>(S - 1, EPI) + n_fh<MODE>(S, EPI)>();
      else wait_vmcnt<n_st<MODE, DBG>(S - 1, EPI) + n_fh<MODE>(S, EPI)>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half: the MFMAs of k-half 1; the fragments of step S+1's k-half 0 arrive; the slots step S has released are
    // refilled (W(S+2), A(S+3) -- of the next tile behind the seam); row phase of patch S
    const bool more = S + 1 < NK || has_next;
    // group 0: row tile 0; the patch row is requested first, its LDS round trip runs under the twelve MFMAs of groups 0 and 1
    if constexpr (PATCH_ROWS) slice_rows_a(IntC<S>{});
    if (more) { read_q(IntC<0>{}, SA1, SB1, 0, af0, bf0); read_q(IntC<1>{}, SA1, SB1, 0, af0, bf0); }
    if constexpr (S + STB < NK) issueB(IntC<S + STB>{}, sWc, SB);
    else if (has_next) issueB(IntC<(S + STB) % NK>{}, sWn, SB);
    if constexpr (S + STA < NK) issueA2(IntC<S + STA>{}, sAc, SA, 0);
    else if (has_next) issueA2(IntC<(S + STA) % NK>{}, sAn, SA, 0);
    mma_row(acc, IntC<0>{}, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    // group 1: row tiles 1, 2
    if (more) { read_q(IntC<2>{}, SA1, SB1, 0, af0, bf0); read_q(IntC<3>{}, SA1, SB1, 0, af0, bf0); }
    if constexpr (S + STA < NK) issueA2(IntC<S + STA>{}, sAc, SA, 1);
    else if (has_next) issueA2(IntC<(S + STA) % NK>{}, sAn, SA, 1);
    mma_row(acc, IntC<1>{}, af1, bf1);
    mma_row(acc, IntC<2>{}, af1, bf1);
    if constexpr (PATCH && MODE >= RES_LN) {
      // my residual rows were requested in step S-1's first half (step 11 of the K loop before for patch 0); younger: that
      // step's DMA and stores, this step's first-half load and DMA (all of them issued: S + 3 < 12)
      f32x4& rr = (S & 1) ? rres1 : rres0;
      asm volatile("s_waitcnt vmcnt(%1) ; PPRETIRE %0" : "+v"(rr) : "n"(2 * DMA + n_st<MODE, DBG>(S - 1, EPI) + n_lr<MODE>(S, EPI)) : "memory");
      slice_rows_b(IntC<S>{}, rr);
    }
    __builtin_amdgcn_sched_barrier(0);
    // group 2: row tile 3, then rounding and the store of the patch row, then what the next step's register phase reads
    mma_row(acc, IntC<3>{}, af1, bf1);
    if constexpr (PATCH_ROWS) slice_rows_c(IntC<S>{});
    if constexpr (EPI && S + 1 < NPATCH) slice_stats_prefetch(S + 1);
    if constexpr (S == 11 && !(DBG & 2)) { slice_vectors(); slice_stats_prefetch(0); }      // this tile's own epilogue starts with the next K loop
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- first tile: ring prologue (the only one this workgroup pays)
  coords(u, m0c, n0c);
  int sAc = m0c * a.lda * 2, sWc = n0c * a.ldw * 2;
  issueB(IntC<0>{}, sWc, 0);
  issueA(IntC<0>{}, sAc, 0);
  issueB(IntC<1>{}, sWc, 1);
  issueA(IntC<1>{}, sAc, 1);
  issueA(IntC<2>{}, sAc, 2);
  wait_vmcnt<PWA + PWB + PWA>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG & 1) {
    stamp[2] = __builtin_amdgcn_s_memtime();
    stamp[3] = __builtin_amdgcn_s_memrealtime();
  }
  read_half(0, 0, 0, af0, bf0);
  int ntiles = 0;
  bool last_in_x = true;
  // one tile: twelve K-steps into `acc` while the epilogue of the tile in `prv` (E = 1) rides along; returns whether a next tile exists
  int sAn = 0, sWn = 0, m0n = 0, n0n = 0, un = 0;
  bool has_next = false;
  auto look_ahead = [&]() {
    un = u + gpx;
    has_next = un < u_end;
    m0n = 0;
    n0n = 0;
    if (has_next) coords(un, m0n, n0n);
    sAn = m0n * a.lda * 2;
    sWn = n0n * a.ldw * 2;
  };
  auto advance = [&]() {      // the tile just accumulated becomes "the tile before"
    m0p = m0c;
    n0p = n0c;
    ++ntiles;
    u = un;
    m0c = m0n;
    n0c = n0n;
    sAc = sAn;
    sWc = sWn;
  };
#define UFND_PP_TILE(ACC, PRV, E)                                                                                                   \
  {                                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) ACC[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; \
    kstep(ACC, PRV, IntC<0>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<1>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn);   \
    kstep(ACC, PRV, IntC<2>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<3>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn);   \
    kstep(ACC, PRV, IntC<4>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<5>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn);   \
    kstep(ACC, PRV, IntC<6>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<7>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn);   \
    kstep(ACC, PRV, IntC<8>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<9>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn);   \
    kstep(ACC, PRV, IntC<10>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); kstep(ACC, PRV, IntC<11>{}, IntC<E>{}, has_next, sAc, sWc, sAn, sWn); \
  }
  look_ahead();
  UFND_PP_TILE(accX, accY, 0)
  while (has_next) {
    advance();
    look_ahead();
    UFND_PP_TILE(accY, accX, 1)
    last_in_x = false;
    if (!has_next) break;
    advance();
    look_ahead();
    UFND_PP_TILE(accX, accY, 1)
    last_in_x = true;
  }
#undef UFND_PP_TILE
  m0p = m0c;
  n0p = n0c;
  ++ntiles;
  if (!last_in_x) {      // (once per workgroup: the un-overlapped last epilogue is written for one set)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) accX[i][j] = accY[i][j];
  }
  if constexpr (DBG & 1) {
    stamp[4] = __builtin_amdgcn_s_memtime();
    stamp[5] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- the last tile's epilogue, the only one that is not hidden (same slices, one after the other; its statistics and
  // vectors are in LDS since step 11, and so is the request for patch 0's residual rows)
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
#define UFND_PP_LAST(P_)                                                                                             \
  {                                                                                                                  \
    slice_stats_prefetch(P_);                                                                                        \
    slice_regs(accX, IntC<P_>{});                                                                                    \
    if constexpr (MODE >= RES_LN) {                                                                                  \
      if constexpr ((P_) + 1 < NPATCH) slice_res_load((P_) + 1, m0p, n0p, (((P_) + 1) & 1) ? rres1 : rres0);         \
      f32x4& rr_ = ((P_) & 1) ? rres1 : rres0;                                                                       \
      if constexpr ((P_) + 1 < NPATCH) asm volatile("s_waitcnt vmcnt(1) ; PPRETIRE %0" : "+v"(rr_) : : "memory");                  \
      else asm volatile("s_waitcnt vmcnt(0) ; PPRETIRE %0" : "+v"(rr_) : : "memory");                                              \
      slice_rows(IntC<P_>{}, rr_);                                                                                   \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
    } else {                                                                                                         \
      slice_rows(IntC<P_>{}, rres0);                                                                                 \
    }                                                                                                                \
  }
  if constexpr (MODE >= RES_LN) asm volatile("s_waitcnt vmcnt(0) ; PPRETIRE %0" : "+v"(rres0) : : "memory");
  slice_vectors();      // (fetched again here: nothing the K loops prefetch is live across the loop exit)
  UFND_PP_LAST(0) UFND_PP_LAST(1) UFND_PP_LAST(2) UFND_PP_LAST(3) UFND_PP_LAST(4) UFND_PP_LAST(5) UFND_PP_LAST(6) UFND_PP_LAST(7)
#undef UFND_PP_LAST

  if constexpr (MODE == FOLD) {
    if (a.guard) {
      const float worst = wave_max(guard_max);
      float* gw = reinterpret_cast<float*>(smem + GRD_OFF);
      if (lane == 0) gw[wave] = worst;
      __syncthreads();
      if (threadIdx.x == 0) {
        float w = gw[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) w = fmaxf(w, gw[q]);
        atomicMax(reinterpret_cast<int*>(a.guard + (blockIdx.x & 1023)), __float_as_int(w));
      }
    }
  }
  if constexpr (DBG & 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[6] = __builtin_amdgcn_s_memtime();
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.stamps[(size_t)blockIdx.x * 8 + i] = stamp[i];
      a.stamps[(size_t)gridDim.x * 8 + blockIdx.x] = (unsigned long long)ntiles;
    }
  }
}

// Does this call fit the persistent form?  (K = 768; whole 256-row panels; bf16 output only; the residual, if any, on the bf16
// stream with row statistics out.)
static bool pp_shape_ok(int M, int N, int K) { return K == pp::NK * BK && M % pp::BM == 0 && N % pp::BN == 0 && M >= pp::BM; }
static int pp_mode_of(const GemmArgs& a) {
  if (a.out_f32 || a.residual || !a.out_bf16 || a.aux) return -1;
  if (a.a_stats) return (a.colsum && !a.residual_b && !a.out_stats && !a.r_stats) ? pp::FOLD : -1;
  if (a.residual_b) return a.out_stats ? (a.r_stats ? pp::RES_LN : pp::RES) : -1;
  return (a.out_stats || a.r_stats) ? -1 : pp::PLAIN;
}
static int pp_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    return cus;
  }();
  return n;
}
// launch; dbg = 1: the stamps build (diagnostics library only)
static int launch_pp(GemmArgs& a, int dbg, hipStream_t stream) {
  const int mode = pp_mode_of(a);
  if (mode < 0 || !pp_shape_ok(a.M, a.N, a.K)) {
    ufnd_set_error("gemm_bf16 (persistent form): unsupported call (M=%d N=%d K=%d: needs K = 768, M %% 256 == 0, N %% 128 == 0, bf16 output only, "
                   "bf16 residual stream with out_stats)", a.M, a.N, a.K);
    return UFND_ERR_INVALID;
  }
  if (mode != pp::FOLD && mode != pp::PLAIN && a.act != UFND_ACT_NONE) {
    ufnd_set_error("gemm_bf16 (persistent form): a residual call carries no activation");
    return UFND_ERR_INVALID;
  }
  a.m_tiles = a.M / pp::BM;
  a.n_tiles = a.N / pp::BN;
  a.pp_rows = 4;
  const int T = a.m_tiles * a.n_tiles;
  int G = pp_cu_count();
  G = (G < T ? G : T) & ~7;
  if (G < 8) {
    ufnd_set_error("gemm_bf16 (persistent form): %d tiles are too few", T);
    return UFND_ERR_INVALID;
  }
  const dim3 grid(G), block(512);
#define UFND_PP_LAUNCH(MODE_, ACT_)                                                                                   \
  do {                                                                                                                \
    if (dbg) {                                                                                                        \
      UFND_PP_DIAG_LAUNCH(MODE_, ACT_)                                                                                \
    } else {                                                                                                          \
      hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 0>), grid, block, 0, stream, a);                                \
    }                                                                                                                 \
  } while (0)
#ifdef UFND_DIAG
#define UFND_PP_DIAG_LAUNCH(MODE_, ACT_)                                                                      \
  if (dbg == 1) hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 1>), grid, block, 0, stream, a);            \
  else if (dbg == 5) hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 5>), grid, block, 0, stream, a);       \
  else if (dbg == 9) hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 9>), grid, block, 0, stream, a);       \
  else if (dbg == 17) hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 17>), grid, block, 0, stream, a);     \
  else if (dbg == 33) hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 33>), grid, block, 0, stream, a);     \
  else hipLaunchKernelGGL((gemm_pp_kernel<MODE_, ACT_, 3>), grid, block, 0, stream, a);
#else
#define UFND_PP_DIAG_LAUNCH(MODE_, ACT_) { ufnd_set_error("gemm_bf16 (persistent form): no stamps build in this library"); return UFND_ERR_INVALID; }
#endif
  if (mode == pp::FOLD) {
    if (a.act == UFND_ACT_GELU) UFND_PP_LAUNCH(pp::FOLD, UFND_ACT_GELU);
    else if (a.act == UFND_ACT_QUICK_GELU) UFND_PP_LAUNCH(pp::FOLD, UFND_ACT_QUICK_GELU);
    else UFND_PP_LAUNCH(pp::FOLD, UFND_ACT_NONE);
  } else if (mode == pp::RES_LN) {
    UFND_PP_LAUNCH(pp::RES_LN, UFND_ACT_NONE);
  } else if (mode == pp::RES) {
    UFND_PP_LAUNCH(pp::RES, UFND_ACT_NONE);
  } else {
    if (a.act == UFND_ACT_GELU) UFND_PP_LAUNCH(pp::PLAIN, UFND_ACT_GELU);
    else if (a.act == UFND_ACT_QUICK_GELU) UFND_PP_LAUNCH(pp::PLAIN, UFND_ACT_QUICK_GELU);
    else UFND_PP_LAUNCH(pp::PLAIN, UFND_ACT_NONE);
  }
#undef UFND_PP_LAUNCH
#undef UFND_PP_DIAG_LAUNCH
  return UFND_OK;
}

static int pp_stat_parts(int N) { return ((N / 32) % 2 == 0 && N / 32 <= 24) ? N / 32 : 0; }
// Automatic choice of the persistent form (measured on MI355X, tools/gemm_pp_bench.py, interleaved rounds, cold operands;
// profiles/r04_gemm_pp_*): it wins where the one-tile kernels' epilogue is long -- a folded LayerNorm + GELU / quick-GELU
// (FFN1: 16,384 x 3072 106 -> 101 us, 65,536 x 3072 436 -> 393 us, ViT 6,400 x 3072 46.1 -> 42.9 us) -- and a workgroup gets at
// least two tiles; with a short epilogue (Q/K/V, the residual Linears) the 256 x 256 / 256 x 192 tiles' denser K loop wins
// (66.7 vs 70.1 us, 36.7 vs 38.7 us), and at one and a half tiles per workgroup (4,096 rows) the tail does (30.6 vs 37.3 us).
// With SEVERAL tiles per workgroup and an even split the picture changes: the prologue and the one un-overlapped epilogue amortise,
// and every mode wins (profiles/r04_gemm_pp_bench.txt, last blocks): 65,536 rows Q/K/V 293 -> 268 us, LayerNorm-residual out-projection
// 145 -> 114 us; 32,768 x 768 (exactly 3 tiles per workgroup) 66.6 -> 55.0 us; 24,576 x 2304 (6.75) 109.6 -> 97.1 us -- while 16,384 x 2304
// (4.5 tiles per workgroup: a fifth round that is half empty) loses, 66.7 -> 70.1 us.  So: at least three tiles per workgroup AND at
// least 93 % of the last round filled.
static bool pp_pick(const GemmArgs& a) {
  const int mode = pp_mode_of(a);
  if (mode < 0 || !pp_shape_ok(a.M, a.N, a.K) || (mode != pp::FOLD && mode != pp::PLAIN && a.act != UFND_ACT_NONE)) return false;
  const long long tiles = (long long)(a.M / pp::BM) * (a.N / pp::BN);
  if (mode == pp::FOLD && a.act != UFND_ACT_NONE && tiles >= pp::MIN_TILES) return true;
  const long long G = pp_cu_count() & ~7, rounds = (tiles + G - 1) / G;
  return G > 0 && tiles >= 3 * G && tiles * 100 >= rounds * G * 93;
}

}  // namespace
