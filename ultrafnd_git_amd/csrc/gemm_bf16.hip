// bf16 GEMM for the encoder Linears -- the product entry points (include/ultrafnd_hip.h).  Kernel, tile table and
// launcher: gemm_bf16_kernel.hpp.  No environment overrides, no ablation kernels, no stamps here: those live in
// diag/gemm_diag.hip (libultrafnd_hip_diag.so).
#include "gemm_bf16_kernel.hpp"

// the persistent, software-pipelined form (gemm_bf16_pp.hpp), compiled in gemm_bf16_pp.hip
#define UFND_GEMM_TILE_PP UFND_GEMM_TILE_PERSISTENT
int ufnd_pp_pick(const void* gemm_args);
int ufnd_pp_launch(void* gemm_args, void* stream);

namespace {
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
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0, nullptr};
  if (tile_cfg == UFND_GEMM_TILE_PP || (tile_cfg < 0 && ufnd_pp_pick(&a))) return ufnd_pp_launch(&a, stream_);      // the persistent, software-pipelined form
  const int cfg = tile_cfg < 0 ? auto_cfg(M, N, K) : tile_cfg;
  UFND_REQUIRE(cfg < kNumTiles && kTiles[cfg].built, "gemm_bf16: tile config %d is not part of this library (ufnd_gemm_bf16_tile_info)", cfg);
  UFND_REQUIRE(N % kTiles[cfg].bn == 0, "gemm_bf16: tile config %d needs N %% %d == 0", cfg, kTiles[cfg].bn);
  int rc = launch_cfg(cfg, 0, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_gemm_bf16_tile_info(int tile_cfg, int* bm, int* bn, int* ln_aware) {
  if (tile_cfg < 0 || tile_cfg >= kNumTiles || !kTiles[tile_cfg].built) return 0;
  if (bm) *bm = kTiles[tile_cfg].bm;
  if (bn) *bn = kTiles[tile_cfg].bn;
  if (ln_aware) *ln_aware = kTiles[tile_cfg].lnx;
  return 1;
}
extern "C" int ufnd_gemm_bf16_tile_count(void) { return kNumTiles; }

static int stat_parts_for(int cfg, int N) {
  const TileCfg& t = kTiles[cfg];
  const int tn = t.bn / t.wn;
  if (!t.built || !t.lnx || N % t.bn != 0 || tn % 32 != 0 || (N / 32) % 2 != 0 || N / 32 > 24) return 0;
  return N / 32;
}
extern "C" int ufnd_gemm_bf16_stat_parts(int M, int N, int K) {
  return stat_parts_for(auto_cfg(M, N, K), N);      // (the persistent form is never the automatic choice of a call with out_stats)
}

extern "C" int ufnd_gemm_bf16_ln(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                                 float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                                 const ufnd_gemm_ln* ln, void* stream_) {
  UFND_REQUIRE(A && W && ln && (out_bf16 || out_f32), "gemm_bf16_ln: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "gemm_bf16_ln: M=%d N=%d K=%d (need N%%64==0, K%%64==0)", M, N, K);
  UFND_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ufnd_aligned(A, 16) && ufnd_aligned(W, 16),
               "gemm_bf16_ln: A/W strides must be multiples of 8 and pointers 16-B aligned");
  UFND_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= N && ufnd_aligned(residual, 16)), "gemm_bf16_ln: residual alignment");
  UFND_REQUIRE(!out_f32 || (ldf % 4 == 0 && ldf >= N && ufnd_aligned(out_f32, 16)), "gemm_bf16_ln: out_f32 alignment");
  UFND_REQUIRE(!out_bf16 || (ldo % 8 == 0 && ldo >= N && ufnd_aligned(out_bf16, 16)), "gemm_bf16_ln: out_bf16 alignment");
  UFND_REQUIRE(!bias || ufnd_aligned(bias, 16), "gemm_bf16_ln: bias must be 16-B aligned");
  UFND_REQUIRE(act >= 0 && act <= 2, "gemm_bf16_ln: act=%d", act);
  UFND_REQUIRE(!(ln->a_stats && ln->r_stats), "gemm_bf16_ln: a_stats and r_stats are mutually exclusive");
  UFND_REQUIRE(!(residual && ln->residual_bf16), "gemm_bf16_ln: residual (fp32) and residual_bf16 are mutually exclusive");
  UFND_REQUIRE(!ln->residual_bf16 || (ln->ldrb % 8 == 0 && ln->ldrb >= N && ufnd_aligned(ln->residual_bf16, 16)), "gemm_bf16_ln: residual_bf16 alignment");
  UFND_REQUIRE(ln->a_stats || act == UFND_ACT_NONE, "gemm_bf16_ln: an activation is only fused together with a folded LayerNorm (a_stats)");
  UFND_REQUIRE(ln->width > 0, "gemm_bf16_ln: width (the LayerNorm dimension) must be positive");
  if (ln->a_stats) {
    UFND_REQUIRE(!residual && !ln->residual_bf16 && !ln->out_stats,
                 "gemm_bf16_ln: a folded LayerNorm (a_stats) takes no residual and writes no out_stats (that epilogue is compiled without them)");
    UFND_REQUIRE(ln->colsum && ufnd_aligned(ln->colsum, 16) && ufnd_aligned(ln->a_stats, 16), "gemm_bf16_ln: colsum / a_stats alignment");
    UFND_REQUIRE(ln->a_parts >= 2 && ln->a_parts <= 24 && ln->a_parts % 2 == 0, "gemm_bf16_ln: a_parts=%d (even, 2..24)", ln->a_parts);
  }
  if (ln->r_stats) {
    UFND_REQUIRE((residual || ln->residual_bf16) && ln->r_gamma && ln->r_beta && ufnd_aligned(ln->r_gamma, 16) && ufnd_aligned(ln->r_beta, 16) &&
                     ufnd_aligned(ln->r_stats, 16), "gemm_bf16_ln: r_stats needs residual, r_gamma, r_beta (16-B aligned)");
    UFND_REQUIRE(ln->r_parts >= 2 && ln->r_parts <= 24 && ln->r_parts % 2 == 0, "gemm_bf16_ln: r_parts=%d (even, 2..24)", ln->r_parts);
  }
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0, nullptr};
  a.a_stats = ln->a_stats; a.colsum = ln->colsum; a.r_stats = ln->r_stats; a.r_gamma = ln->r_gamma; a.r_beta = ln->r_beta;
  a.out_stats = ln->out_stats; a.a_parts = ln->a_parts; a.r_parts = ln->r_parts; a.a_eps = ln->a_eps; a.r_eps = ln->r_eps;
  a.inv_h = 1.0f / (float)ln->width;
  a.residual_b = (const __bf16*)ln->residual_bf16;
  a.ldrb = ln->ldrb;
  a.guard = ln->a_stats ? ln->guard : nullptr;
  if (ln->tile_cfg == UFND_GEMM_TILE_PP || (ln->tile_cfg < 0 && ufnd_pp_pick(&a))) return ufnd_pp_launch(&a, stream_);      // the persistent, software-pipelined form
  const int cfg = ln->tile_cfg < 0 ? auto_cfg(M, N, K) : ln->tile_cfg;
  UFND_REQUIRE(cfg < kNumTiles && kTiles[cfg].built && kTiles[cfg].lnx && N % kTiles[cfg].bn == 0,
               "gemm_bf16_ln: no LayerNorm-aware kernel for M=%d N=%d K=%d (tile %d)", M, N, K, cfg);
  if (ln->out_stats) {
    UFND_REQUIRE(stat_parts_for(cfg, N) > 0 && ufnd_aligned(ln->out_stats, 16), "gemm_bf16_ln: out_stats unsupported for this shape / tile");
  }
  int rc = launch_cfg(cfg, 4, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// BertSelfAttention of one layer in ONE launch: fused Q/K/V projection (optionally of LayerNorm(X), folded) + attention.
extern "C" int ufnd_qkv_attention_bf16(const void* X, const void* Wqkv, const float* bqkv, const int32_t* key_mask, void* ctx,
                                       int B, int L, int heads, int ldx, int ldw, const ufnd_gemm_ln* ln, void* stream_) {
  UFND_REQUIRE(X && Wqkv && ctx, "qkv_attention: null operand");
  UFND_REQUIRE(L == 128 && heads >= 2 && heads % 2 == 0 && heads <= 64 && B >= 1 && B <= 16384,
               "qkv_attention: B=%d L=%d heads=%d (this kernel is built for 128-token samples and an even head count; "
               "use ufnd_gemm_bf16[_ln] + ufnd_attention_bf16 otherwise)", B, L, heads);
  const int H = heads * 64;
  UFND_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= H && ldw >= H && ufnd_aligned(X, 16) && ufnd_aligned(Wqkv, 16) && ufnd_aligned(ctx, 16),
               "qkv_attention: strides must be multiples of 8 and pointers 16-B aligned");
  UFND_REQUIRE(!bqkv || ufnd_aligned(bqkv, 16), "qkv_attention: bias alignment");
  GemmArgs a{(const __bf16*)X, (const __bf16*)Wqkv, bqkv, nullptr, nullptr, nullptr, B * L, 3 * H, H, ldx, ldw, 0, 0, 0, UFND_ACT_NONE, 0, 0, nullptr};
  if (ln && ln->a_stats) {
    UFND_REQUIRE(ln->colsum && ufnd_aligned(ln->colsum, 16) && ufnd_aligned(ln->a_stats, 16), "qkv_attention: colsum / a_stats alignment");
    UFND_REQUIRE(ln->a_parts >= 2 && ln->a_parts <= 24 && ln->a_parts % 2 == 0 && ln->width > 0, "qkv_attention: a_parts=%d width=%d", ln->a_parts, ln->width);
    a.a_stats = ln->a_stats; a.colsum = ln->colsum; a.a_parts = ln->a_parts; a.a_eps = ln->a_eps;
    a.inv_h = 1.0f / (float)ln->width;
    a.guard = ln->guard;
  }
  a.att_mask = key_mask;
  a.att_ctx = (__bf16*)ctx;
  a.att_h = H;
  a.att_scale_log2e = 0.125f * 1.44269504088896340736f;      // 1 / sqrt(64) * log2(e)
  a.m_tiles = B;
  a.n_tiles = heads / 2;
  a.xcd_cols = (a.n_tiles % 2 == 0 && a.m_tiles >= 4) ? 2 : 1;
  hipLaunchKernelGGL((gemm_bf16_kernel<128, 384, 2, 4, 3, 2, 16, 0, 0, 1, 1>), dim3(a.m_tiles * a.n_tiles), dim3(512), 0, (hipStream_t)stream_, a);
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
