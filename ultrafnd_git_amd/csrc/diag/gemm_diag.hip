// Diagnostics build of the bf16 GEMM (libultrafnd_hip_diag.so; never loaded by the product package): every tile of
// the table, the timing-only ablation kernels (no MFMA / no in-loop DMA: results are garbage), the in-kernel stamps
// build and the UFND_GEMM_FORCE_CFG override.  Used by tools/gemm_sweep.py and tools/gemm_stamps.py.
#define UFND_DIAG 1
#include "../gemm_bf16_kernel.hpp"

// out = act(A W^T + bias) + residual with an explicit tile; tile_cfg + 100 / + 200 select the timing-only ablations.
extern "C" int ufnd_diag_gemm_bf16_ex(const void* A, const void* W, const float* bias, const float* residual, void* out_bf16,
                                      float* out_f32, int M, int N, int K, int lda, int ldw, int ldr, int ldo, int ldf, int act,
                                      int tile_cfg, void* stream_) {
  UFND_REQUIRE(A && W && (out_bf16 || out_f32), "diag gemm: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "diag gemm: M=%d N=%d K=%d", M, N, K);
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, lda, ldw, ldr, ldo, ldf, act, 0, 0, nullptr};
  int mode = 0, cfg = tile_cfg < 0 ? auto_cfg(M, N, K) : tile_cfg;
  if (cfg >= 200) { mode = 2; cfg -= 200; } else if (cfg >= 100) { mode = 1; cfg -= 100; }
  UFND_REQUIRE(cfg < kNumTiles, "diag gemm: unknown tile config %d", cfg);
  UFND_REQUIRE(N % kTiles[cfg].bn == 0, "diag gemm: tile config %d needs N %% %d == 0", cfg, kTiles[cfg].bn);
  int rc = launch_cfg(cfg, mode, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// One launch of tile `tile_cfg` built with in-kernel clock stamps.  stamps receives 8 uint64 per block: {s_memtime,
// s_memrealtime} at kernel entry, after the first K-step has landed, after the K loop, after the last store has drained.
// ln != NULL times the LayerNorm-aware kernel of that tile (bias, residual, out_f32 as in ufnd_gemm_bf16_ln; strides = N).
extern "C" int ufnd_diag_gemm_bf16_stamps(const void* A, const void* W, void* out_bf16, int M, int N, int K, int tile_cfg,
                                     unsigned long long* stamps, const ufnd_gemm_ln* ln, const float* bias, const float* residual,
                                     float* out_f32, void* stream_) {
  UFND_REQUIRE(A && W && out_bf16 && stamps, "gemm_bf16_stamps: null operand");
  UFND_REQUIRE(M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0, "gemm_bf16_stamps: M=%d N=%d K=%d", M, N, K);
  UFND_REQUIRE(tile_cfg >= 0 && tile_cfg < kNumTiles && N % kTiles[tile_cfg].bn == 0, "gemm_bf16_stamps: tile config %d", tile_cfg);
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, residual, (__bf16*)out_bf16, out_f32, M, N, K, K, K, N, N, N, 0, 0, 0, stamps};
  int abl = 3;
  if (ln) {      // the LayerNorm-aware kernel of this tile, same extras as ufnd_gemm_bf16_ln (unchecked: diagnostics)
    UFND_REQUIRE(kTiles[tile_cfg].lnx, "gemm_bf16_stamps: tile %d has no LayerNorm-aware kernel", tile_cfg);
    a.a_stats = ln->a_stats; a.colsum = ln->colsum; a.r_stats = ln->r_stats; a.r_gamma = ln->r_gamma; a.r_beta = ln->r_beta;
    a.out_stats = ln->out_stats; a.a_parts = ln->a_parts; a.r_parts = ln->r_parts; a.a_eps = ln->a_eps; a.r_eps = ln->r_eps;
    a.inv_h = 1.0f / (float)ln->width;
    a.residual_b = (const __bf16*)ln->residual_bf16;      // bf16 residual stream (round 3)
    a.ldrb = ln->ldrb;
    a.guard = ln->a_stats ? ln->guard : nullptr;          // fold guard inside the GEMM (ABI v4)
    if (ln->tile_cfg > 0) a.act = ln->tile_cfg;           // (this entry takes the tile as an argument: the field carries the activation)
    abl = 5;
  }
  int rc = launch_cfg(tile_cfg, abl, a, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// The fused projection + attention kernel with stamps: 10 uint64 per workgroup -- {s_memtime, s_memrealtime} at entry, first
// K-step landed, K loop done, END, and (slots 8, 9) projection epilogue done / attention starts.
extern "C" int ufnd_diag_qkv_attention_stamps(const void* X, const void* Wqkv, const float* bqkv, const int32_t* key_mask, void* ctx, int B,
                                              int heads, const ufnd_gemm_ln* ln, unsigned long long* stamps, void* stream_) {
  UFND_REQUIRE(X && Wqkv && ctx && stamps && heads % 2 == 0, "diag qkv_attention: bad argument");
  const int H = heads * 64;
  GemmArgs a{(const __bf16*)X, (const __bf16*)Wqkv, bqkv, nullptr, nullptr, nullptr, B * 128, 3 * H, H, H, H, 0, 0, 0, UFND_ACT_NONE, 0, 0, stamps};
  if (ln && ln->a_stats) {
    a.a_stats = ln->a_stats; a.colsum = ln->colsum; a.a_parts = ln->a_parts; a.a_eps = ln->a_eps;
    a.inv_h = 1.0f / (float)ln->width;
    a.guard = ln->guard;
  }
  a.att_mask = key_mask;
  a.att_ctx = (__bf16*)ctx;
  a.att_h = H;
  a.att_scale_log2e = 0.125f * 1.44269504088896340736f;
  a.m_tiles = B;
  a.n_tiles = heads / 2;
  a.xcd_cols = (a.n_tiles % 2 == 0 && a.m_tiles >= 4) ? 2 : 1;
  hipLaunchKernelGGL((gemm_bf16_kernel<128, 384, 2, 4, 3, 2, 16, 0, 1, 1, 1>), dim3(a.m_tiles * a.n_tiles), dim3(512), 0, (hipStream_t)stream_, a);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
