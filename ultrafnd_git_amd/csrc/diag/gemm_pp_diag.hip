// Diagnostics build of the persistent bf16 GEMM (libultrafnd_hip_diag.so): stamps and timing-only ablations.  Compiled like
// gemm_bf16_pp.hip, without packed fp32 instructions.
#define UFND_DIAG 1
#include "../gemm_bf16_pp.hpp"

// The persistent form (gemm_bf16_pp.hpp) with stamps: 8 uint64 per workgroup -- {s_memtime, s_memrealtime} at entry, ring
// prologue landed, last K loop done, END -- followed by one uint64 per workgroup: its tile count (stamps holds 9 * 256 words).
// dbg = 1: stamps; dbg = 3: stamps on the timing-only build without epilogue slices (results are garbage).
extern "C" int ufnd_diag_gemm_pp_stamps(const void* A, const void* W, void* out_bf16, int M, int N, int K, unsigned long long* stamps,
                                        const ufnd_gemm_ln* ln, const float* bias, int act, int dbg, void* stream_) {
  UFND_REQUIRE(A && W && out_bf16 && stamps, "gemm_pp_stamps: null operand");
  GemmArgs a{(const __bf16*)A, (const __bf16*)W, bias, nullptr, (__bf16*)out_bf16, nullptr, M, N, K, K, K, N, N, N, act, 0, 0, stamps};
  if (ln) {
    a.a_stats = ln->a_stats; a.colsum = ln->colsum; a.r_stats = ln->r_stats; a.r_gamma = ln->r_gamma; a.r_beta = ln->r_beta;
    a.out_stats = ln->out_stats; a.a_parts = ln->a_parts; a.r_parts = ln->r_parts; a.a_eps = ln->a_eps; a.r_eps = ln->r_eps;
    a.inv_h = 1.0f / (float)ln->width;
    a.residual_b = (const __bf16*)ln->residual_bf16;
    a.ldrb = ln->ldrb;
    a.guard = ln->a_stats ? ln->guard : nullptr;
  }
  int rc = launch_pp(a, dbg, (hipStream_t)stream_);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
