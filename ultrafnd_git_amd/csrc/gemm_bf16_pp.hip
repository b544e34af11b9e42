// The persistent, software-pipelined bf16 GEMM as a translation unit of its own: it is compiled WITHOUT packed fp32
// instructions (build.py: -target-feature -packed-fp32-ops for this file only).  Its epilogue slices are fillers between MFMAs,
// where one v_pk_fma_f32 costs +22 cycles against two v_fma_f32 (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the
// one-tile-per-workgroup kernels of gemm_bf16.hip keep the packed forms, which are the faster ones in a stand-alone epilogue.
// (A function-level target attribute would do the same, but it also stops the kernel's lambdas from being inlined into it.)
#include "gemm_bf16_pp.hpp"

// internal interface to gemm_bf16.hip (GemmArgs lives in an anonymous namespace of a shared header: same layout in both units)
__attribute__((visibility("hidden"))) int ufnd_pp_pick(const void* gemm_args) { return pp_pick(*static_cast<const GemmArgs*>(gemm_args)) ? 1 : 0; }
__attribute__((visibility("hidden"))) int ufnd_pp_launch(void* gemm_args, void* stream) {
  GemmArgs& a = *static_cast<GemmArgs*>(gemm_args);
  if (a.out_stats && (pp_stat_parts(a.N) == 0 || !ufnd_aligned(a.out_stats, 16))) {
    ufnd_set_error("gemm_bf16 (persistent form): out_stats unsupported for this shape");
    return UFND_ERR_INVALID;
  }
  int rc = launch_pp(a, 0, (hipStream_t)stream);
  if (rc != UFND_OK) return rc;
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
