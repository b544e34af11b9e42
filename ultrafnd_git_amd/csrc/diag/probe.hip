// Diagnostics: where did the workgroups of a launch run?  (libultrafnd_hip_diag.so only.)
#include "../common.hpp"

namespace {
// one workgroup per CU (all of the CU's LDS), every workgroup stays resident for `spin` shader cycles so that the
// set of CUs seen is the set the stream may use
__global__ __launch_bounds__(256) void where_kernel(uint32_t* out, unsigned long long spin) {
  __shared__ int big[40 * 1024 - 64];
  big[threadIdx.x] = (int)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw + (uint32_t)(big[17] - 17);
    out[2 * blockIdx.x + 1] = xcc;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

// out: 2 x blocks uint32 {HW_ID, XCC_ID} of every workgroup; spin_cycles is capped at 2^26 (about 30 ms).
extern "C" int ufnd_diag_where(uint32_t* out, int blocks, unsigned long long spin_cycles, void* stream) {
  UFND_REQUIRE(out && blocks >= 1 && blocks <= 4096, "diag_where: bad argument");
  if (spin_cycles > (1ull << 26)) spin_cycles = 1ull << 26;
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, spin_cycles);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
