// Runtime pieces of the step that are not kernels: compute-unit partitions for the step's concurrent chains.
//
// The reference runs its whole step on one default queue (src/training/forensic_trainer.py:285-298: forward,
// backward, clip, AdamW in program order on one device).  Here the frozen text encoder, the frozen visual encoder
// and the head -> exchange -> optimizer chain are three independent chains on three HIP streams; left to
// themselves their whole-CU GEMM workgroups fight for the same 256 compute units.  A stream created with a CU
// mask only ever runs on its own share of the chip, so every chain runs at its stand-alone speed.
#include <hip/hip_ext.h>

#include "common.hpp"

// Stream restricted to the compute units whose bits are set in mask_words (n_words x 32 bits, bit b of the whole
// mask = logical CU b as the driver numbers them -- ufnd_diag_where() in the diagnostics library reports where
// workgroups of a masked stream actually ran).  *stream_out is a hipStream_t.
extern "C" int ufnd_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out) {
  UFND_REQUIRE(mask_words && stream_out && n_words >= 1 && n_words <= 32, "stream_create_cu_mask: bad argument");
  bool any = false;
  for (int i = 0; i < n_words; ++i) any = any || mask_words[i] != 0;
  UFND_REQUIRE(any, "stream_create_cu_mask: empty mask");
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words);
  if (e != hipSuccess) {
    ufnd_set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
    return UFND_ERR_LAUNCH;
  }
  *stream_out = (void*)s;
  return UFND_OK;
}

extern "C" int ufnd_stream_destroy(void* stream) {
  UFND_REQUIRE(stream, "stream_destroy: null stream");
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    ufnd_set_error("hipStreamDestroy failed: %s", hipGetErrorString(e));
    return UFND_ERR_LAUNCH;
  }
  return UFND_OK;
}

// Compute units of the current device (256 on MI355X), or a negative value on error.
extern "C" int ufnd_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}
