// One 64-key block of the online softmax, for one 16-query tile column of a wave (attention.hip and the fused projection +
// attention kernel of gemm_bf16_kernel.hpp share it: the two must produce the same bits).
//
// Scores arrive as the S^T accumulators s[kt][qt] (row = key 16 kt + 4 g + r, column = the lane's query); on return pf holds
// exp2(score - m_new) rounded to bf16 in the B-operand order of the P V product, m_run / l_run are updated, and the factor the
// running output must be multiplied by is returned.
//
// The softmax is VALU-bound (32 scores per lane per block against 32 MFMAs per wave), so the common case -- a block WITHOUT a
// masked or out-of-range key (`any_masked` false: wave-uniform, from the block's key-bias words) -- takes the short path: the maximum
// over the raw accumulators, then ONE fused multiply-add and ONE v_exp_f32 per score (exp2(s * scale - m), scale > 0 commutes with
// the maximum).  Blocks with a masked key keep HF's semantics (masked score = the finfo.min-like constant, so a fully masked row
// degenerates to a uniform average) at three more instructions per score.
#pragma once
#include "common.hpp"

// A 64-key block whose keys are ALL masked (or beyond L) changes nothing once every query of the wave has seen a live key: its scores
// are the -3e38 constant (or -inf), so m_new = m_run, alpha = 1, every p = exp2(-3e38 - m_run) = +0 exactly, l and O stay as they are.
// Wave-uniform test for that case (kb = the block's 64 key-bias words, one per lane): the wave then skips the block's 32 + 32 MFMAs
// and its softmax -- bit-identical output.  Padding is a suffix of masked keys, so for a batch of ragged lengths this is most of
// the key blocks of the short samples.  A wave whose queries have seen only masked keys so far (m_run still at the mask constant:
// HF's uniform-average degenerate case) does NOT skip.
__device__ __forceinline__ bool masked_block_is_noop(const float* kb, int lane, const float (&m_run)[2]) {
  return __all(kb[lane] != 0.0f) && __all(m_run[0] > -1.0e30f && m_run[1] > -1.0e30f);
}

template <int KT>
__device__ __forceinline__ float online_softmax_block(f32x4 (&s)[KT][2], const int qt, const float* kbias, const bool any_masked, const int g,
                                                      const float scale_log2e, float& m_run, float& l_run, bf16x8 (&pf)[KT / 2][2]) {
#pragma clang fp contract(off)
  float mx = -INFINITY, lsum = 0.0f, m_new, alpha;
  if (any_masked) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const f32x4 kbv = *reinterpret_cast<const f32x4*>(kbias + kt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (kbv[r] == 0.0f) ? s[kt][qt][r] * scale_log2e : kbv[r];
        s[kt][qt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    m_new = fmaxf(m_run, mx);                    // finite: every block holds a key < L
    alpha = fast_exp2(m_run - m_new);            // first block: exp2(-inf) = 0
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(s[kt][qt][r] - m_new);
        lsum += p;
        pf[kt >> 1][qt][(kt & 1) * 4 + r] = (__bf16)p;
      }
  } else {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    m_new = fmaxf(m_run, mx * scale_log2e);
    alpha = fast_exp2(m_run - m_new);
    const float neg_m = -m_new;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[kt][qt][r], scale_log2e, neg_m));
        lsum += p;
        pf[kt >> 1][qt][(kt & 1) * 4 + r] = (__bf16)p;
      }
  }
  l_run = __builtin_fmaf(l_run, alpha, lsum);
  m_run = m_new;
  return alpha;
}
