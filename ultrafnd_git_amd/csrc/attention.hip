// Multi-head self-attention, forward only: ctx = softmax(Q K^T / sqrt(64) + mask) V.
// (BertSelfAttention eager path / CLIPAttention; head_dim 64, 12 heads in both encoders.)
//
// One workgroup = one (batch, head, 128-query block); wave w owns 32 queries.  Keys/values are
// walked in blocks of 64 with an online softmax, so L = 50 (ViT), 128 and 512 (BERT) share
// the code and nothing of size L x L ever exists.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16; C/D: column = lane & 15, row = 4*(lane>>4)+r):
//   S^T[key][query] = K Q^T      A = K rows from LDS (ds_read_b128, swizzled), B = Q from registers
//     -> a lane holds ONE query (its column) and 4 keys per tile: the softmax statistics of a
//        query live in the 4 lanes {q, q+16, q+32, q+48}: two __shfl_xor, no LDS.
//   O^T[d][query] = V^T P^T      B = P^T: the S^T accumulators, converted to bf16 IN PLACE --
//        the accumulator layout of step 1 is exactly the B-operand layout of step 2 once the
//        contraction index is ordered as (keys 4g..4g+3 of tile 2s, keys 4g..4g+3 of tile 2s+1)
//        A = V^T fragments in that same key order, produced from the row-major V tile in LDS
//        by ds_read_b64_tr_b16 (hardware transpose): 4 keys x 16 d per 16-lane group.
// Masking follows HF: masked keys get the constant finfo.min-like score (a fully masked row
// degenerates to a uniform average, not NaN); keys beyond L contribute exactly zero.
#include "attn_softmax.hpp"

// No implicit contraction: the fused projection + attention kernel (gemm_bf16_kernel.hpp, ATT) repeats this kernel's
// arithmetic operation for operation and must produce the same bits; fused multiply-adds are spelled out.
#pragma clang fp contract(off)

namespace {

constexpr int QB = 128;          // queries per workgroup (4 waves x 32; the short-sequence form: 2 waves, 64 queries)
constexpr float NEG_MASK = -3.0e38f;

__device__ __forceinline__ bf16x8 k_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// KB = keys per block (64: 162 registers, three workgroups per CU; 128: 240 registers, two -- measured at L = 512, B = 128 in round 3:
// 218 against 211 us per launch, not selected)
// NW = waves per workgroup (32 queries each).  Sequences of <= 64 tokens (ViT-B/32: 50) run with NW = 2: the 4-wave form spends
// half its waves on clamped duplicate queries there, and its 216 registers allow 8 waves per CU either way -- twice the
// samples in flight with 2-wave workgroups (a block is one dependent chain: loads -> S -> softmax -> O -> store).
template <int KB, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 2 ? 3 : 2))) void attention_kernel(const __bf16* qkv, const int32_t* mask, __bf16* ctx, int L,
                                                        int heads, float scale_log2e, const int32_t* cu, float* lse, int nqb) {
  __shared__ __attribute__((aligned(16))) char ks[KB * 128];
  __shared__ __attribute__((aligned(16))) char vs[KB * 128];
  __shared__ __attribute__((aligned(16))) float kbias[KB];

  // 1-D grid of (query block, head, sample), query block fastest, in XCD-contiguous order: at L = 512 the four query blocks of a
  // (sample, head) each walk the same 128 KB of K / V -- dealt round-robin over the XCDs they fetched it four times over the
  // fabric (1.07 GB per launch at B = 128: the launch ran at the fabric's rate, not the matrix pipes')
  const int lid = xcd_contiguous_id((int)blockIdx.x, (int)gridDim.x);
  const int qb = lid % nqb, h = (lid / nqb) % heads, b = lid / (nqb * heads);
  const int H = heads * 64, ld = 3 * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, g = lane >> 4;
  size_t tok0 = (size_t)b * L;
  if (cu) {              // packed (un-padded) sequences: rows cu[b] .. cu[b+1], every key valid
    tok0 = (size_t)cu[b];
    L = cu[b + 1] - cu[b];
    if (qb * (NW * 32) >= L) return;      // (block-uniform, before any barrier)
  }

  // Q fragments (B operand): lane (query fr, g) holds Q[query][8g + 32kk .. +7]
  bf16x8 qf[2][2];
  int qrow[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qb * (NW * 32) + wave * 32 + qt * 16 + fr;
    qrow[qt] = q;
    const int qc = q < L ? q : L - 1;
    const __bf16* src = qkv + (tok0 + qc) * ld + h * 64 + 8 * g;
    qf[qt][0] = *reinterpret_cast<const bf16x8*>(src);
    qf[qt][1] = *reinterpret_cast<const bf16x8*>(src + 32);
  }

  f32x4 o[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  // K / V blocks travel HBM -> registers -> LDS.  With several blocks to walk (the 4-wave form: L > 64) the loads of block k+1
  // are issued right after block k has been written to LDS and stay in flight under block k's S^T / softmax / PV work (the
  // issue-early / write-late split: their latency was exposed in front of every block before); one register set, +16 VGPRs.
  constexpr int NPC = KB * 8 / (NW * 64);       // 16-B pieces per thread per matrix
  constexpr bool PREFETCH = NW == 4;
  bf16x8 kreg[NPC], vreg[NPC];
  float breg = -INFINITY;
  auto load_block = [&](int kb0) {
#pragma unroll
    for (int it = 0; it < NPC; ++it) {
      const int idx = tid + NW * 64 * it, key = idx >> 3, c = idx & 7;
      const int kr = (kb0 + key) < L ? (kb0 + key) : L - 1;
      const __bf16* src = qkv + (tok0 + kr) * ld + H + h * 64 + c * 8;
      kreg[it] = *reinterpret_cast<const bf16x8*>(src);
      vreg[it] = *reinterpret_cast<const bf16x8*>(src + H);
    }
    if (tid < KB) {
      const int key = kb0 + tid;
      breg = -INFINITY;
      if (key < L) breg = (!mask || mask[tok0 + key] != 0) ? 0.0f : NEG_MASK;
    }
  };
  if constexpr (PREFETCH) load_block(0);
  for (int kb0 = 0; kb0 < L; kb0 += KB) {
    __syncthreads();  // previous block's LDS reads are done
    // ---- stage K and V tiles (row-major 128-B rows, swizzled 16-B chunks) and the key bias
    if constexpr (!PREFETCH) load_block(kb0);
#pragma unroll
    for (int it = 0; it < NPC; ++it) {
      const int idx = tid + NW * 64 * it, key = idx >> 3, c = idx & 7;
      *reinterpret_cast<bf16x8*>(ks + key * 128 + ((c ^ ((key >> 1) & 7)) << 4)) = kreg[it];
      *reinterpret_cast<bf16x8*>(vs + key * 128 + ((c ^ (((key >> 1) & 3) << 1)) << 4)) = vreg[it];
    }
    if (tid < KB) kbias[tid] = breg;
    __syncthreads();
    if constexpr (PREFETCH) {
      if (kb0 + KB < L) load_block(kb0 + KB);
    }

    if constexpr (KB == 64) {
      if (masked_block_is_noop(kbias, lane, m_run)) continue;      // (attn_softmax.hpp; both barriers of the iteration are behind us / at the loop top)
    }
    // ---- S^T = K Q^T : 8 key tiles x 2 query tiles
    constexpr int KT = KB / 16;
    f32x4 s[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const bf16x8 kf = k_frag(ks, kt * 16 + fr, g + 4 * kk);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][kk], s[kt][qt], 0, 0, 0);
      }

    // ---- online softmax (scores kept in the log2 domain: exp(x) = exp2(x * log2 e)); attn_softmax.hpp
    bool any_masked = __any(kbias[lane] != 0.0f);            // (the block's key-bias words, 64 per pass; wave-uniform)
    if constexpr (KB == 128) any_masked = any_masked || __any(kbias[64 + lane] != 0.0f);
    bf16x8 pf[KT / 2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float alpha = online_softmax_block<KT>(s, qt, kbias, any_masked, g, scale_log2e, m_run[qt], l_run[qt], pf);
      if (!__all(alpha == 1.0f)) {       // (the running maximum rarely moves after the first blocks: skip the rescale then)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
      }
    }

    // ---- O^T += V^T P^T : contraction over the block's 128 keys in 4 steps of 32
#pragma unroll
    for (int ksd = 0; ksd < KT / 2; ++ksd) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // hardware-transposed read: 16-lane group g, lane 4q+p supplies &V[key][16dt + 4p]
        const int qq = fr >> 2, pp = fr & 3;
        const int key0 = 32 * ksd + 4 * g + qq;
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

  // ---- normalise and store: lane holds O^T[d = 16dt + 4g + r][query fr]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // training forwards keep the query's log-sum-exp (log2 domain, as the scores above): the backward recomputes P from it
    if (lse && g == 0 && qrow[qt] < L) lse[(tok0 + qrow[qt]) * heads + h] = m_run[qt] + log2f(l);
    if (qrow[qt] < L) {
      __bf16* dst = ctx + (tok0 + qrow[qt]) * H + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(__bf16)(o[dt][qt][0] * inv), (__bf16)(o[dt][qt][1] * inv), (__bf16)(o[dt][qt][2] * inv),
                     (__bf16)(o[dt][qt][3] * inv)};
        *reinterpret_cast<bf16x4*>(dst + dt * 16) = ov;
      }
    }
  }
}

}  // namespace

extern "C" int ufnd_attention_bf16_lse(const void* qkv, const int32_t* key_mask, void* ctx, float* lse, int B, int L, int heads,
                                       void* stream_);
extern "C" int ufnd_attention_bf16(const void* qkv, const int32_t* key_mask, void* ctx, int B, int L, int heads,
                                   void* stream_) {
  return ufnd_attention_bf16_lse(qkv, key_mask, ctx, nullptr, B, L, heads, stream_);
}

extern "C" int ufnd_attention_bf16_lse(const void* qkv, const int32_t* key_mask, void* ctx, float* lse, int B, int L, int heads,
                                       void* stream_) {
  UFND_REQUIRE(qkv && ctx, "attention: null operand");
  UFND_REQUIRE(B >= 1 && L >= 1 && L <= 4096 && heads >= 1 && heads <= 64, "attention: B=%d L=%d heads=%d", B, L, heads);
  UFND_REQUIRE(ufnd_aligned(qkv, 16) && ufnd_aligned(ctx, 16), "attention: 16-B alignment required");
  UFND_REQUIRE((long long)B * heads * ufnd_cdiv(L, QB) < (1ll << 31), "attention: grid too large");
  const float scale_log2e = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
  if (L <= 64)
    hipLaunchKernelGGL((attention_kernel<64, 2>), dim3(heads * B), dim3(128), 0, (hipStream_t)stream_,
                       (const __bf16*)qkv, key_mask, (__bf16*)ctx, L, heads, scale_log2e, (const int32_t*)nullptr, lse, 1);
  else
    hipLaunchKernelGGL((attention_kernel<64, 4>), dim3(ufnd_cdiv(L, QB) * heads * B), dim3(256), 0, (hipStream_t)stream_,
                       (const __bf16*)qkv, key_mask, (__bf16*)ctx, L, heads, scale_log2e, (const int32_t*)nullptr, lse, ufnd_cdiv(L, QB));
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_attention_bf16_varlen(const void* qkv, const int32_t* cu_seqlens, void* ctx, int B, int max_len, int heads,
                                          void* stream_) {
  UFND_REQUIRE(qkv && ctx && cu_seqlens, "attention_varlen: null operand");
  UFND_REQUIRE(B >= 1 && B <= 65535 && max_len >= 1 && max_len <= 4096 && heads >= 1 && heads <= 64, "attention_varlen: B=%d max_len=%d heads=%d",
               B, max_len, heads);
  UFND_REQUIRE(ufnd_aligned(qkv, 16) && ufnd_aligned(ctx, 16), "attention_varlen: 16-B alignment required");
  const float scale_log2e = 0.125f * 1.44269504088896340736f;
  hipLaunchKernelGGL((attention_kernel<64, 4>), dim3(ufnd_cdiv(max_len, QB) * heads * B), dim3(256), 0, (hipStream_t)stream_,
                     (const __bf16*)qkv, (const int32_t*)nullptr, (__bf16*)ctx, max_len, heads, scale_log2e, cu_seqlens, (float*)nullptr,
                     ufnd_cdiv(max_len, QB));
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}
