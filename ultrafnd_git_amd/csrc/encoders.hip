// Row-wise pieces of the two encoders: LayerNorm, BERT embeddings, masked mean-pool + L2,
// ViT patchify / token assembly, frame pooling.  All HBM-bound: one wave per row, 16-B loads,
// wave-shuffle reductions, fp32 statistics; outputs feed the bf16 GEMMs directly.
#include "common.hpp"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// normalise NI*4 values per lane (row of H = 256*NI) held in v[]; returns via v[]
template <int NI>
__device__ __forceinline__ void ln_row(f32x4 (&v)[NI], int H, float eps, const float* gamma, const float* beta, int lane) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = v[i][k] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = 4 * lane + 256 * i;
    const f32x4 gm = ld4(gamma + col), bt = ld4(beta + col);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[i][k] = (v[i][k] - mean) * rstd * gm[k] + bt[k];
  }
}

template <int NI>
__device__ __forceinline__ void store_row(const f32x4 (&v)[NI], __bf16* ob, float* of, size_t row, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = 4 * lane + 256 * i;
    if (of) *reinterpret_cast<f32x4*>(of + row * H + col) = v[i];
    if (ob) {
      bf16x4 o = {(__bf16)v[i][0], (__bf16)v[i][1], (__bf16)v[i][2], (__bf16)v[i][3]};
      *reinterpret_cast<bf16x4*>(ob + row * H + col) = o;
    }
  }
}

// {sum, sum of squares} of the row held in v[] -> stats[row] as TWO partials ({s, q}, {0, 0}): the
// layout ufnd_gemm_bf16_ln reads its a_stats in (even partial counts)
template <int NI>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[NI], float* stats, size_t row, int lane) {
  float s = 0.0f, q = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += v[i][k]; q += v[i][k] * v[i][k]; }
  s = wave_sum(s);
  q = wave_sum(q);
  if (lane == 0) *reinterpret_cast<f32x4*>(stats + row * 4) = f32x4{s, q, 0.0f, 0.0f};
}

template <int NI>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int ldx, const float* gamma, const float* beta,
                                                        __bf16* ob, float* of, int M, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) v[i] = ld4(x + (size_t)row * ldx + 4 * lane + 256 * i);
  ln_row<NI>(v, H, eps, gamma, beta, lane);
  store_row<NI>(v, ob, of, row, H, lane);
}

template <int NI>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* ids, const float* word, const float* pos,
                                                         const float* type0, const float* gamma, const float* beta,
                                                         __bf16* ob, float* of, int M, int L, int H, int vocab,
                                                         float eps, const int32_t* pos_ids) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
  int l = pos_ids ? pos_ids[row] : row % L;         // packed (un-padded) rows carry their position
  l = l < 0 ? 0 : (l >= L ? L - 1 : l);
  f32x4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = 4 * lane + 256 * i;
    v[i] = ld4(word + (size_t)id * H + col) + ld4(pos + (size_t)l * H + col) + ld4(type0 + col);
  }
  if (gamma) ln_row<NI>(v, H, eps, gamma, beta, lane);      // (gamma == NULL: the raw sums, for training forwards that keep them)
  store_row<NI>(v, ob, of, row, H, lane);
}

// masked mean over tokens (phase 1: grid (H/256, B), 256 threads = 4 token-groups x 64 lanes x 4 columns,
// LDS-reduced), then L2 normalise (phase 2, one block per sample).  text_blocks.py:82-86,100.
__global__ __launch_bounds__(256) void meanpool_kernel(const float* hidden, const int32_t* mask, float* out, int L, int H) {
  __shared__ f32x4 part[4][64];
  __shared__ float cnts[4];
  const int b = blockIdx.y, col = blockIdx.x * 256 + 4 * (threadIdx.x & 63), grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.0f;
  // eight tokens of the group per pass: their mask words and rows are requested together (a mask load -> branch -> row
  // load chain per token made this kernel 27 us for 12.6 MB); rows are added in token order, only where the mask keeps them
  for (int l0 = grp; l0 < L; l0 += 32) {
    int mk[8];
    f32x4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int l = l0 + 4 * u;
      mk[u] = l < L ? mask[(size_t)b * L + l] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int l = l0 + 4 * u < L ? l0 + 4 * u : L - 1;
      x[u] = ld4(hidden + ((size_t)b * L + l) * H + col);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (mk[u] != 0) {
        acc += x[u];
        cnt += 1.0f;
      }
    }
  }
  part[grp][lane] = acc;
  if (lane == 0) cnts[grp] = cnt;
  __syncthreads();
  if (grp == 0) {
    const f32x4 s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const float denom = fmaxf((cnts[0] + cnts[1]) + (cnts[2] + cnts[3]), 1e-6f);
    *reinterpret_cast<f32x4*>(out + (size_t)b * H + col) = s / denom;
  }
}
// the same reduction over a PACKED sequence (rows cu[b] .. cu[b+1]): a token joins the group its original position
// selects and groups add in token order, so the sum is bit-identical to the padded kernel's
__global__ __launch_bounds__(256) void meanpool_packed_kernel(const float* hidden, const int32_t* cu, const int32_t* pos_ids, float* out,
                                                              int H) {
  __shared__ f32x4 part[4][64];
  __shared__ float cnts[4];
  const int b = blockIdx.y, col = blockIdx.x * 256 + 4 * (threadIdx.x & 63), grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = cu[b], r1 = cu[b + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.0f;
  for (int r = r0; r < r1; ++r) {
    if ((pos_ids[r] & 3) == grp) {
      acc += ld4(hidden + (size_t)r * H + col);
      cnt += 1.0f;
    }
  }
  part[grp][lane] = acc;
  if (lane == 0) cnts[grp] = cnt;
  __syncthreads();
  if (grp == 0) {
    const f32x4 s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const float denom = fmaxf((cnts[0] + cnts[1]) + (cnts[2] + cnts[3]), 1e-6f);
    *reinterpret_cast<f32x4*>(out + (size_t)b * H + col) = s / denom;
  }
}
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* x, int H) {
  __shared__ float sh[4];
  float* row = x + (size_t)blockIdx.x * H;
  float sq = 0.0f;
  for (int c = threadIdx.x; c < H; c += 256) sq += row[c] * row[c];
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float nrm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])) + 1e-9f;
  for (int c = threadIdx.x; c < H; c += 256) row[c] = row[c] / nrm;
}

// frames (N,3,S,S) fp32 -> patches (N*G*G, 3*P*P) bf16, element order (c, ky, kx).
// One workgroup per (frame, patch row py): its threads walk the 3 x P image rows of that band in order, 8 pixels (32 B in, 16 B
// out) per thread and step, so the reads are whole contiguous image rows and the index arithmetic is 32-bit with compile-time
// divisors (PT = patch size, GT = patches per row; 0 = run-time values).  The first version spent its time in 64-bit div / mod
// per four pixels: 45 us for 115 MB (2.6 TB/s).
template <int PT, int GT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* frames, __bf16* patches, int S_, int P_) {
  const int P = PT ? PT : P_, G = GT ? GT : S_ / P_, S = P * G;
  const int n = blockIdx.x / G, py = blockIdx.x - n * G;
  const int K = 3 * P * P, CPR = P / 8;              // 8-pixel chunks per patch row
  const int chunks = 3 * P * G * CPR;                // of this band
  const float* src = frames + ((size_t)n * 3 * S + (size_t)py * P) * S;       // + (c * S + ky) * S + px * P + kx
  __bf16* dst = patches + ((size_t)n * G * G + (size_t)py * G) * K;           // + px * K + c * P * P + ky * P + kx
  for (int q = threadIdx.x; q < chunks; q += 256) {
    const int x8 = q % (G * CPR), r = q / (G * CPR);      // chunk inside the image row; image row of the band (c, ky)
    const int ky = r % P, c = r / P;
    const int px = x8 / CPR, kx = (x8 - px * CPR) * 8;
    const float* sp = src + ((size_t)c * S + ky) * S + x8 * 8;
    const f32x4 v0 = ld4(sp), v1 = ld4(sp + 4);
    bf16x8 o = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3], (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
    *reinterpret_cast<bf16x8*>(dst + (size_t)px * K + c * P * P + ky * P + kx) = o;
  }
}

// x[n][0] = cls + pos[0]; x[n][1+p] = patch_emb[n*P+p] + pos[1+p]; then pre-LayerNorm
template <int NI>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* pe, const float* cls, const float* pos,
                                                           const float* gamma, const float* beta, float* of, __bf16* ob,
                                                           float* stats, int M, int P, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int n = row / (P + 1), t = row % (P + 1);
  f32x4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = 4 * lane + 256 * i;
    const f32x4 base = (t == 0) ? ld4(cls + col) : ld4(pe + ((size_t)n * P + (t - 1)) * H + col);
    v[i] = base + ld4(pos + (size_t)t * H + col);
  }
  if (gamma) ln_row<NI>(v, H, eps, gamma, beta, lane);      // (gamma == NULL: the raw sums)
  store_row<NI>(v, ob, of, row, H, lane);
  if (stats) row_stats<NI>(v, stats, row, lane);
}

// per-frame L2 normalise, mean over frames, L2 normalise again (F > 1); one block per sample
__global__ __launch_bounds__(256) void l2norm_frames_kernel(const float* e, float* out, int F, int D) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  float acc[4] = {0, 0, 0, 0};
  for (int f = 0; f < F; ++f) {
    const float* row = e + ((size_t)b * F + f) * D;
    float sq = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) sq += row[c] * row[c];
    sq = wave_sum(sq);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float nrm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])) + 1e-9f;
    int n = 0;
    for (int c = threadIdx.x; c < D; c += 256, ++n) acc[n] += row[c] / nrm;
  }
  if (F == 1) {
    int n = 0;
    for (int c = threadIdx.x; c < D; c += 256, ++n) out[(size_t)b * D + c] = acc[n];
    return;
  }
  float sq = 0.0f;
  int n = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++n) {
    acc[n] /= (float)F;
    sq += acc[n] * acc[n];
  }
  sq = wave_sum(sq);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float nrm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])) + 1e-9f;
  n = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++n) out[(size_t)b * D + c] = acc[n] / nrm;
}

// encode_fields (text_blocks.py:108-128): mean of the VALID part vectors of a record, then
// v / (||v|| + 1e-9); a record without parts yields the zero vector.  One block per record.
__global__ __launch_bounds__(256) void field_mean_l2_kernel(const float* parts, const int32_t* valid, float* out, int Mx,
                                                            int D) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  float acc[4] = {0, 0, 0, 0};
  int cnt = 0;
  for (int m = 0; m < Mx; ++m) {
    if (valid[(size_t)n * Mx + m] == 0) continue;
    ++cnt;
    const float* row = parts + ((size_t)n * Mx + m) * D;
    int k = 0;
    for (int c = threadIdx.x; c < D; c += 256, ++k) acc[k] += row[c];
  }
  float sq = 0.0f;
  int k = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++k) {
    acc[k] = cnt ? acc[k] / (float)cnt : 0.0f;
    sq += acc[k] * acc[k];
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float nrm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])) + 1e-9f;
  k = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++k) out[(size_t)n * D + c] = cnt ? acc[k] / nrm : 0.0f;
}

#define NI_LAUNCH(H, KERNEL, GRID, STREAM, ...)                                                       \
  do {                                                                                                \
    if ((H) == 256) hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);         \
    else if ((H) == 512) hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);    \
    else if ((H) == 768) hipLaunchKernelGGL((KERNEL<3>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);                    \
  } while (0)

inline bool h_ok(int H) { return H == 256 || H == 512 || H == 768 || H == 1024; }

}  // namespace

extern "C" int ufnd_layernorm(const float* x, int ldx, const float* gamma, const float* beta, void* out_bf16,
                              float* out_f32, int M, int H, float eps, void* stream_) {
  UFND_REQUIRE(x && gamma && beta && (out_bf16 || out_f32) && M >= 1, "layernorm: null argument");
  UFND_REQUIRE(h_ok(H), "layernorm: H=%d (supported 256/512/768/1024)", H);
  UFND_REQUIRE(ldx % 4 == 0 && ldx >= H && ufnd_aligned(x, 16) && ufnd_aligned(gamma, 16) && ufnd_aligned(beta, 16),
               "layernorm: alignment");
  NI_LAUNCH(H, layernorm_kernel, dim3(ufnd_cdiv(M, 4)), (hipStream_t)stream_, x, ldx, gamma, beta, (__bf16*)out_bf16,
            out_f32, M, H, eps);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0,
                               const float* gamma, const float* beta, void* x_bf16, float* x_f32, int B, int L, int H,
                               int vocab, float eps, void* stream_) {
  UFND_REQUIRE(ids && word && pos && type0 && ((gamma && beta) || (!gamma && !beta)) && (x_bf16 || x_f32), "bert_embed: null argument");
  UFND_REQUIRE(h_ok(H) && B >= 1 && L >= 1 && vocab >= 1, "bert_embed: B=%d L=%d H=%d vocab=%d", B, L, H, vocab);
  UFND_REQUIRE(ufnd_aligned(word, 16) && ufnd_aligned(pos, 16) && ufnd_aligned(type0, 16), "bert_embed: alignment");
  const int M = B * L;
  NI_LAUNCH(H, bert_embed_kernel, dim3(ufnd_cdiv(M, 4)), (hipStream_t)stream_, ids, word, pos, type0, gamma, beta,
            (__bf16*)x_bf16, x_f32, M, L, H, vocab, eps, (const int32_t*)nullptr);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_bert_embed_packed(const int64_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                                      const float* type0, const float* gamma, const float* beta, void* x_bf16, float* x_f32,
                                      int T, int max_pos, int H, int vocab, float eps, void* stream_) {
  UFND_REQUIRE(ids && pos_ids && word && pos && type0 && gamma && beta && (x_bf16 || x_f32), "bert_embed_packed: null argument");
  UFND_REQUIRE(h_ok(H) && T >= 1 && max_pos >= 1 && vocab >= 1, "bert_embed_packed: T=%d H=%d vocab=%d", T, H, vocab);
  UFND_REQUIRE(ufnd_aligned(word, 16) && ufnd_aligned(pos, 16) && ufnd_aligned(type0, 16), "bert_embed_packed: alignment");
  NI_LAUNCH(H, bert_embed_kernel, dim3(ufnd_cdiv(T, 4)), (hipStream_t)stream_, ids, word, pos, type0, gamma, beta,
            (__bf16*)x_bf16, x_f32, T, max_pos, H, vocab, eps, pos_ids);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_masked_meanpool_l2(const float* hidden, const int32_t* mask, float* out, int B, int L, int H,
                                       void* stream_) {
  UFND_REQUIRE(hidden && mask && out && B >= 1 && L >= 1, "meanpool: null argument");
  UFND_REQUIRE(H >= 256 && H % 256 == 0 && ufnd_aligned(hidden, 16) && ufnd_aligned(out, 16), "meanpool: H=%d (multiple of 256)", H);
  hipLaunchKernelGGL(meanpool_kernel, dim3(H / 256, B), dim3(256), 0, (hipStream_t)stream_, hidden, mask, out, L, H);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, out, H);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_meanpool_l2_packed(const float* hidden, const int32_t* cu_seqlens, const int32_t* pos_ids, float* out, int B, int H,
                                       void* stream_) {
  UFND_REQUIRE(hidden && cu_seqlens && pos_ids && out && B >= 1, "meanpool_packed: null argument");
  UFND_REQUIRE(H >= 256 && H % 256 == 0 && ufnd_aligned(hidden, 16) && ufnd_aligned(out, 16), "meanpool_packed: H=%d (multiple of 256)", H);
  hipLaunchKernelGGL(meanpool_packed_kernel, dim3(H / 256, B), dim3(256), 0, (hipStream_t)stream_, hidden, cu_seqlens, pos_ids, out, H);
  UFND_CHECK_LAUNCH();
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, out, H);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_vit_patchify(const float* frames, void* patches, int N, int image, int patch, void* stream_) {
  UFND_REQUIRE(frames && patches && N >= 1, "patchify: null argument");
  UFND_REQUIRE(patch % 8 == 0 && image % patch == 0 && ufnd_aligned(frames, 16) && ufnd_aligned(patches, 16),
               "patchify: image=%d patch=%d (patch a multiple of 8, 16-B aligned buffers)", image, patch);
  const int G = image / patch;
  if (patch == 32 && G == 7) hipLaunchKernelGGL((patchify_kernel<32, 7>), dim3(N * G), dim3(256), 0, (hipStream_t)stream_, frames, (__bf16*)patches, image, patch);
  else hipLaunchKernelGGL((patchify_kernel<0, 0>), dim3(N * G), dim3(256), 0, (hipStream_t)stream_, frames, (__bf16*)patches, image, patch);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_vit_assemble(const float* patch_emb, const float* cls, const float* pos, const float* gamma,
                                 const float* beta, float* x_f32, void* x_bf16, float* stats, int N, int P, int H, float eps,
                                 void* stream_) {
  UFND_REQUIRE(patch_emb && cls && pos && ((gamma && beta) || (!gamma && !beta)) && (x_f32 || x_bf16) && N >= 1 && P >= 1, "vit_assemble: null argument");
  UFND_REQUIRE(h_ok(H), "vit_assemble: H=%d", H);
  const int M = N * (P + 1);
  UFND_REQUIRE(!stats || ufnd_aligned(stats, 16), "vit_assemble: stats alignment");
  NI_LAUNCH(H, vit_assemble_kernel, dim3(ufnd_cdiv(M, 4)), (hipStream_t)stream_, patch_emb, cls, pos, gamma, beta, x_f32,
            (__bf16*)x_bf16, stats, M, P, H, eps);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_l2norm_frames(const float* e, float* out, int B, int F, int D, void* stream_) {
  UFND_REQUIRE(e && out && B >= 1 && F >= 1 && D >= 1 && D <= 1024, "l2norm_frames: B=%d F=%d D=%d", B, F, D);
  hipLaunchKernelGGL(l2norm_frames_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream_, e, out, F, D);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_field_mean_l2(const float* parts, const int32_t* valid, float* out, int N, int Mx, int D, void* stream_) {
  UFND_REQUIRE(parts && valid && out && N >= 1 && Mx >= 1 && D >= 1 && D <= 1024, "field_mean_l2: N=%d M=%d D=%d", N, Mx, D);
  hipLaunchKernelGGL(field_mean_l2_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream_, parts, valid, out, Mx, D);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

// ------------------------------------------------------------------------------------------------
// Fold guard: the largest |mean| / std among the rows of a LayerNorm-statistics buffer (see ufnd_gemm_bf16_ln).
// ------------------------------------------------------------------------------------------------
namespace {
// Four lanes per row: lane j of a quad adds 16-B chunks j, j + 4, ... of the row's `parts` {sum, sumsq} pairs (parts is even),
// the quad combines by DPP.  Rows are `nbuf` buffers of M rows each, `buf_stride` floats apart (one launch looks at every
// statistics buffer of an encoder pass).
__global__ __launch_bounds__(256) void ln_fold_guard_kernel(const float* stats, int M, int parts, int nbuf, size_t buf_stride, float inv_h,
                                                            float eps, float* guard) {
  __shared__ float sh[4];
  float worst = 0.0f;
  const int nq = parts >> 1, sub = threadIdx.x & 3;
  const long long total = (long long)M * nbuf;
  // two rows per quad and pass, their (at most 3 + 3: parts <= 24) chunk loads issued together (clamped, masked at use): the
  // kernel is a stream of 192-B rows and was latency-bound with one load in flight per lane (36 us for 55 MB)
  const long long step = (long long)gridDim.x * 64;
  for (long long r = (long long)blockIdx.x * 64 + (threadIdx.x >> 2); r < total; r += 2 * step) {
    f32x4 v[2][3];
    bool live[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long long rk = r + k * step;
      live[k] = rk < total;
      const long long rc = live[k] ? rk : r;
      const int bi = (int)(rc / M), row = (int)(rc - (long long)bi * M);
      const f32x4* p = reinterpret_cast<const f32x4*>(stats + bi * buf_stride + (size_t)row * parts * 2);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int c = sub + 4 * u;
        v[k][u] = p[c < nq ? c : sub];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float sm = 0.0f, sq = 0.0f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (sub + 4 * u < nq) {
          sm += v[k][u][0] + v[k][u][2];
          sq += v[k][u][1] + v[k][u][3];
        }
      }
      sm += quad_xor1(sm); sq += quad_xor1(sq);
      sm += quad_xor2(sm); sq += quad_xor2(sq);
      const float mean = sm * inv_h;
      const float var = fmaxf(sq * inv_h - mean * mean, 0.0f);
      if (live[k]) {
        float ratio = fabsf(mean) * rsqrtf(var + eps);
        ratio = ratio == ratio ? ratio : INFINITY;      // (a NaN statistic must trip the guard: fmaxf would drop it)
        worst = fmaxf(worst, ratio);
      }
    }
  }
  worst = wave_max(worst);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<int*>(guard), __float_as_int(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]))));   // non-negative floats order like ints
}
}  // namespace

extern "C" int ufnd_ln_fold_guard_multi(const float* stats, int M, int parts, int nbuf, size_t buf_stride, int width, float eps, float* guard,
                                        void* stream_) {
  UFND_REQUIRE(stats && guard && M >= 1 && parts >= 2 && parts <= 24 && parts % 2 == 0 && nbuf >= 1 && width >= 1 && ufnd_aligned(stats, 16) &&
               (nbuf == 1 || (buf_stride % 4 == 0 && buf_stride >= (size_t)M * parts * 2)),
               "ln_fold_guard: M=%d parts=%d (even) nbuf=%d stride=%zu", M, parts, nbuf, buf_stride);
  const long long rows = (long long)M * nbuf;
  const long long want = (rows + 63) / 64;
  const int blocks = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(ln_fold_guard_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, stats, M, parts, nbuf, buf_stride,
                     1.0f / (float)width, eps, guard);
  UFND_CHECK_LAUNCH();
  return UFND_OK;
}

extern "C" int ufnd_ln_fold_guard(const float* stats, int M, int parts, int width, float eps, float* guard, void* stream_) {
  return ufnd_ln_fold_guard_multi(stats, M, parts, 1, 0, width, eps, guard, stream_);
}
