// fp32 skinny-GEMM family for the Tier-A head (batch rows M = 4..256, weights streamed once).
// All three use v_mfma_f32_32x32x2_f32: exact fp32 (bit-identical to an fmaf chain), 157 TF peak,
// so at M = 32 the kernels are HBM-bound on the weight stream, not MFMA-bound.
#pragma once
#include "common.hpp"

#define UFND_GEMM_MAX_PROB 16

// Y[m][n] = act(sum_k X[m][k] W[n][k] + bias[n])            (nn.Linear forward)
struct NtProb {
  const float* X;     // (M, K) row stride ldx (ldx % 4 == 0, 16-B aligned)
  const float* W;     // (N, K) row stride ldw
  const float* bias;  // (N) or null
  float* Y;           // (M, N) row stride ldy; when ksplit > 1: partials [ksplit][M][N]
  float* Z;           // optional pre-activation copy (M, N) row stride ldz
  int M, N, K, ldx, ldw, ldy, ldz;
  int act;            // 0 none, 1 exact GELU
  float drop_p;       // dropout on the output (train), 0 = off
  uint32_t drop_layer;
  int ksplit;         // grid-level split of K (1 => epilogue in-kernel)
};
int launch_nt(const NtProb* probs, int nprob, const ufnd_step_state* st, hipStream_t stream);

// dX[m][k] = (sum_n dY[m][n] W[n][k]) * gelu'(actZ[m][k]) * dropmask + add[m][k]
struct NnProb {
  const float* dY;    // (M, N) row stride lddy (lddy % 4 == 0)
  const float* W;     // (N, K) row stride ldw
  float* out;         // (M, K) row stride ldo; when nsplit > 1: partials [nsplit][M] rows of stride ldo
  const float* actZ;  // optional (M, K) row stride ldz
  const float* add;   // optional (M, K) row stride ldadd
  int M, N, K, lddy, ldw, ldo, ldz, ldadd;
  float drop_p;
  uint32_t drop_layer;
  int drop_ld;        // logical row stride used for the dropout element index
  int nsplit;
};
int launch_nn(const NnProb* probs, int nprob, const ufnd_step_state* st, hipStream_t stream);

// dW[n][k] = sum_m dY[m][n] X[m][k];  db[n] = sum_m dY[m][n]
struct TnProb {
  const float* dY;    // (M, N) row stride lddy
  const float* X;     // (M, K) row stride ldx
  float* dW;          // (N, K) row stride ldw
  float* db;          // (N) or null
  int M, N, K, lddy, ldx, ldw;
  // Batch rows in SEGMENTS (gathered data-parallel factors: rank r's rows live at dY + r * seg_dy, X + r * seg_x, seg_rows rows
  // each, M = ranks x seg_rows); seg_rows == 0: one contiguous panel.
  int seg_rows, seg_dy, seg_x;
};
int launch_tn(const TnProb* probs, int nprob, hipStream_t stream);
