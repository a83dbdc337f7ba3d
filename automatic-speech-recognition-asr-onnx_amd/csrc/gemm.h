// GEMM entry points: C[M][N] = A[M][K] * W[N][K]^T with a fused epilogue.
// Both operands are K-contiguous (activations row-major, weights in torch Linear layout),
// so the same LDS image / fragment read serves A and W.
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3 };

struct GemmArgs {
  const void* A = nullptr; int lda = 0;     // [rows][K]; rows readable up to the 128-row tile edge
  const void* W = nullptr; int ldw = 0;     // [N][K]
  int M = 0, N = 0, K = 0;                  // M = valid rows (stores are masked to m < M); N % 128 == 0; K % 64 == 0
  const float* bias = nullptr;              // [N] (nullable)
  const float* add_t = nullptr; int ld_add_t = 0;   // + add_t[n * ld + m]  (transposed f32 matrix, nullable)
  const float* add = nullptr; int ld_add = 0;       // + add[m * ld + n]    (row-major f32, nullable)
  int act = ACT_NONE;
  float* out_f32 = nullptr; int ld_out_f32 = 0;     // row-major f32 out (nullable)
  void* out_lo = nullptr; int ld_out_lo = 0;        // row-major out in the operand dtype, for n < n_split (nullable)
  void* out_t = nullptr; int ld_out_t = 0;          // transposed out in the operand dtype for n >= n_split: out_t[(n-n_split)*ld + m]
  int n_split = 1 << 30;
  // fused row arg-max over n < n_valid (CTC / LM head): partial (max, idx) per 64-column slab
  float* amax_val = nullptr; int32_t* amax_idx = nullptr; int n_valid = 0;
};

// operand dtype selects the kernel: bf16 MFMA (performance mode) or exact-f32 MFMA (verification mode)
void launch_gemm_bf16(const GemmArgs& g, hipStream_t s);
void launch_gemm_f32(const GemmArgs& g, hipStream_t s);

// reduce the per-slab arg-max partials written by the GEMM epilogue: ids[m] = first index of the row max
void launch_argmax_reduce(const float* val, const int32_t* idx, int M, int n_slabs, int32_t* ids, hipStream_t s);
