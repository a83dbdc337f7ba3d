/* Probe / test hooks of the MI355X ASR engine -- libasr_mi355x_probe.so.
 *
 * NOT part of the product C ABI (include/asr_mi355x.h): nothing here is what the reference's onnxruntime binding for the
 * hot path would call. These entries exist for tests/ (kernel-selection parity at the benchmarked sizes) and tools/
 * (tuning probes); the product library libasr_mi355x.so does not export them. */
#ifndef ASR_MI355X_PROBE_H
#define ASR_MI355X_PROBE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One bf16 GEMM C[M][N] = A[M][K] W[N][K]^T through the product's dispatcher with the epilogues of the batch-64 SANM path.
 * Operands are rounded to bf16 on upload. Outputs are host arrays; `kernel` receives the kernel family that ran. */
typedef struct asr_probe_gemm_desc {
  int32_t M, N, K;
  const float* a;          /* [M][K] */
  const float* w;          /* [N][K] */
  const float* bias;       /* [N] or NULL */
  const float* add;        /* [M][N] f32 additive term or NULL */
  int32_t act;             /* 0 none, 1 relu, 2 gelu(erf), 3 gelu(tanh) */
  int32_t ln;              /* 1: C = LayerNorm_noaffine(bf16(A)) W^T + bias evaluated inside the GEMM (row statistics + column sums) */
  float ln_eps;
  int32_t argmax;          /* 1: fused row arg-max over n < n_valid -> out_ids[M] (no matrix output) */
  int32_t n_valid;
  int32_t variant;         /* -1 heuristic; 0..7 pins a kernel variant (csrc/gemm.hip) */
  float* out_lo;           /* [M][N] bf16 results widened to f32, or NULL */
  float* out_f32;          /* [M][N] f32 results, or NULL */
  float* out_stats;        /* [M][N/32][2] (sum, sum of squares) of the bf16 outputs per 32-column group, or NULL */
  int32_t* out_ids;        /* [M] when argmax */
  char kernel[32];         /* out: "t288w", "t288w_amax", "t144w", "t144", "big", "pipe", "pipe_splitk", "skinny" */
} asr_probe_gemm_desc;
int asr_probe_gemm(asr_probe_gemm_desc* d);

/* Decode-shaped GEMM (M <= 64 rows) as a captured chain of dependent launches over `cold_mb` megabytes of weight copies (larger than
 * the Infinity Cache => every launch streams its weights from HBM, like a decoder stack does once per token): microseconds per launch.
 * epilogue: 0 bias -> bf16, 1 bias + GELU -> bf16, 2 bias + residual -> f32, 3 LayerNorm prologue (f32 rows in) -> bf16. */
int asr_probe_gemm_chain(int M, int N, int K, int epilogue, int cold_mb, int replays, float* us_per_launch);
const char* asr_probe_last_kernel(void);      /* kernel family of the last asr_probe_gemm_chain */

/* FP8 mode (ASR_PRECISION_FP8W). Row quantiser: w_bf16[N][K] -> e4m3 bytes out8[N][K], power-of-two scale[N], optional exact bf16
 * dequantisation dq_bf16[N][K]. Decode GEMM on host arrays: out[M][N] f32 = a[M][K] (bf16, M <= 64) x either w_bf16 or (w8, scale);
 * fold != 0 applies the folded LayerNorm (column sums from w_bf16: pass the dequantised copy next to byte weights). */
int asr_probe_quantize_fp8(const uint16_t* w_bf16, int N, int K, uint8_t* out8, float* scale, uint16_t* dq_bf16);
/* MXFP4 mode: block quantiser (out4 [N][K / 2] nibbles, scale8 [N][K / 32] e8m0 bytes, dq the exact bf16 dequantisation) and the decode GEMM over them */
int asr_probe_quantize_mxfp4(const uint16_t* w_bf16, int N, int K, uint8_t* out4, uint8_t* scale8, uint16_t* dq_bf16);
int asr_probe_decode_gemm_mxfp4(int M, int N, int K, const uint16_t* a, const uint8_t* w4, const uint8_t* scale8, const uint16_t* w_dq, const float* bias,
                                int fold, float* out);
int asr_probe_decode_gemm(int M, int N, int K, const uint16_t* a, const uint16_t* w_bf16, const uint8_t* w8, const float* scale,
                          const float* bias, int fold, float* out);

/* FP8 matrix-pipe GEMM (precision mode ASR_PRECISION_FP8MM) on host arrays: out = act((a8 w8^T) a_scale w_scale[n] + bias) as e4m3 bytes (out8), or
 * (a8 w8^T) a_scale w_scale[n] + bias + add as f32 (out_f32). iters > 0 also times the launch (microseconds in *us). */
int asr_probe_gemm_fp8(int M, int N, int K, const uint8_t* a8, const uint8_t* w8, const float* w_scale, float a_scale, const float* bias,
                       const float* add, int act, uint8_t* out8, float* out_f32, int iters, float* us);

/* launches per GEMM kernel family since the last reset, as "family=count;..." (host-side counters: hipGraph replays do not
 * count, so reset, run a session once on a new batch geometry, read). reset != 0 clears the counters after the read. */
int asr_probe_gemm_counts(int reset, char* buf, int cap);

/* Tuning hook: time `iters` launches of the bf16 GEMM on device-resident pseudo-random operands.
 * variant: -1 heuristic, 0..7 kernel variants (csrc/gemm.hip). epilogue: 0 bias->lo, 1 bias+relu->lo,
 * 2 bias+residual->f32, 3 two residual terms->f32, 4 transposed store, 5 LayerNorm-folded FFN-1, 6 producer epilogue. */
int asr_probe_gemm_bench(int variant, int M, int N, int K, int epilogue, int iters, float* avg_ms);
/* microseconds per grid-wide barrier of a cooperative launch with n_workgroups x 512 threads (single counter + __threadfence) */
int asr_probe_grid_barrier(int n_workgroups, int iters, float* us_per_barrier);
/* hierarchical barrier (per-XCD arrival counters, relaxed agent-scope atomics, no fence): mode 1 = one release flag, 2 = one flag
 * per XCD; also checks that an sc1 payload written before a barrier is visible after it. mode + 16 * KiB makes every workgroup
 * also read KiB kibibytes of one shared buffer per round through sc1 loads (+ 8: through plain cached loads). */
int asr_probe_grid_barrier2(int n_workgroups, int iters, int mode, float* us_per_barrier);

#ifdef __cplusplus
}
#endif
#endif /* ASR_MI355X_PROBE_H */
