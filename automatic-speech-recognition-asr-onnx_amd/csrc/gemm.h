// GEMM entry points: C[M][N] = A[M][K] * W[N][K]^T with a fused epilogue.
// Both operands are K-contiguous (activations row-major, weights in torch Linear layout),
// so the same LDS image / fragment read serves A and W.
#pragma once
#include "common.h"

// ACT_SWIGLU: W rows interleave gate / up (row 2j = gate_j, row 2j + 1 = up_j); the epilogue stores silu(gate_j) * up_j to out_lo column j
// (out_lo is [M][N / 2]); no other output term may be set.
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3, ACT_SWIGLU = 4 };

struct GemmArgs {
  const void* A = nullptr; int lda = 0;     // [rows][K]; rows readable up to the 128-row tile edge
  const void* W = nullptr; int ldw = 0;     // [N][K]
  int M = 0, N = 0, K = 0;                  // M = valid rows (stores are masked to m < M); N % 128 == 0; K % 64 == 0
  const float* bias = nullptr;              // [N] (nullable)
  const float* add = nullptr; int ld_add = 0;       // + add[m * ld + n]   (row-major f32, nullable; readable to the tile edge)
  // second row-major f32 term, added AFTER the activation: + add2[row(m) * ld + n], row(m) = add2_rows ? add2_rows[m] : m
  const float* add2 = nullptr; int ld_add2 = 0; const int32_t* add2_rows = nullptr;
  int act = ACT_NONE;
  float* out_f32 = nullptr; int ld_out_f32 = 0;     // row-major f32 out (nullable)
  void* out_lo = nullptr; int ld_out_lo = 0;        // row-major out in the operand dtype (nullable)
  // lo_group > 0: grouped layout instead -- out_lo[(n / lo_group) * ld_out_lo + m * lo_group + n % lo_group]
  // (per-head K/V slabs [group][row][lo_group] for the decoder's cross-attention; ld_out_lo = slab stride)
  int lo_group = 0;
  // transposed out in the operand dtype: out_t[n * ld + m]. Exclusive with every row-major term above
  // (the kernel then runs in the un-swapped MFMA orientation: 4 consecutive m per lane).
  void* out_t = nullptr; int ld_out_t = 0;
  // fused row arg-max over n < n_valid (CTC / LM head): partial (max, idx) per 64-column slab
  float* amax_val = nullptr; int32_t* amax_idx = nullptr; int n_valid = 0;
  // Skinny path (decode), M <= 32: when ln_x is set, A is produced on the fly as LayerNorm(ln_x) -- every workgroup
  // normalises the few activation rows itself (two-pass statistics in f32), so no separate LayerNorm launch and no bf16
  // round trip of the normalised rows through HBM. gamma/beta nullable (affine folded into W, Export_Whisper.py:215-225).
  const float* ln_x = nullptr; int ld_ln_x = 0; const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 1e-5f;
  // Skinny path: RMSNorm of the A rows folded into the product -- RMSNorm(x) W^T = rstd(x) (x W^T) -- A holds the RAW rows in bf16,
  // every workgroup accumulates sum(x^2) from the A fragments it streams anyway and scales its output rows by rsqrt(mean + a_rms_eps)
  // before bias / residual terms. a_rms_eps > 0 enables it (the norm weight must be folded into W).
  float a_rms_eps = 0.0f;
  // Skinny path, byte weights (precision mode ASR_PRECISION_FP8W of Qwen3-ASR): W8 holds OCP e4m3 bytes [N][K] (row pitch ldw8 bytes) with one power-of-two f32 scale
  // per output column; a lane loads 8 bytes per fragment, widens them to bf16 in registers (exact) and the finished sum is multiplied by the scale -- bit for bit the
  // product over the dequantised bf16 copy (which W keeps for every other path). With W8 set a launch of <= 64 rows always streams weights (skinny kernel).
  const unsigned char* W8 = nullptr; int ldw8 = 0; const float* w_scale = nullptr;
  // Skinny path, MXFP4 weights (precision mode ASR_PRECISION_MXFP4W of Qwen3-ASR): e2m1 nibbles [N][K / 2] + one e8m0 scale byte per (row, 32 k) [N][K / 32]
  // (launch_quantize_rows_mxfp4); the K-step of the MFMA is the block of the format, the widening instruction applies the scale. W keeps the dequantised bf16 copy.
  const unsigned char* W4 = nullptr; const unsigned char* w_scale4 = nullptr;
  // LayerNorm evaluated inside the GEMM (144-row-tile kernel only; K must span the whole normalised row): A holds the RAW rows
  // x in bf16, row statistics over the first ln_dim columns are accumulated from the LDS tiles during the MFMA loop and
  // C = rstd (x W^T - mean ln_colsum) + bias. Needs the LayerNorm affine folded into W / bias; ln_colsum[n] = sum_k W[n][k].
  const float* ln_colsum = nullptr; int ln_dim = 0;
  // row statistics hand-over between GEMMs: a producer with st_out set writes, per row and per 32-column group, the
  // (sum, sum of squares) of the bf16-ROUNDED values it stores to out_lo -- st_out[m * (N / 32) + n / 32]; a LayerNorm-fused
  // consumer given ln_stats_in (ln_slots groups per row) sums them in a fixed order instead of re-deriving the statistics
  // from its LDS tiles in every column tile.
  float2* st_out = nullptr;
  const float2* ln_stats_in = nullptr; int ln_slots = 0;
  // skinny path, split-K ACROSS workgroups (more workgroups stream the weights when N / 16 alone cannot fill the chip): partial
  // sums go to sk_ws ([splits][rows16][N] f32), the last workgroup of a column granule (ticket in sk_cnt[N / 16], self-resetting)
  // adds them in split order and runs the epilogue -- deterministic. Both buffers are per session; null => no split.
  float* sk_ws = nullptr; size_t sk_ws_bytes = 0; int32_t* sk_cnt = nullptr; int sk_splits = 0;   // sk_splits: set by the launcher
  // device-side row count (bf16 kernels): rows >= min(M, *m_dev) are neither computed nor stored -- row tiles past it exit at once.
  // Lets a graph-captured launch sized for the worst case follow a data-dependent row count (Paraformer token rows) without a host sync.
  // tiled path, split-K across workgroups (set by the launcher for small grids with long K): grid.y = k_splits, every split writes its f32
  // partial product to sk_ws[split][M][N]; a second launch sums the partials in split order and runs the epilogue -- deterministic
  int k_splits = 0;
  // split-K second pass only, N == 1024 (one output row per 256-thread workgroup of the reduce launch): the finished f32 row is also RMS-normalised
  // (affine-free: rsqrt(mean(x^2) + rms_eps)) and stored as the operand rows of the NEXT GEMM -- the stand-alone RMSNorm launch between a
  // residual-producing projection and the projection that consumes its norm disappears. Ask gemm_reduce_can_norm() first.
  void* rms_out = nullptr; int ld_rms_out = 0; float rms_eps = 1e-6f;
  const int32_t* m_dev = nullptr;
  int group_m = 0;   // (set by the launcher) row tiles walked per column tile before moving on: keeps wide weight matrices L2-resident
  uint32_t* dbg_clk = nullptr;   // ping-pong kernel, timed instance (bench hook): per-phase segment clocks of two waves of workgroup 0
  int dbg = 0;   // tuning ablations (bench hook only): 1 = no refills, 2 = no MFMA, 4 = no epilogue
};

// operand dtype selects the kernel: bf16 MFMA (performance mode) or exact-f32 MFMA (verification mode)
void launch_gemm_bf16(const GemmArgs& g, hipStream_t s);
bool gemm_reduce_can_norm(const GemmArgs& g);     // true when launch_gemm_bf16(g) takes the tiled split-K pass whose reduce launch can also write rms_out
void launch_gemm_f32(const GemmArgs& g, hipStream_t s);
// csrc/gemm_pp.hip: 256 x 256 tiles, two wave groups alternating matrix / memory segments. launch_gemm_pp returns false when the shape or the
// epilogue has no instance (the caller then takes the other tilings).
bool gemm_pp_supported(const GemmArgs& g);
bool launch_gemm_pp(const GemmArgs& g, hipStream_t s, int var = 0);
bool gemm_ln_fusable(const GemmArgs& g);   // true when launch_gemm_bf16 would accept g with ln_colsum set
void gemm_set_variant(int v);   // tuning hook: -1 = built-in heuristic
void gemm_reload_env();         // re-read the ASR_GEMM_* / ASR_SKINNY_* / ASR_DECODE_* switches and the device's CU count (called at session creation)
int gemm_env_decode_nt();       // ASR_DECODE_NT, ASR_DECODE_KS (0 = the cost model), ASR_DECODE_ATTN_WAVE, CUs of the device: as of the last reload
int gemm_env_decode_ks();
bool gemm_env_decode_attn_wave();
bool gemm_env_decode_attn_online();
int gemm_env_decode_rb();
int gemm_env_cus();
const char* gemm_last_kernel(); // kernel family of this thread's last launch_gemm_bf16 ("t288w", "t144", "pipe", "skinny", ...): test hook
bool gemm_skinny144_enabled();
void gemm_kernel_counts_reset();
int gemm_kernel_counts(char* buf, int cap);   // "family=launches;..." since the last reset (host-side: graph replays do not count)

// ---- decode-step GEMM (csrc/decode_gemm.hip): out[M <= 64][N] = A W^T with the grid shaped to one even round of the chip (32-column
// granules / K split across workgroups) and, with `colsum`, the affine-free LayerNorm of the raw bf16 rows in A folded into the product
struct DecGemmArgs {
  const bf16_t* A = nullptr; int lda = 0;            // [M][K] bf16 (colsum set: the RAW residual rows, LayerNorm applied inside)
  const bf16_t* W = nullptr; int ldw = 0;            // [N][K]
  const unsigned char* W8 = nullptr; const float* w_scale = nullptr;    // FP8 mode instead of W: e4m3 bytes [N][K] (pitch ldw) + one power-of-two scale per output column
  const unsigned char* W4 = nullptr; const unsigned char* w_scale4 = nullptr;   // MXFP4 mode instead of W: e2m1 nibbles [N][K / 2] + e8m0 block scales [N][K / 32] (launch_quantize_rows_mxfp4)
  int M = 0, N = 0, K = 0;
  int plan_M = 0;                                    // rows the grid shape (column granule, K splits: the summation order) is planned for; 0 = M. The decode chains of whisper.hip plan every
                                                     // chain for the largest one, so a sequence's result does not depend on the chain it rides in
  const float* bias = nullptr;
  const float* colsum = nullptr; float ln_eps = 1e-5f;      // c[n] = sum_k W[n][k]: out = rstd (A W^T - mean c) + bias
  const float* add = nullptr; int ld_add = 0;        // + f32 residual rows
  int act = ACT_NONE;                                // NONE / RELU / GELU_ERF / GELU_TANH
  float* out_f32 = nullptr; int ld_out_f32 = 0;      // either or both outputs
  bf16_t* out_lo = nullptr; int ld_out_lo = 0;
  float* ws = nullptr; size_t ws_bytes = 0; int32_t* cnt = nullptr;     // split-K partials [splits][rows16][N] + self-resetting tickets [N / 16] (per session)
  unsigned long long* dbg_clk = nullptr;             // tuning (tools/probes/decode_gemm_clock.py): thread 0 of the first and of the last workgroup stamp wall_clock64() at five points ([2][5])
};
bool decode_gemm_supported(const DecGemmArgs& g);
void decode_gemm_plan(const DecGemmArgs& g, int* nt, int* splits, int* row_blocks = nullptr);
void launch_decode_gemm(const DecGemmArgs& g, hipStream_t s);
// touch the weight bytes of a coming launch_decode_gemm(g) from the workgroups (hence XCDs) that will stream them; for a side branch of the decode graph
void launch_decode_gemm_prefetch(const DecGemmArgs& g, hipStream_t s);
void launch_colsum_bf16(const bf16_t* W, int ldw, int N, int K, float* c, hipStream_t s);

// ---- FP8 matrix-pipe GEMM (csrc/gemm_fp8.hip, precision mode ASR_PRECISION_FP8MM): e4m3 operands [rows][K bytes], power-of-two scales applied in the epilogue
struct Fp8GemmArgs {
  const unsigned char* A = nullptr; int lda = 0;      // [M][K] e4m3 bytes (pitch in bytes); value = byte * a_scale
  const unsigned char* W = nullptr; int ldw = 0;      // [N][K] e4m3 bytes; value = byte * w_scale[n]
  int M = 0, N = 0, K = 0;                            // N % 256 == 0, K % 256 == 0
  const float* w_scale = nullptr; float a_scale = 1.0f;
  const float* bias = nullptr;                        // [N]
  const float* add = nullptr; int ld_add = 0;         // f32 residual rows (f32 output only)
  int act = ACT_NONE;                                 // byte output only
  unsigned char* out8 = nullptr; int ld_out8 = 0; float out_inv_scale = 1.0f;    // e4m3 bytes of act(...) * out_inv_scale, saturating at +-448
  float* out_f32 = nullptr; int ld_out_f32 = 0;
  unsigned long long* sat_count = nullptr;            // byte output: += the number of elements whose scaled value left +-448 (or was NaN) before the clamp
  int group_m = 0;                                    // (set by the launcher)
};
void launch_gemm_fp8(const Fp8GemmArgs& g, hipStream_t s);

// reduce the per-slab arg-max partials written by the GEMM epilogue: ids[m] = first index of the row max
void launch_argmax_reduce(const float* val, const int32_t* idx, int M, int n_slabs, int32_t* ids, hipStream_t s);
