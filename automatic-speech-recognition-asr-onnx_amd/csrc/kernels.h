// Non-GEMM kernels of the ASR hot path (gfx950).
#pragma once
#include "common.h"

// Per-utterance geometry, built on the host for every run and uploaded with one copy.
struct UttPlan {
  int64_t audio_off;   // first sample in the packed audio buffer
  int32_t n_samples;
  int32_t n_frames;    // fbank / STFT frames
  int32_t frame_off;   // first row in the packed mel buffer
  int32_t n_lfr;       // low-frame-rate rows (SenseVoice/Paraformer) or encoder positions (Whisper)
  int32_t T;           // encoder sequence length (n_lfr + prompts)
  int32_t row_off;     // first row in the packed activation matrices (multiple of 16)
  int32_t lang;        // language selector index
  int32_t blk0;        // first front-end workgroup of this utterance
};

// ---- Kaldi fbank: frames -> |DFT|^2 -> mel -> ln  (SenseVoice/Export_SenseVoice.py:139-160,275-278)
// dft_packed / mel_packed are MFMA-fragment-ordered constant tables built by arena.py.
struct FbankArgs {
  const float* audio;          // packed samples
  const UttPlan* plan;
  const int32_t* blk_utt;      // per workgroup: utterance
  const int32_t* blk_f0;       // per workgroup: first frame (multiple of 64)
  const float* dft_packed;     // [n_bin_tiles][2][n_kchunks][64 lanes][4]
  const float* mel_packed;     // [n_mel_tiles][n_bin_tiles][64 lanes][4]
  float* mel_out;              // [total_frames][n_mels]
  int n_bin_tiles;             // ceil((nfft/2+1)/16)
  int n_kchunks;               // win_length / 16
  int n_mel_tiles;             // n_mels / 16
  int n_mels;
  int win, hop;
  float log_floor;             // FLT_EPSILON (Kaldi) / 1e-10 (Whisper)
  // Whisper variant (Whisper/STFT_Process.py:224-246): reflect pad nfft/2 left, nfft/2 - hop right (last frame dropped),
  // log10 instead of ln, and the per-workgroup maximum written to blk_max for the per-utterance clamp
  int whisper;
  int dbg = 0;                 // ASR_FBANK_DBG (fbank_split_kernel, timing-only ablations; results are garbage by design): 1 no DFT steps, 2 no mel projection, 4 no audio loads, 8 no basis refills
  float* blk_max;
  // bf16 sessions: the DFT on the bf16 matrix pipe with split operands (launch_fbank_split_table): null = exact-f32 MFMA
  const void* dft_split = nullptr;
};
void launch_fbank(const FbankArgs& a, int n_blocks, hipStream_t s);
// three-term bf16 split (hi + mid + lo = the f32 value to 2^-24) of the packed DFT basis, as 16x16x32 fragments:
// out[(bin tile * 2 + re/im) * ceil(win / 32) + k chunk][term][lane] x 16 bytes; rows k >= win are zero
size_t fbank_split_table_bytes(int n_bin_tiles, int win);
void launch_fbank_split_table(const float* dft_packed, int n_bin_tiles, int n_kchunks16, void* out, hipStream_t s);

// ---- Whisper log-mel finish (Export_Whisper.py:425-427): max(x, utterance_max - 8), (x + 4) / 4, written in the
// GAPPED time-major layout the conv stem reads: utterance b owns rows [2*row_off_b, 2*row_off_b + 2*rows_b); row
// 2*row_off_b is the conv's left zero pad, frame f sits at row 2*row_off_b + 1 + f, everything else is zero.
template <typename T>
void launch_whisper_mel_finish(const float* mel, const float* blk_max, const UttPlan* plan, const int32_t* grow_utt,
                               int n_gapped_rows, int n_mels, T* out, hipStream_t s);
// zero the rows of a gapped-layout matrix that are not frames (conv zero padding after conv1)
// dst row m (row space of `to`, utterance row_utt[m]) = src row of the same (utterance, position) in the row space of `from`; pad rows zero
void launch_compact_rows(const float* src, const UttPlan* from, const UttPlan* to, const int32_t* row_utt, int rows, int d, float* dst, hipStream_t s);
template <typename T>
void launch_zero_gap_rows(T* buf, int ld, int n_cols, const UttPlan* plan, const int32_t* grow_utt, int n_gapped_rows,
                          hipStream_t s);

// affine-free LayerNorm of f32 rows written as OCP e4m3 bytes of y * inv_scale (saturating): operand rows of the FP8 matrix-pipe GEMM (csrc/gemm_fp8.hip)
void launch_layernorm_fp8(const float* x, int ld_x, int rows, int D, float eps, float inv_scale, unsigned char* out, int ld_out, hipStream_t s);

// ---- decoder token + position embedding: x[b*n + i] = embed[ids[b*n + i]] + pos[hist + i]   (Export_Whisper.py:450-497)
template <typename T>
void launch_embed_pos(const int32_t* ids, int rows, int n, int hist, const int32_t* hist_dev, const T* embed, const float* pos, int d,
                      float* x, hipStream_t s);
void launch_add_scalar(int32_t* p, int v, hipStream_t s);      // *p += v (device-side history counter)

// ---- decoder attention, head_dim 64, n <= 8 new queries per sequence. Keys/values are rows of 64:
//   self : cache [b][h][S_max][64]; the n new rows are appended from `kv_new` (fused qkv GEMM output) and attended
//          with the reference's additive -128 causal mask (Export_Whisper.py:468-480,640-647)
//   cross: per-(layer, head) slabs [row][64] written by the encoder's cross-KV GEMM; no mask (:654-660)
struct DecAttnArgs {
  const void* q; int ld_q; int q_col0;
  const void* kv_new; int ld_new; int k_col0, v_col0;          // null for cross
  void* k_base; void* v_base;
  int64_t stride_b, stride_h;                                  // element strides of (b, h) inside k_base / v_base
  const UttPlan* plan;                                         // cross: row_off / n_lfr per sequence; null for self
  int hist, n, n_heads, causal;
  int max_keys = 0;                                            // upper bound of keys per sequence (sizes the LDS score buffer)
  int sc_ld = 0;                                               // (set by the launcher)
  const int32_t* hist_dev;                                     // when set, the history length is read from device memory (graph replay)
  void* out; int ld_out;
  const float* k_scale = nullptr; const float* v_scale = nullptr;   // FP8 cross-K/V: k_base / v_base hold e4m3 bytes, scale[head][sequence] per slab
  // paged self-KV cache (block table; stride_b / stride_h unused): position s of (sequence b, head h) lives at
  //   k_base + page_table[b * pages_per_seq + (s >> 4)] * page_stride + h * 1024 + (s & 15) * 64       (pages of 16 positions x 64 dims per head)
  const int32_t* page_table = nullptr; int pages_per_seq = 0; int64_t page_stride = 0;
  // a launch over the sub-batch [b0, b0 + grid.x) of the session's batch (the decode chains of whisper.hip): every per-sequence index is b0 + blockIdx.x;
  // scale_ld = sequences per head row of k_scale / v_scale (0: the launch's own grid.x)
  int b0 = 0, scale_ld = 0;
};
// FP8 (OCP e4m3, power-of-two scales) quantisers of precision mode ASR_PRECISION_FP8W
void launch_quantize_rows_fp8(const bf16_t* W, int ld, int N, int K, unsigned char* W8, float* scale, bf16_t* Wdq, hipStream_t s);
// MXFP4 (e2m1 + one e8m0 scale per 32 k): W4 [N][K / 2] bytes, S [N][K / 32] bytes, Wdq (nullable) the exact bf16 dequantisation
void launch_quantize_rows_mxfp4(const bf16_t* W, int ld, int N, int K, unsigned char* W4, unsigned char* S, bf16_t* Wdq, hipStream_t s);
void launch_quantize_crosskv_fp8(bf16_t* slabs, size_t slab_elems, int n_slabs, const UttPlan* plan, int batch, unsigned char* out8, float* scale,
                                 int dq_in_place, hipStream_t s);
template <typename T>
void launch_decode_attention(const DecAttnArgs& a, int batch, hipStream_t s);

// ids[r] = first arg-max over n < n_valid of logits[r][n] + (extra ? extra[n] : 0)
void launch_argmax_rows(const float* logits, int ld, int rows, int n_valid, const float* extra, int32_t* ids, hipStream_t s);
// NO_SPEECH_DETECTION (Export_Whisper.py:334-348): prob[r] = softmax(logits[r] - penalty)[no_speech_id] over the first n_valid columns, where
// `penalty` is the permanent suppress bias the logits carry (-128 on the suppressed ids: subtracting it re-adds the +128 the reference adds back)
void launch_no_speech_prob(const float* logits, int ld, int rows, int n_valid, const float* penalty, int no_speech_id, float* prob, hipStream_t s);

// ---- penalty-greedy head (Export_Whisper.py:312-325 APPLY_PENALTY + :243-251 GREEDY_SEARCH): logits of the last `range` saved
// ids are multiplied by `value` once `range` ids are saved (the host's rule, Inference_Whisper_ONNX.py:630-632); gather first,
// scatter second, so a repeated id is scaled once. In place on the f32 logits. n_saved lives on the device (graph replay).
// partial = 1 (Qwen3-ASR, Export_Qwen_ASR.py:1403-1415): the window is save_id[:, -range:] of whatever exists -- fewer ids are penalised too.
void launch_apply_penalty(float* logits, int ld, int rows, const int32_t* save_ids, int ld_save, const int32_t* n_saved,
                          int range, float value, hipStream_t s, int partial = 0);
// save_ids[r][*n_saved] = next[r] (the counter itself is advanced by launch_add_scalar afterwards)
void launch_append_ids(const int32_t* next, int rows, int32_t* save_ids, int ld_save, const int32_t* n_saved, hipStream_t s);

// ---- TOPK_TOPP_SAMPLING head (Export_Whisper.py:263-308), one workgroup per sequence: repetition penalty on every saved id
// (negative logits multiplied, others divided; gather before scatter), + `extra` bias (BEGIN_SUPPRESS), x 1/temperature, top-k
// (k <= 64, ties to the lower index), soft-max + exclusive-cumsum top-p cut, Gumbel-max with clamped uniforms. The uniforms come
// from `noise` ([rows][top_k], caller-supplied: parity tests) or, when null, from a counter-based generator keyed by
// (seed, step = *n_saved, row, rank). Writes next[r]; the caller appends it to the history.
struct SampleArgs {
  float* logits; int ld; int rows; int n_valid;
  const float* extra;
  const int32_t* save_ids; int ld_save; const int32_t* n_saved;
  float temperature, top_p, repetition_penalty; int top_k;
  const float* noise; uint64_t seed;
  int32_t* next;
};
void launch_sample_topk_topp(const SampleArgs& a, hipStream_t s);

// ---- LFR stacking + CMVN + positions + prompt rows (Export_SenseVoice.py:280-287)
struct LfrArgs {
  const float* mel;            // [frames][n_mels]
  const UttPlan* plan;
  const int32_t* row_utt;      // per packed row: utterance (or -1)
  const float* cmvn_means;     // [feat]
  const float* cmvn_vars;      // [feat]
  const float* speech_pos;     // [max_lfr][feat]
  const float* language_embed; // [n_lang][feat]   (position-folded)
  const float* system_embed;   // [n_prompt-1][feat]
  float* out;                  // [rows][ld_out]
  int ld_out, feat, n_mels, lfr_m, lfr_n, n_prompt, n_rows;
  // affine_mode 1 (Paraformer, Export_Paraformer.py:483): x * cmvn_vars + bias_table[j] (bias_table = speech_pos), no prompts
  int affine_mode = 0;
  bf16_t* out_lo = nullptr;    // optional bf16 copy of `out` (same leading dimension): the operand of the LayerNorm-fused projection
};
void launch_lfr_cmvn(const LfrArgs& a, hipStream_t s);

// ---- LayerNorm over the last dim, one wave per row. gamma/beta may be null (affine folded away).
// Output in operand dtype with zero fill of columns [D, ld_fill).
template <typename OutT>
void launch_layernorm(const float* x, int ld_x, int rows, int D, const float* gamma, const float* beta, float eps,
                      OutT* out, int ld_out, int fill_to, hipStream_t s, const int32_t* rows_dev = nullptr);   // rows_dev: device-side row count


// ---- multi-head self-attention over packed ragged utterances (no mask inside an utterance).
// q, k: [rows][ld] row-major; vt: [H*HD][ld_vt] (time-contiguous); ctx out: [rows][ld_ctx].
// Scale is pre-folded into q and k (d^-1/4 each, Export_SenseVoice.py:210-216).
struct AttnArgs {
  const void* q; const void* k; int ld_qk;   // ld_qk: row stride of k (and of q unless ld_q is set)
  int ld_q = 0;
  const void* vt; int ld_vt;
  void* ctx; int ld_ctx;
  const UttPlan* plan;
  const int32_t* qb_utt;       // per query block: utterance
  const int32_t* qb_q0;        // per query block: first query row inside the utterance
  int n_qblocks, n_heads;
  // bf16 kernel geometry (attention_geometry): a query block = n_waves * qt tiles of 16 rows; max_T picks the chunk size.
  // The f32 kernel uses fixed 64-row blocks (qt = 0).
  int qt = 0, n_waves = 4, max_T = 0;
  // cross-attention: queries follow q_plan (row_off / T of the query rows), keys / values follow plan; null => self-attention
  const UttPlan* q_plan = nullptr;
  // bf16 kernels: causal = 1 masks keys > the query's own index (decoder prefill); kv_group > 1 = grouped-query attention,
  // q head h reads K / V head h / kv_group
  int causal = 0, kv_group = 1;
};
// query-block geometry for the longest utterance of a batch: rows per block = 16 * qt * n_waves
void attention_geometry(int max_T, int head_dim, int* qt, int* n_waves);
void launch_attention_bf16_hd128(const AttnArgs& a, hipStream_t s);
void launch_attention_bf16_hd64(const AttnArgs& a, hipStream_t s);
void launch_attention_f32(const AttnArgs& a, int head_dim, hipStream_t s);

// ---- fused SANM attention half for windows of <= 144 rows (sanm_fused.hip): per (utterance, head) workgroup
// q|k|v projection -> attention -> FSMN, intermediates in LDS. h: [rows][ld_h] operand-dtype LayerNorm output;
// wqkv: [3 d][ldw] (q rows, k rows, v rows); ctx: bf16 [rows][ld_ctx]; mem: f32 [rows][ld_mem] (rows < T16 written).
struct SanmFusedArgs {
  const void* h; int ld_h; int K;
  const void* wqkv; int ldw;
  const float* bqkv;
  const float* wfsmn; const float* bfsmn;
  const UttPlan* plan; int n_utts, n_heads, d;
  void* ctx; int ld_ctx;
  float* mem; int ld_mem;
  int n_rows_alloc;            // rows of h that may be read (rows past an utterance's end are read but never used)
  // LayerNorm evaluated inside the projection: h holds the RAW rows x (bf16), row statistics over the first ln_dim columns
  // are accumulated from the LDS tiles while the MFMA loop runs and q|k|v = rstd (x W^T - mean colsum) + b. Needs the
  // LayerNorm affine folded into wqkv / bqkv; ln_colsum[n] = sum_k wqkv[n][k]. Null => h is already normalised.
  const float* ln_colsum = nullptr; int ln_dim = 0; float ln_eps = 1e-5f;
  const float2* ln_stats_in = nullptr; int ln_slots = 0;   // producer-side row statistics (GemmArgs::st_out layout); null => computed here
  int dbg = 0;
};
bool sanm_fused_supported(int max_T, int d_head, int n_heads, int d, int fsmn_taps, int K);

// ---- one whole SANM block per launch (csrc/sanm_block.hip): clusters of four workgroups per <= 144-row window, d = 512, 4 heads of 128,
// FFN 2048, LayerNorms folded into the projections (bf16 mode). Buffers are the packed row-major activations of the session.
// per-block constants of the 8-wave kernel (a launch walks `n_layers` consecutive entries of a device-side table)
struct SanmBlockLayer {
  const float *bqkv, *cqkv, *wfsmn, *bfsmn, *b1, *c1, *b2;
  const void* wpack;                                               // fragment-major copy of the block's four matrices (launch_sanm_block8_pack)
};
struct SanmBlockArgs {
  const bf16_t* wqkv; const float* bqkv; const float* cqkv;      // [1536][512] (LayerNorm affine folded), bias, column sums
  const float* wfsmn; const float* bfsmn;                          // [512][11], [512] (linear_out.bias rides here)
  const bf16_t* wout;                                              // [512][512]
  const bf16_t* w1; const float* b1; const float* c1;              // [2048][512] (LayerNorm affine folded), bias, column sums
  const bf16_t* w2; const float* b2;                               // [512][2048]
  const SanmBlockLayer* layers = nullptr; int n_layers = 1;        // 8-wave kernel: the launch's blocks (device table), first entry = the first block of the launch
  int flag_stride = 0;                                             // ... words between the exchange counters of consecutive blocks (flags points at the first block's)
  unsigned* place = nullptr;                                       // ... [n_utts] placement words of this launch (zeroed beforehand): byte h = XCD + 1 of workgroup (w, h)
  int times_layer = 0;                                             // ... which block of the launch `times` stamps
  int st_in_n = 16;                                                // partials per row in st_in: 16 (a GEMM epilogue's 32-column groups) or 4 (the 8-wave kernel's own records, one per workgroup, slots 0..3)
  int opt = 0;                                                     // tuning switches of the 8-wave kernel (ASR_SANM_BLOCK8_OPT): 1 = no L2 warm-up loads, 2 = deeper W fragment queues, 4 = always acquire-fence at an exchange (default: clusters that share an XCD read the payload with sc1 loads instead)
  int ffnk = 0;                                                    // round 6: FFN-2 K-split over the workgroup's own hidden columns, f16 partials exchanged instead of hid (needs the matching wpack order; `hid` then holds the partial images)
  const void* wpack = nullptr;                                     // round-4 kernel (sanm_block8.hip): fragment-major copy of the four matrices (launch_sanm_block8_pack)
  const bf16_t* x_lo; const float2* st_in;                         // block input rows (bf16) + their row statistics [rows][16] (null: derived in the kernel)
  float* x;                                                        // residual stream f32 [rows][512]: read (phase B) and overwritten (phase D) in place
  bf16_t* x_lo_out; float2* st_out;                                // bf16 copy + row statistics of the block output (may alias x_lo / st_in)
  bf16_t* ctx; bf16_t* x1_lo; float2* st1; bf16_t* hid;            // exchange buffers: [rows][512], [rows][512], [rows][16], [rows][2048]
  const UttPlan* plan; int utt0, n_utts;                           // windows utt0 .. utt0 + n_utts - 1 (all <= 144 rows)
  unsigned* flags;                                                 // [n_utts][4] exchange counters of THIS launch, zeroed beforehand
  unsigned* err;                                                   // err[0] != 0 after the run: a workgroup gave up waiting for its cluster
  int n_rows_alloc; float ln_eps; int scatter;                     // scatter != 0: test placement (a cluster spread over four XCDs)
  unsigned long long* times;                                       // tuning: [workgroups][16] wall-clock stamps (100 MHz) at the phase boundaries, or null
  int fault = 0;                                                   // tests: workgroup 5 withholds its first exchange count (its cluster then gives up after the bounded spin)
};
bool sanm_block_supported(int max_T, int d_head, int n_heads, int d, int d_ffn, int fsmn_taps);
int sanm_block_max_utts();                                         // windows one launch can take (all workgroups co-resident)
void launch_sanm_block8(const SanmBlockArgs& a, hipStream_t s);       // round-4 form: 8 waves, chunked A operand, register-streamed packed weights (needs a.wpack)
size_t sanm_block8_pack_bytes();                                      // bytes of one block's packed weights
void launch_sanm_block8_pack(const bf16_t* wqkv, const bf16_t* wout, const bf16_t* w1, const bf16_t* w2, void* dst, bool ffnk, hipStream_t s);
void launch_rows_to_bf16(const float* x, bf16_t* y, size_t n, hipStream_t s);     // f32 -> bf16 (RNE) copy, n a multiple of 8
void launch_sanm_qkv_attn(const SanmFusedArgs& a, hipStream_t s);

// ---- FSMN memory: depth-wise conv (k taps, zero padded inside each utterance) over V + bias.
// vt: [C][ld] time-contiguous (all rows of the padded packed layout); out: f32 row-major [n_rows_pad][ld_out]
// (Export_SenseVoice.py:217-220,240-244). n_rows_pad is a multiple of 64 (pad rows are written as zeros).
template <typename InT>
void launch_fsmn(const InT* vt, int ld, const float* w, const float* b, int C, int ktaps, const UttPlan* plan,
                 const int32_t* row_utt, int n_rows_pad, float* out, int ld_out, hipStream_t s);

// ---- CTC greedy collapse, circular next-neighbour rule (Export_SenseVoice.py:290-296)
void launch_ctc_collapse(const int32_t* frame_ids, const UttPlan* plan, int n_utts, int blank_id, int32_t* token_ids,
                         int max_tokens, int32_t* num_id, hipStream_t s);

// ---- Paraformer CIF predictor + decoder helpers (Paraformer/Non-Streaming/Export_Paraformer.py:499-563)
// [x[t-1] | x[t] | x[t+1]] rows with zero padding at utterance edges: the k=3 conv as one GEMM (K = 3 d)
template <typename T>
void launch_shift3(const T* x, int d, const UttPlan* plan, const int32_t* row_utt, int n_rows, T* out, hipStream_t s);
// alpha[m] = sigmoid(dot(h[m], w) + b)
template <typename T>
void launch_alpha(const T* h, int d, const float* w, const float* b, int rows, float* alpha, hipStream_t s);
// continuous integrate-and-fire: float64 prefix sum of alpha (+ tail threshold), fire where floor() increments, acoustic
// embedding = difference of completed prefix integrals. Writes the token rows in the utterance's own row range and a
// DEVICE-side token plan (row_off, T = max(N,1), n_lfr = N) so the decoder needs no host round trip.
void launch_cif_scan(const float* alpha, const float* enc_out, int d, const UttPlan* plan, int n_utts, float tail_threshold,
                     float* acoustic, UttPlan* token_plan, int32_t* num_id, hipStream_t s);
// out[t] = res[t] + depth-wise conv (k taps, zero padded inside the token sequence) of x, all row-major f32
void launch_fsmn_rows(const float* x, const float* res, const float* w, int d, int ktaps, const UttPlan* token_plan,
                      const int32_t* row_utt, int n_rows, float* out, hipStream_t s, const int32_t* rows_dev = nullptr);
// compact token layout: utterance b's max(N_b, 1) token rows move from its own row range to offset sum_{j<b} roundup16(max(N_j, 1));
// writes the compact plan, the row -> utterance map of the compact rows (-1 past the end, n_rows_max entries) and the total row
// count (device side: the decoder's GEMM / LayerNorm launches are sized for the worst case and stop at this count)
void launch_token_compact(const UttPlan* own_plan, int n_utts, int n_rows_max, UttPlan* compact_plan, int32_t* row_utt, int32_t* total_rows,
                          hipStream_t s);
void launch_compact_rows(const float* src, const UttPlan* own_plan, const UttPlan* compact_plan, int n_utts, int d, float* dst, hipStream_t s);
// token_ids[b][i] = ids[row_off_b + i], i < N_b
void launch_gather_tokens(const int32_t* ids, const UttPlan* token_plan, int n_utts, int32_t* token_ids, int max_tokens, hipStream_t s);


// ---- streaming Paraformer (Paraformer/Streaming/Export_Paraformer_Streaming.py:386-553). Every active stream owns a 16-row slot
// (13 rows used: 4 carried + 9 new); UttPlan.lang carries the stream id that indexes the per-stream state.
struct StreamLfrArgs {
  const float* mel; const UttPlan* plan; const float* cmvn_vars; const float* pos_bias;   // pos_bias[p] = means * vars + position p + 1
  const float* prev; const int32_t* start;      // state: carried rows [stream][n_prev][ld], absolute LFR position per stream
  float* out; int ld; int feat, n_mels, lfr_m, lfr_n, n_prev, n_new, n_rows, n_frames, pos_rows;
};
void launch_stream_lfr(const StreamLfrArgs& a, hipStream_t s);
// prev[stream] = rows [n_new .. n_new + n_prev) of the slot (the last n_prev of the n_prev + n_new rows); start[stream] += n_new
void launch_stream_carry(const float* x, int ld, const UttPlan* plan, int n_active, int n_prev, int n_new, float* prev, int32_t* start,
                         hipStream_t s);
// soft-max attention of the slot's query rows over [cached keys of the stream | the slot's current rows], head_dim 128
struct StreamAttnArgs {
  const void* q; int ld_q, q_col0;
  const void* k; int ld_k, k_col0; const void* v; int ld_v, v_col0; int n_cur;      // current rows (row-major, operand dtype)
  const void* cache_k; const void* cache_v; const int32_t* cache_len; int cap;        // [stream][head][cap][128]
  const UttPlan* q_plan;        // T = number of query rows (0 => nothing to do), row_off, lang = stream id
  int n_heads;
  void* ctx; int ld_ctx;
  // encoder layers: the same workgroup also (a) rolls its (stream, head) K / V history -- keep the last `cap` of (history ++ the first
  // roll_rows current rows) -- and (b) computes the FSMN memory term of the chunk rows (depth-wise conv over the CURRENT rows of V, taps
  // outside them are zero) for its 128 channels; both null / 0 for the decoder's cross-attention
  int roll_rows = 0;
  const float* fsmn_w = nullptr; const float* fsmn_b = nullptr; int ktaps = 0; float* mem = nullptr; int d = 0;
};
template <typename T> void launch_stream_attn(const StreamAttnArgs& a, int n_active, hipStream_t s);
// cache = last `cap` of (cache ++ current rows [0, n_app)); skipped for streams whose cond_plan[i].T == 0 (when cond_plan is given)
template <typename T>
void launch_stream_cache_roll(void* cache_k, void* cache_v, const int32_t* cache_len, int cap, const void* k, int ld_k, int k_col0,
                              const void* v, int ld_v, int v_col0, int n_app, const UttPlan* plan, const UttPlan* cond_plan, int n_active,
                              int n_heads, hipStream_t s);
// encoder FSMN over the slot's n_cur rows (zero padded at both ends), identity folded into the centre tap: mem = b + conv(v)
template <typename T>
void launch_stream_fsmn(const T* v, int ld_v, int v_col0, const float* w, const float* b, int d, int ktaps, int n_cur, int n_rows, float* mem,
                        hipStream_t s);
// unrolled integrate-and-fire over rows [0, n_int) of every slot with the carried (hidden, alphas) (:438-462)
void launch_stream_cif(const float* alpha, const float* enc, int d, const UttPlan* plan, int n_active, int n_int, float* cif_hidden,
                       float* cif_alphas, float* frames_out, UttPlan* token_plan, int32_t* num, hipStream_t s);
// decoder FSMN over [hist | tokens] (valid conv, identity folded into the LAST tap) + residual; history advances when tokens exist
void launch_stream_dec_fsmn(const float* x, const float* res, const float* w, int d, int ktaps, const UttPlan* token_plan, int n_active,
                            float* hist, float* out, hipStream_t s);
void launch_stream_advance(const UttPlan* plan, const UttPlan* token_plan, int n_active, int en_add, int en_cap, int de_add, int de_cap,
                           int32_t* en_len, int32_t* de_len, hipStream_t s);
// Snapshot / restore of the recurrent state of the ACTIVE streams of a chunk step (K/V histories of every layer, decoder FSMN histories, carried rows, CIF
// state, history lengths): a fused chunk step whose cluster launch gave up has rolled the histories of the layers in front of the one that stalled, so the step
// can only be redone (on the per-launch path) from a copy taken in front of it. A segment = one state buffer laid out [outer][stream][per_stream bytes];
// the shadow has the same layout. restore = false: live -> shadow, true: shadow -> live.
struct StreamStateSeg { unsigned char* live; unsigned char* shadow; unsigned long long per_stream, outer_stride; int n_outer, first_item; };
void launch_stream_state_copy(const StreamStateSeg* segs, int n_segs, int n_items, const UttPlan* plan, int n_active, bool restore, hipStream_t s);

// ---- Paraformer online encoder layers of one chunk step as ONE launch (stream_layers.hip): clusters of four workgroups per stream, weights streamed
// from a fragment-major copy of every layer. bf16 sessions, d = 512 / 4 heads / d_ffn = 2048 / history + 16 <= 64 keys.
struct StreamLayer {
  const unsigned char* wpack;                               // launch_stream_layers_pack of this layer's q|k|v, out, w1, w2
  const float *bqkv, *wfsmn, *bfsmn, *b1, *b2;
  bf16_t* cache_k; bf16_t* cache_v;                        // this layer's histories [stream][head][cap][128]
};
struct StreamLayersArgs {
  const UttPlan* plan;                                      // per active stream: row_off (its 16-row slot), lang = stream id
  int n_streams, n_layers, n_cur, cap, roll_rows, ktaps;
  float ln_eps;
  const int32_t* cache_len;                                 // [stream] history rows (the same for every layer)
  const StreamLayer* layers;                                // device table
  float* x;                                                 // [rows][512] residual stream: in = rows entering the first layer of the table, out = rows leaving the last
  float* xb; bf16_t* ctx; bf16_t* hid;                      // the clusters' exchange buffers [rows][512] f32, [rows][512], [rows][2048]
  unsigned* flags;                                          // [n_layers][n_streams][4] counters, zero at launch
  int opt = 0;                                              // tuning: 1 = no L2 warm-up of the next phase's weights, 2 = FFN weights warmed while waiting for exchanges 1 / 2 instead of under the attention, 8 = a stream's four heads on one XCD instead of placement by head; bits 4..6 = phases (A, C, D) whose second weight batch is requested right behind the exchanged rows
  unsigned* err;                                            // raised by a cluster that gave up waiting (the launch's results are void)
  unsigned long long* times = nullptr; int times_layer = 0; // tuning: thread 0 of every workgroup stamps wall_clock64() at 13 points of layer `times_layer` ([wg][16])
  int fault = 0;                                            // tests: workgroup 5 withholds its count on the first exchange of the first layer (its cluster gives up)
};
size_t stream_layers_pack_bytes();
void launch_stream_layers_pack(const bf16_t* wqkv, const bf16_t* wout, const bf16_t* w1, const bf16_t* w2, void* dst, hipStream_t s);
bool stream_layers_supported(int d, int d_ffn, int n_heads, int cap, int n_cur, int ktaps);
void launch_stream_layers(const StreamLayersArgs& a, hipStream_t s);

// ---- SANM blocks of a SMALL batch of <= 144-row windows, a run of blocks as one launch (sanm_tiles.hip): a workgroup per (16-row tile, head), the streaming
// encoder's weight images (launch_stream_layers_pack) and layer table entries (cache pointers unused)
struct SanmTilesArgs {
  const UttPlan* plan;                                      // per window: T, row_off
  const int32_t* tile_win; const int32_t* tile_idx;         // per tile cluster: its window, its tile inside the window
  int n_tiles, n_windows, n_layers;
  float ln_eps;
  const StreamLayer* layers;                                // device table
  float* x;                                                 // [rows][512] residual stream, in place
  float* xb; bf16_t* ctx; bf16_t* hid;                      // exchange buffers of a tile's four heads: [rows][512] f32, [rows][512], [rows][2048]
  bf16_t* kv; size_t kv_parity_stride;                      // exchange buffer of a head's tiles: [block parity][rows][k | v][512], elements between the two
  unsigned* flags; int flag_stride;                         // per layer: [n_tiles][4] + [n_windows][4] counters, zero at launch; flag_stride = words per layer
  unsigned* err;
  int opt = 0;                                              // tuning: 1 = no L2 warm-up, 2 = a tile's four heads on one XCD instead of placement by head
  unsigned long long* times = nullptr; int times_layer = 0;
};
bool sanm_tiles_supported(int max_T, int d, int d_ffn, int n_heads, int d_head, int ktaps);
int sanm_tiles_max_tiles();
void launch_sanm_tiles(const SanmTilesArgs& a, hipStream_t s);

// ---- Paraformer online decoder layers of one chunk step as ONE launch (stream_dec.hip): the same clusters; a stream without a fired token leaves at once
struct StreamDecLayer {
  const unsigned char* wpack;                               // launch_stream_dec_pack of this layer's w1, w2, wq, wkv, wo
  const float *b1, *b2, *n2_g, *n2_b, *wfsmn, *bq, *bkv, *bo;
  float* fsmn_hist;                                         // [stream][10][512] f32
  bf16_t* cache_k; bf16_t* cache_v;                        // [stream][head][cap][128]
  int full;                                                 // 0: FFN-only block (the last one)
};
struct StreamDecArgs {
  const UttPlan* token_plan;                                // per active stream: T = fired tokens, row_off (its 16-row slot), lang = stream id
  int n_streams, n_layers, n_cur, cap;
  float ln_eps;
  const int32_t* cache_len;                                 // [stream] K/V history rows
  const StreamDecLayer* layers;                             // device table
  const bf16_t* enc;                                        // [rows][512] the chunk's encoder rows (after_norm, bf16)
  float* dec;                                               // [rows][512] fired frames in, decoder output rows out
  float* x1; float* x2; float* hid; bf16_t* ctx;            // exchange buffers [rows][512] f32 x 2, [rows][2048] f32, [rows][512] bf16
  unsigned* flags;                                          // [n_layers][n_streams][8] counters, zero at launch
  unsigned* err;
  int opt = 0;                                              // tuning: 1 = no L2 warm-up, 8 = placement by head (the encoder launch's default; slower here)
  unsigned long long* times = nullptr; int times_layer = 0;
};
size_t stream_dec_pack_bytes();
void launch_stream_dec_pack(const bf16_t* w1, const bf16_t* w2, const bf16_t* wq, const bf16_t* wkv, const bf16_t* wo, bool full, void* dst, hipStream_t s);
bool stream_dec_supported(int d, int d_ffn, int n_heads, int cap, int n_cur, int ktaps);
void launch_stream_dec(const StreamDecArgs& a, hipStream_t s);

