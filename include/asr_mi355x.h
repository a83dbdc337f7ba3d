/* asr_mi355x.h -- C ABI of the MI355X-native ASR hot path (libasr_mi355x.so).
 *
 * This library replaces what the reference executes behind
 *   onnxruntime.InferenceSession.run_with_iobinding(binding, run_options)
 * (process->native boundary: SenseVoice/Inference_SenseVoice_ONNX.py:303,
 *  Whisper/Inference_Whisper_ONNX.py:247-248,640), i.e. the arithmetic that
 * Export_*.py freezes into the .onnx graphs: in-graph STFT/FBank front-end,
 * encoder, and CTC / autoregressive decoder. The reference has no native code of
 * its own; its FFI for this path is the onnxruntime Python binding, so the
 * reference-side stub that binds these symbols is a ctypes shim exposing the
 * onnxruntime API subset the Inference_*_ONNX.py scripts use.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary;
 *  - every function returns ASR_OK (0) or an error code; asr_last_error() returns a
 *    thread-local message (the Python shim raises it, mirroring ORT's exceptions;
 *    nothing is swallowed);
 *  - a session owns one HIP stream (or borrows the caller's, asr_session_set_stream)
 *    and is re-entrant per session; runs are synchronous: outputs are complete when the
 *    call returns (RunOptions "disable_synchronize_execution_providers"="0",
 *    SenseVoice/Inference_SenseVoice_ONNX.py:142-147);
 *  - there is NO CPU fallback: every entry point fails with ASR_ERR_NO_DEVICE when no
 *    gfx950 device is visible.
 */
#ifndef ASR_MI355X_H
#define ASR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_ABI_VERSION 1

enum asr_status {
  ASR_STATUS_OK = 0,
  ASR_STATUS_INVALID = 1,
  ASR_STATUS_HIP = 2,
  ASR_STATUS_NOT_FOUND = 3,
  ASR_STATUS_UNSUPPORTED = 4,
  ASR_STATUS_NO_DEVICE = 5
};

/* Arithmetic mode. BF16: bf16 MFMA operands, f32 accumulate, f32 residual stream / LayerNorm /
 * soft-max / front-end. F32: exact-f32 MFMA everywhere (verification mode for the
 * "logits within 1e-3" parity bar). The weight arena must be built for the same mode.
 *
 * FP8W (opt-in, Whisper and Qwen3-ASR sessions; the low-bit counterpart of the reference's quantised decoders, Whisper/Optimize_ONNX.py:81-96,
 * Optimize_ONNX_Common.py:27,55-60, README.md:70 q4f32): bf16 mode whose decoder projection weights (Whisper: and the cross-K/V cache) are stored as
 * OCP e4m3 bytes with power-of-two scales (per output column / per (sequence, head) slab) and widened to bf16 in registers -- activations, accumulation,
 * self-KV cache, encoder and vocabulary projection are unchanged. Takes a bf16 arena; quantisation happens at session creation. Qwen3-ASR: the four
 * projections of every decoder layer (q|k|v, o, gate|up, down); decode steps of <= 64 rows stream the bytes, every other pass (prefill, wider beam steps)
 * reads their exact bf16 dequantisation, so all passes of a session see the same effective weights. Needs d_model, d_ffn and heads x head_dim multiples of 256.
 *
 * FP8MM (opt-in, Whisper sessions only): FP8W plus the encoder's feed-forward pair on the FP8 MATRIX pipe -- fc1 / fc2 weights as e4m3 bytes with one
 * power-of-two scale per output column, their activation operands as e4m3 bytes: the second LayerNorm's output at unit scale (its affine pair is folded
 * into fc1, so |value| <= sqrt(d_model) < 448), the GELU output as value * 2^-shift (asr_whisper_set_fp8_act_shift, default 0) saturating at 448 -- every
 * element that meets the clamp is counted (asr_whisper_fp8_stats): a count that moves says the shift is too small for this checkpoint --, products on v_mfma_scale_f32_16x16x128_f8f6f4 at unit block scales with f32 accumulation; attention, the other projections,
 * the residual stream and the decoder are FP8W's. Needs d_model and d_ffn multiples of 256. The reference's counterpart: its MatMulNBits /
 * dynamic-int8 graphs (Optimize_ONNX_Common.py:55-60).
 *
 * MXFP4W (opt-in, Whisper and Qwen3-ASR sessions; round 6): FP8W with the decoder projection weights stored as OCP MXFP4 instead of e4m3 -- e2m1 elements, two per byte,
 * one e8m0 scale per 32 consecutive input channels (4.25 bits per weight; the 4-bit counterpart of the reference's MatMulNBits Q4 graphs, README.md:70 q4f32).
 * The block of the format is the K-step of the bf16 MFMA, so a decode-step fragment is widened -- exactly: 2 significant bits x a power of two -- by
 * v_cvt_scalef32_pk_bf16_fp4 with its block's scale; every other pass reads the exact bf16 dequantisation. Whisper's cross-K/V cache is FP8W's; Qwen3-ASR: the four
 * projections of every decoder layer (the reference's own headline for this family is a 4-bit decoder, README.md:70 q4f32). */
enum asr_precision { ASR_PRECISION_BF16 = 0, ASR_PRECISION_F32 = 1, ASR_PRECISION_FP8W = 2, ASR_PRECISION_FP8MM = 3, ASR_PRECISION_MXFP4W = 4 };

enum asr_mem { ASR_MEM_HOST = 0, ASR_MEM_DEVICE = 1 };

typedef struct asr_session asr_session;

int asr_abi_version(void);
const char* asr_last_error(void);
int asr_device_count(int* count);

/* Foreign kernels on a GPU this library computes on -- the RCCL collectives of the data-parallel path (weight-arena broadcast, hypothesis
 * gather; the reference has no counterpart, its runtime is one onnxruntime session per process) or any other kernel the host program launches itself.
 * The SANM block / tile kernels and the fused streaming step need every workgroup of their grid resident at once; a foreign kernel holding a few CUs can
 * split a cluster (the launch then gives up after a bounded spin and the batch is redone cluster-free). So bracket foreign work:
 *   asr_device_foreign_begin(dev)  blocks until no cluster pass is in flight on `dev`; until the matching _end every compute call on `dev` takes its
 *                                  cluster-free path (same results; the call never waits, so the two sides cannot deadlock);
 *   asr_device_foreign_end(dev)    after the foreign kernels have FINISHED on the device (synchronise their stream first).
 * Sections nest (a count per device, process-wide). _stats: out4 = {sections opened, cluster passes diverted to the cluster-free path, sections that had
 * to wait for a cluster pass, cluster passes admitted}. dist.py brackets every torch.distributed collective on a CUDA device with this pair. */
int asr_device_foreign_begin(int device_id);
int asr_device_foreign_end(int device_id);
int asr_device_foreign_stats(int device_id, int64_t* out4);

/* ------------------------------------------------------------------ SenseVoice (non-AR, CTC)
 * Replaces SenseVoiceSmall.onnx == SENSE_VOICE.forward (SenseVoice/Export_SenseVoice.py:271-296).
 * Graph I/O it mirrors (Export_SenseVoice.py:375-379): audio (1,1,audio_len) f32 int16-range,
 * language_idx (1,) i32  ->  token_ids (num_token,) i32, num_id (1,) i32.
 * Extension over the batch-1 reference: B independent utterances per call, ragged lengths. */
typedef struct asr_sensevoice_config {
  int32_t sample_rate, n_mels, nfft, win_length, hop_length;
  int32_t lfr_m, lfr_n;
  int32_t d_model, n_heads, d_head, d_ffn;
  int32_t n_blocks;       /* total SANM blocks (encoders0 + encoders + tp_encoders) */
  int32_t n_main;         /* blocks before after_norm (encoders0 + encoders)          */
  int32_t fsmn_kernel;
  int32_t vocab, blank_id;
  int32_t n_prompt;       /* prompt rows prepended to the speech rows (1 language + 3 system) */
  int32_t n_languages;    /* rows of the language embedding table */
  int32_t max_audio_len;
  int32_t reserved[8];
} asr_sensevoice_config;

/* `arena` is the flat weight arena built by arena.py (manifest + tensors). With
 * ASR_MEM_HOST it is copied to HBM; with ASR_MEM_DEVICE the pointer is borrowed and must outlive
 * the session (same ownership rule as SessionOptions.add_initializer,
 * Whisper/Inference_Whisper_ONNX.py:242-243) -- used after the RCCL weight broadcast. */
int asr_sensevoice_create(const asr_sensevoice_config* cfg, const void* arena, size_t arena_bytes, int arena_mem,
                          int device_id, int precision, asr_session** out);

/* audio: packed f32 samples; utterance b spans [audio_offsets[b], audio_offsets[b+1]).
 * language_idx: B selector indices (host). token_ids_out: host [B][max_tokens] (row b holds num_id_out[b]
 * ids, rest untouched); num_id_out: host [B]. Fails if any utterance is shorter than one frame. */
int asr_sensevoice_run(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                       const int32_t* language_idx, int32_t* token_ids_out, int max_tokens, int32_t* num_id_out);

/* sequence length (prompt + LFR rows) the graph produces for an utterance of n_samples */
int asr_sensevoice_seq_len(const asr_sensevoice_config* cfg, int n_samples, int* seq_len);
/* Counters of a SenseVoice / Paraformer session's cluster kernels: out8 = {forward passes redone cluster-free because a cluster gave up, batches left on the
 * cluster-free path after the last give-up, passes that took the cluster-free path because a foreign section was open (asr_device_foreign_begin),
 * block kernel enabled, 0, 0, 0, 0}. */
int asr_sanm_stats(asr_session* s, int32_t* out8);

/* ------------------------------------------------------------------ Paraformer (non-streaming, CIF + NAR decoder)
 * Replaces Paraformer.onnx == PARAFORMER.forward (Paraformer/Non-Streaming/Export_Paraformer.py:474-563).
 * Graph I/O it mirrors (:600-606): audio (1,1,audio_len) f32 int16-range -> token_ids (1,num_token) i32, num_id (1,) i32. */
typedef struct asr_paraformer_config {
  int32_t sample_rate, n_mels, nfft, win_length, hop_length, lfr_m, lfr_n;
  int32_t d_model, n_heads, d_head, d_ffn, n_blocks, fsmn_kernel;
  int32_t n_dec, n_dec3, d_dec_ffn, cif_kernel, vocab, max_audio_len;
  float tail_threshold;
  int32_t reserved[8];
} asr_paraformer_config;

int asr_paraformer_create(const asr_paraformer_config* cfg, const void* arena, size_t arena_bytes, int arena_mem, int device_id,
                          int precision, asr_session** out);
/* same calling convention as asr_sensevoice_run (no language input): token_ids_out host [B][max_tokens], num_id_out host [B]
 * (num_id is the CIF fire count; an utterance can legitimately yield zero tokens). */
int asr_paraformer_run(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                       int32_t* token_ids_out, int max_tokens, int32_t* num_id_out);

/* Streaming Paraformer: replaces Paraformer_Streaming_Encoder.onnx + Paraformer_Streaming_Decoder.onnx (PARAFORMER_ENCODER /
 * PARAFORMER_DECODER, Paraformer/Streaming/Export_Paraformer_Streaming.py:330-553) and the state shuttling of
 * Inference_Paraformer_Streaming_ONNX.py:309-449. The 100 in_en_key/value tensors, in_previous_mel_features, in_cif_hidden,
 * in_cif_alphas, start_idx, in_de_fsmn / in_de_key / in_de_value of a stream live in HBM inside the session, indexed by a
 * stream id in [0, max_streams); one step advances n_streams DIFFERENT streams by one chunk_samples-sample chunk each
 * (audio: [n_streams][chunk_samples], int16-range float) and runs the decoder for the streams whose CIF fired -- streams
 * without a fired frame keep their decoder state, exactly like the host loop (:420-447). token_ids_out: host
 * [n_streams][max_tokens] (max_tokens >= LFR rows per chunk + 1), num_id_out: host [n_streams]. The arena is the
 * non-streaming Paraformer arena built for streaming (position table up to MAX_CONTINUE_STREAMING, causal decoder FSMN). */
int asr_paraformer_stream_create(const asr_paraformer_config* cfg, const void* arena, size_t arena_bytes, int arena_mem, int device_id,
                                 int precision, int chunk_samples, int look_back_encoder, int look_back_decoder, int max_streams,
                                 asr_session** out);
int asr_paraformer_stream_reset(asr_session* s, int stream_id);      /* -1 = every stream: empty histories, zero CIF state */
int asr_paraformer_stream_step(asr_session* s, const float* audio, int audio_mem, const int32_t* stream_ids, int n_streams,
                               int32_t* token_ids_out, int max_tokens, int32_t* num_id_out);
/* Which path the chunk steps of a streaming session took (no reference counterpart: the reference runs one stream per InferenceSession and has no
 * co-tenancy to manage). A bf16 step runs the encoder / decoder layer loops as two cluster launches when (i) no more than fused_max streams are active,
 * (ii) no other session of this process is computing on the same GPU (otherwise the per-launch path, which leaves CUs to the other tenant and waits on nobody);
 * when other sessions merely exist on the GPU the step first snapshots the active streams' recurrent state, so that a cluster launch that gave up is restored and
 * redone on the per-launch path instead of costing the streams their history. out8 = {give-ups recovered, steps that took the per-launch path because the GPU
 * was shared, snapshots taken, fused_max, steps left of the per-launch cool-down after a give-up, 1 if the session can fuse at all, 0, 0}. */
int asr_paraformer_stream_stats(asr_session* s, int32_t* out8);

/* ------------------------------------------------------------------ Whisper (encoder + KV-cache decoder)
 * Replaces the merged graphs Whisper_ProbePrefillGreedy / Whisper_PrefillGreedy / Whisper_DecodeGreedy
 * (Whisper/Shared_Merged.py:864-888; I/O planner Whisper/Inference_Whisper_ONNX.py:323-392), i.e.
 * STFT_Process + WHISPER_ENCODER.forward (Export_Whisper.py:422-447), WHISPER_DECODER_EMBED / _PREFILL / _DECODE /
 * WHISPER_DECODER.forward (:450-497,614-667) and the BEGIN_SUPPRESS / ARGMAX heads (:228-260).
 * The reference passes 2 x n_layers self-KV and 2 x n_layers cross-KV tensors through Python on every call; here
 * they are session state: `encode` fills the cross-KV slabs, `prefill` resets and fills the self-KV cache (paged: 16-position
 * pages from a per-session pool behind a block table that grows with the sequences; ASR_KV_PAGED=0 keeps contiguous extents), `decode`
 * appends one position in place. Batch B = independent utterances (the reference is batch 1). */
typedef struct asr_whisper_config {
  int32_t sample_rate, n_mels, nfft, hop_length;
  int32_t d_model, n_heads, d_head, d_ffn, n_enc_layers, n_dec_layers;
  int32_t vocab, max_source_positions, max_target_positions, max_audio_len;
  int32_t gelu_tanh;      /* 0 = erf GELU as exported (:428); 1 = tanh GELU, what ORT runs with
                             optimization.enable_gelu_approximation=1 (Inference_Whisper_ONNX.py:166) */
  int32_t reserved[9];
} asr_whisper_config;

int asr_whisper_create(const asr_whisper_config* cfg, const void* arena, size_t arena_bytes, int arena_mem, int device_id,
                       int precision, asr_session** out);
/* audio: packed f32 samples in [-1, 1] (audio_pcm_scale 32768, Export_Whisper.py:1068); n_positions_out (host, B,
 * nullable) receives the encoder length (n_samples / 160 + 1) / 2 of each utterance. */
int asr_whisper_encode(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                       int32_t* n_positions_out);
/* ids: host [B][n] prompt tokens (n <= 8), history is reset to 0 like the reference's prefill. next_ids_out (host B,
 * nullable): arg-max(logits + begin_suppress); logits_out (host [B][vocab], nullable): raw logits incl. the -128
 * suppress penalty -- what language detection and NO_SPEECH_DETECTION read (Inference_Whisper_ONNX.py:780-805). */
int asr_whisper_prefill(asr_session* s, const int32_t* ids, int n, int32_t* next_ids_out, float* logits_out);
/* one position per sequence. ids: host [B], or NULL to feed the device-resident arg-max of the previous step (no
 * host round trip); next_ids_out / logits_out nullable (NULL, NULL => fully asynchronous step). */
int asr_whisper_decode(asr_session* s, const int32_t* ids, int32_t* next_ids_out, float* logits_out);
/* greedy continuation after a prefill: tokens_out host [B][max_new], n_out host [B]; stops per sequence at eos_id
 * (not emitted) -- the loop of _decode_tokens (Inference_Whisper_ONNX.py:584-663); the head is the one selected by
 * asr_whisper_set_penalty (plain arg-max by default). */
int asr_whisper_generate(asr_session* s, int max_new, int eos_id, int32_t* tokens_out, int32_t* n_out);
/* decode head: repeat_penalty == 1 => ARGMAX (Export_Whisper.py:254-260); otherwise penalty-greedy = APPLY_PENALTY (:312-325)
 * + GREEDY_SEARCH (:243-251): the logits of the last penalty_range generated ids (1..64) are multiplied by repeat_penalty
 * once penalty_range ids exist (REPEAT_PENALTY / PENALTY_RANGE, Inference_Whisper_ONNX.py:78-79,630-632). The id history is kept
 * on the device and restarts at every prefill. Applies to prefill / decode / generate; logits_out then holds the
 * penalised logits. */
int asr_whisper_set_penalty(asr_session* s, float repeat_penalty, int penalty_range);
/* FP8MM sessions: the GELU operand of fc2 is stored as value * 2^-shift (0 .. 16; also ASR_FP8MM_ACT_SHIFT at creation); takes effect at the next encode.
 * asr_whisper_fp8_stats: stats[0] = activation elements that met the e4m3 clamp (|value * 2^-shift| > 448, or NaN) since the session was created,
 * stats[1] = the shift in use. Zero / the shift on sessions of other precisions. No reference counterpart (its low-bit graphs quantise activations
 * dynamically per tensor, Optimize_ONNX_Common.py:55-60); here the scale is static and the counter is what makes a wrong one loud. */
int asr_whisper_set_fp8_act_shift(asr_session* s, int shift);
int asr_whisper_fp8_stats(asr_session* s, uint64_t stats[2]);

/* The *PenaltyGreedy graphs run GREEDY_SEARCH (:243-251) on every step, so save_id grows even while the host still feeds
 * penalty_penalty_value = 1.0 (before PENALTY_RANGE ids exist, Inference_Whisper_ONNX.py:630-632): enable = 1 keeps appending the
 * picks to the device-side history whatever the penalty value is (the onnxruntime shim drives the value per step). */
int asr_whisper_track_history(asr_session* s, int enable);
/* NO_SPEECH_DETECTION head (Export_Whisper.py:334-348; graph Whisper_No_Speech_Detection.onnx, host :799-805) on the logits of the last
 * prefill (the probe with [SOT]): prob_out host [B] = softmax(logits + 128 on the permanently suppressed ids)[no_speech_id]. */
int asr_whisper_no_speech_prob(asr_session* s, int no_speech_id, float* prob_out);
/* decode head TOPK_TOPP_SAMPLING (Export_Whisper.py:263-308; USE_SAMPLING / TEMPERATURE / TOP_K / TOP_P /
 * SAMPLING_REPETITION_PENALTY, Inference_Whisper_ONNX.py:71-75): repetition penalty over every previously sampled id,
 * temperature, top-k (1..64), top-p, Gumbel-max. enable = 0 returns to the arg-max / penalty-greedy head. The reference draws
 * torch.rand_like inside the graph; here the uniforms come from a counter-based generator keyed by (seed, step, sequence, rank)
 * -- reproducible run to run, but a different stream than torch's. */
int asr_whisper_set_sampling(asr_session* s, int enable, float temperature, int top_k, float top_p, float repetition_penalty,
                             uint64_t seed);
/* parity hook: the uniforms of the NEXT prefill / decode step, host [batch][top_k] (count = batch * top_k); consumed once. */
int asr_whisper_set_sampling_noise(asr_session* s, const float* uniforms, int count);

/* ------------------------------------------------------------------ Qwen3-ASR (audio encoder + Qwen3 decoder with KV cache)
 * Replaces the merged graphs Qwen_ASR prefill_greedy / decode_greedy and the Embed graph (Qwen_ASR/Shared_Merged.py; I/O planner
 * Qwen_ASR/Inference_Qwen_ASR_ONNX.py:330-366), i.e. STFT_Process + QWEN3_ASR_ENCODER.forward (Export_Qwen_ASR.py:850-927),
 * CONCAT_EMBED (:1428-1435), ROTARY_MASK_PREFILL / _DECODE (:933-1028), DECODER_MAIN.forward (:1265-1336) and the ARGMAX head.
 * The reference hands 2 x n_layers KV tensors through Python per token (Inference_Qwen_ASR_ONNX.py:700-712); here the cache is
 * session state appended in place, and the prompt embeddings (query_embed / language_tail_embed inputs) are gathered on the
 * device from token ids. Batch B = independent utterances with their own prompts (the reference is batch 1). */
typedef struct asr_qwen_config {
  int32_t sample_rate, n_mels, nfft, hop_length;
  int32_t enc_d, enc_heads, enc_ffn, n_enc_layers, conv_channels, n_window, n_window_infer, max_source_positions;
  int32_t d_model, n_heads, n_kv_heads, d_head, d_ffn, n_layers, vocab, max_seq_len, max_audio_len;
  float rms_eps, rope_theta;
  int32_t reserved[9];
} asr_qwen_config;

int asr_qwen_create(const asr_qwen_config* cfg, const void* arena, size_t arena_bytes, int arena_mem, int device_id, int precision,
                    asr_session** out);
/* One call = the reference's prefill launch (:640-668): audio encoding, prompt assembly, rotary / causal mask, decoder prefill,
 * first-token selection. Sequence b's prompt is [pre_ids[pre_offsets[b] : pre_offsets[b+1]] | audio embeddings of utterance b |
 * post_ids[post_offsets[b] : post_offsets[b+1]]] -- pre = head + query + suffix ids, post = tail (+ language tail) ids
 * (:923-927, CONCAT_EMBED). next_ids_out (host B, nullable) = arg-max of the last position; logits_out (host [B][vocab],
 * nullable); ids_len_out (host B, nullable) = prompt length = the reference's kv_seq_len output (:672). Resets the KV cache. */
int asr_qwen_prefill(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch, const int32_t* pre_ids,
                     const int32_t* pre_offsets, const int32_t* post_ids, const int32_t* post_offsets, int32_t* next_ids_out, float* logits_out,
                     int32_t* ids_len_out);
/* one position per sequence (Embed + decode_greedy, :690-716). ids: host [B], or NULL to feed the device-resident arg-max of the
 * previous call; next_ids_out / logits_out nullable (ids NULL and both outputs NULL => asynchronous step). */
int asr_qwen_decode(asr_session* s, const int32_t* ids, int32_t* next_ids_out, float* logits_out);
/* decode heads (strategy selection Inference_Qwen_ASR_ONNX.py:369-376; graphs Shared_Merged.py merge_prefill_* / merge_decode_*):
 * repeat_penalty == 1 => ARGMAX (Export_Qwen_ASR.py:1418-1420); otherwise penalty-greedy = APPLY_PENALTY (:1403-1415, the logits of
 * save_id[:, -penalty_range:] -- fewer ids while the history is shorter -- multiplied by repeat_penalty, decode steps only) +
 * GREEDY_SEARCH (:1342-1345). set_sampling selects TOPK_TOPP_SAMPLING (:1348-1400) for prefill and decode, with the counter-based
 * uniforms of asr_whisper_set_sampling; set_sampling_noise supplies the next step's uniforms [batch][top_k] (parity hook). The id
 * history lives on the device and restarts at every prefill. */
int asr_qwen_set_penalty(asr_session* s, float repeat_penalty, int penalty_range);
int asr_qwen_track_history(asr_session* s, int enable);   /* like asr_whisper_track_history: the *_Penalty_Greedy graphs append every pick */
int asr_qwen_set_sampling(asr_session* s, int enable, float temperature, int top_k, float top_p, float repetition_penalty, uint64_t seed);
int asr_qwen_set_sampling_noise(asr_session* s, const float* uniforms, int count);
/* continuation after a prefill with the selected head (:687-745): tokens_out host [B][max_new], n_out host [B]; a sequence ends at
 * the first id in stop_ids (not emitted) or when the cache is full. */
int asr_qwen_generate(asr_session* s, int max_new, const int32_t* stop_ids, int n_stop, int32_t* tokens_out, int32_t* n_out);
/* The session's KV cache (Qwen_ASR/Export_Qwen_ASR.py:1265-1336 keeps `in_key_i / in_value_i` tensors of the whole history per sequence; the north-star's paged
 * cache): 16-position pages behind one block table for all layers, a free list on the host -- a sequence holds pages for the positions it has, and a sequence that
 * asr_qwen_generate finished gives its pages back while its neighbours go on. out4 = {1 if paged (0: extents of max_seq_len positions, ASR_QWEN_KV_PAGED=0),
 * pages in the pool, pages held by sequences now, high-water mark of pages in use since the last prefill}. */
int asr_qwen_kv_stats(asr_session* s, int32_t* out4);
/* beam search after a prefill (the README's "greedy / beam search" for Qwen3-ASR, README.md:38; the reference ships no code for it, the
 * semantics are written down in oracle/qwen_asr_oracle.py:beam_search_core): width-`beam` (1..8) search over summed log-soft-max scores,
 * no length normalisation. A hypothesis ends at the first id in stop_ids (not emitted) and then stands with its score; an utterance is
 * finished when its best hypothesis has ended (no live one can overtake it: log-probabilities are <= 0) or after max_new ids. Every
 * step is one decoder pass over all B * beam rows; the KV cache is never re-ordered (the attention kernel follows each row's
 * ancestry). Host outputs, hypotheses best-first: tokens_out [B][beam][max_new], n_out [B][beam], scores_out [B][beam] (nullable).
 * Plain arg-max head only (no penalty / sampling). The session's greedy state (asr_qwen_decode / _generate) is left untouched. */
int asr_qwen_beam_search(asr_session* s, int beam, int max_new, const int32_t* stop_ids, int n_stop, int32_t* tokens_out, int32_t* n_out,
                         float* scores_out);

/* ------------------------------------------------------------------ device buffers
 * Backing store of the shim's OrtValue (OrtValue.ortvalue_from_numpy / update_inplace / numpy,
 * SenseVoice/Inference_SenseVoice_ONNX.py:280-299): update_inplace is an H2D copy into the same
 * allocation, numpy() a blocking D2H copy. kind: 0 = host->device, 1 = device->host, 2 = device->device. */
int asr_mem_alloc(int device_id, size_t bytes, void** out);
int asr_mem_free(int device_id, void* ptr);
int asr_mem_copy(int device_id, void* dst, const void* src, size_t bytes, int kind);

/* ------------------------------------------------------------------ session utilities */
int asr_session_destroy(asr_session* s);
int asr_session_set_stream(asr_session* s, void* hip_stream);       /* borrow a caller stream (e.g. torch's) */
int asr_session_device(asr_session* s, int* device_id);

/* Per-kernel-class timing with HIP events on the session stream (bench.py roofline leg).
 * read: up to `cap` classes; names are NUL-terminated, 32 bytes apart in `names`. */
int asr_session_profile_enable(asr_session* s, int enable);
int asr_session_profile_reset(asr_session* s);
int asr_session_profile_read(asr_session* s, int cap, char* names, double* total_ms, int64_t* launches, int* n_out);

/* Debug taps (the reference's decode graphs expose no logits; the 1e-3 logit check needs one).
 * enable before a run; read copies the named f32 tensor of the LAST run to host.
 * names: "mel" "enc_in" "block0" "enc_out" "logits" "frame_ids"(i32) */
int asr_session_taps_enable(asr_session* s, int enable);
int asr_session_tap_shape(asr_session* s, const char* name, int64_t* rows, int64_t* cols);
int asr_session_tap_read(asr_session* s, const char* name, void* host_out, size_t bytes);

/* ------------------------------------------------------------------ operator-level entry points
 * Single kernels on host arrays (device temporaries are internal). Used by the parity tests to pin
 * each HIP kernel against the oracle separately; not part of the reference-facing surface. */
int asr_op_gemm(int precision, const float* a, const float* w, const float* bias, int M, int N, int K, int act,
                float* out);                                        /* out[M][N] = act(a[M][K] w[N][K]^T + bias) */
int asr_op_layernorm(int precision, const float* x, int rows, int D, const float* gamma, const float* beta, float eps,
                     float* out);
int asr_op_attention(int precision, const float* q, const float* k, const float* v, const int32_t* seq_lens, int batch,
                     int n_heads, int d_head, float* ctx);          /* packed rows, row-major [sum T][H*D] */
int asr_op_fsmn(int precision, const float* v, const float* w, const float* b, const int32_t* seq_lens, int batch,
                int channels, int ktaps, float* out);               /* v,out: [sum T][C] row-major */
/* out[M<=64][N] = LayerNorm(x[M][K]; gamma, beta) w[N][K]^T + bias  (bf16 skinny GEMM with the fused LayerNorm prologue) */
int asr_op_gemm_ln(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, int M, int N, int K,
                   float* out);
int asr_op_ctc_collapse(const int32_t* frame_ids, const int32_t* seq_lens, int batch, int blank_id, int32_t* token_ids,
                        int max_tokens, int32_t* num_id);

#ifdef __cplusplus
}
#endif
#endif /* ASR_MI355X_H */
