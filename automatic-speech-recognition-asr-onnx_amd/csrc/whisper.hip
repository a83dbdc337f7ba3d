// Whisper hot path on one MI355X.
//   encode : packed ragged batch -> STFT power / log-mel -> conv stem (two strided-view GEMMs, no im2col) ->
//            N encoder layers -> fused cross-KV projection written straight into per-(layer, head) slabs.
//            Follows WHISPER_ENCODER.forward (Whisper/Export_Whisper.py:422-447) + STFT_Process (:224-246).
//   prefill / decode : token+position embedding -> M decoder layers (self-attention with an in-place KV cache,
//            cross-attention over the slabs, FFN) -> tied proj_out + (-128) suppress penalty -> arg-max.
//            Follows WHISPER_DECODER_EMBED / WHISPER_PREFILL / WHISPER_DECODE / WHISPER_DECODER.forward
//            (:450-497,614-667) and the BEGIN_SUPPRESS / ARGMAX heads (:228-260).
// The reference grows the self-KV by torch.cat every token (O(L^2) copies, :640-641) and shuttles 128 KV tensors
// through Python per step; here the cache is appended in place and token ids stay on the device between steps.
#include <cstdlib>
#include <cstring>

#include "../../include/asr_mi355x.h"
#include "engine.h"
#include "gemm.h"
#include "kernels.h"

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct EncLayer { const void *wqkv, *wo, *w1, *w2; const float *bqkv, *bo, *b1, *b2; };
struct DecLayer { const void *wqkv, *wo, *wcq, *wco, *w1, *w2; const float *bqkv, *bo, *bcq, *bco, *b1, *b2; };
struct Dec8Layer { const unsigned char* w[6]; const float* s[6]; const unsigned char* s4[6]; };      // FP8W: bytes + per-column scales; MXFP4W: nibbles (w) + e8m0 block scales (s4)
struct Enc8Layer { const unsigned char *w1, *w2; const float *s1, *s2; };     // FP8MM mode: the encoder FFN pair as e4m3 bytes + per-row scales        // FP8 mode: e4m3 bytes + per-column scales of wqkv, wo, wcq, wco, w1, w2

struct WhSession : asr_session {
  asr_whisper_config cfg;
  int vpad = 0, n_bin_tiles = 0, n_kchunks = 0, act = ACT_GELU_ERF;
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  const float *dft = nullptr, *melp = nullptr, *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *enc_ln_g = nullptr,
              *enc_ln_b = nullptr, *ckv_b = nullptr, *dec_pos = nullptr, *suppress = nullptr, *begin = nullptr, *dec_ln_g = nullptr,
              *dec_ln_b = nullptr;
  const void *conv1_w = nullptr, *conv2_w = nullptr, *ckv_w = nullptr, *embed = nullptr;

  // encoder state of the current batch
  int batch = 0, rows = 0, Mpad = 0, hist = 0, max_T_enc = 0;
  std::vector<UttPlan> plan;
  DeviceBuffer d_plan, d_audio, d_mel, d_blkmax, d_x0, d_h1, d_xa, d_xb, d_xc, d_h, d_qk, d_vt, d_ctx, d_ffn, d_cross;
  // decoder state
  DeviceBuffer d_kc, d_vc, d_ids, d_next, d_logits, d_dx, d_dqkv, d_dtok, d_hist;
  // Paged self-KV cache (the default; ASR_KV_PAGED=0 keeps one contiguous max_target_positions extent per sequence and head in d_kc / d_vc).
  // Pool d_kvpool: pages of KV_PAGE positions, page-major [page][layer][K | V][head][KV_PAGE][64], so a page carries one 16-position slice of a
  // sequence for every layer and the pool grows by appending pages (the old pool is a prefix of the new one: one copy). Block table d_ptable
  // [batch][pages_per_seq] (int32 page ids, -1 = not allocated), shared by all layers. Pages are handed out a generation at a time (generation j = the
  // j-th page of every sequence of the batch: all sequences of a batch stand at the same position), first for the prompt + 48 positions, then doubling,
  // so a 32-token batch of 64 holds 4 x 64 pages (0.67 GB at large-v3) instead of 64 x 448 positions (9.4 GB).
  static constexpr int KV_PAGE = 16;
  DeviceBuffer d_kvpool, d_ptable;
  bool kv_paged = true;
  int kv_gens = 0, kv_batch = 0, kv_shuffle = 0;      // generations allocated, the batch they were cut for; kv_shuffle (tests): permute the page ids inside every generation
  void ensure_kv_pages(int B, int positions, size_t elem_bytes);
  DeviceBuffer d_save, d_nsaved;       // penalty-greedy: generated ids per sequence [B][max_target_positions] + their count
  bool sampling = false;               // TOPK_TOPP_SAMPLING head (USE_SAMPLING, Inference_Whisper_ONNX.py:71-75)
  float temperature = 0.8f, top_p = 0.95f, samp_rep_penalty = 1.0f;
  int top_k = 10;
  uint64_t samp_seed = 0;
  DeviceBuffer d_nsp;                  // no-speech probabilities of the last prefill
  DeviceBuffer d_noise;                // caller-supplied uniforms [B][top_k] for the next step (parity tests); consumed once
  bool noise_armed = false;
  float penalty_value = 1.0f;          // 1.0 = plain greedy (REPEAT_PENALTY, Inference_Whisper_ONNX.py:78)
  bool track_history = false;          // GREEDY_SEARCH graphs append every pick to save_id even while the penalty value is 1.0
  int penalty_range = 20;
  bool use_graph = true;
  bool use_decode_gemm = true;         // ASR_DECODE_GEMM=0: decode steps through the generic weight-streaming GEMM + LayerNorm prologues
  DeviceBuffer d_colsum, d_dlo;        // column sums of the LayerNorm-folded decoder projections; bf16 copies of the decoder's residual rows
  // precision mode ASR_PRECISION_FP8W (opt-in; everything else as in bf16 mode): the six projections of every decoder layer as e4m3 bytes with a
  // power-of-two scale per output column, read by the decode GEMM (<= 64 rows); their exact bf16 dequantisation serves every other path
  // (prefill, > 64 rows), so all steps of a session see the same effective weights. The cross-K/V slabs are quantised once per batch with a
  // scale per (sequence, head) and streamed as bytes by the decode attention. ASR_FP8_FAKE=1: same quantisation, bf16 kernels throughout.
  // precision mode ASR_PRECISION_FP8MM (opt-in): FP8W plus the encoder's FFN pair on the FP8 matrix pipe (csrc/gemm_fp8.hip): fc1 / fc2 weights as e4m3 bytes with
  // per-row power-of-two scales, their activation operands (the second LayerNorm's output, the GELU output) as e4m3 bytes at unit scale
  // (saturating at 448: the LayerNorm output cannot get there -- its affine pair is folded into fc1, |row| <= sqrt(d_model) --; the GELU output can on a real
  // checkpoint: it is stored as value * 2^-act_shift (ASR_FP8MM_ACT_SHIFT / asr_whisper_set_fp8_act_shift, default 0), fc2 multiplies the shift back, and every
  // element that still meets the clamp is counted -- asr_whisper_fp8_stats; the Python session warns when the count moves)
  bool fp8_mm = false;
  int fp8_act_shift = 0;
  std::vector<Enc8Layer> enc8;
  DeviceBuffer d_ew8, d_ewscale, d_h8, d_ffn8, d_sat;
  bool fp4 = false;                    // precision mode ASR_PRECISION_MXFP4W (opt-in): FP8W with the decoder projections as MXFP4 (e2m1 + e8m0 per 32 k) instead of e4m3
  bool fp8 = false, fp8_fake = false, fp8_weights = true, fp8_kv = true;     // ASR_FP8_WEIGHTS=0 / ASR_FP8_KV=0: leave that half in bf16 (to price the halves separately)
  std::vector<Dec8Layer> dec8;
  DeviceBuffer d_w8, d_wscale, d_wdq, d_cross8, d_cscale;
  hipGraphExec_t dec_graph = nullptr;      // the whole single-token step
  void drop_graphs() {
    if (dec_graph) { (void)hipGraphExecDestroy(dec_graph); dec_graph = nullptr; }
  }
  uint64_t dec_key = 0, dec_eager_key = 0, ws_epoch = 1;
  void* h_plan = nullptr; size_t h_plan_cap = 0;
  void* h_io = nullptr; size_t h_io_cap = 0;

  ~WhSession() override {
    for (DeviceBuffer* b : {&d_plan, &d_audio, &d_mel, &d_blkmax, &d_x0, &d_h1, &d_xa, &d_xb, &d_xc, &d_h, &d_qk, &d_vt, &d_ctx,
                            &d_ffn, &d_cross, &d_kc, &d_vc, &d_kvpool, &d_ptable, &d_ids, &d_next, &d_logits, &d_dx, &d_dqkv, &d_dtok, &d_hist, &d_save, &d_nsaved, &d_noise, &d_nsp, &d_skws, &d_skcnt, &d_colsum, &d_dlo, &d_w8, &d_wscale, &d_wdq, &d_cross8, &d_cscale, &d_ew8, &d_ewscale, &d_h8, &d_ffn8, &d_sat})
      b->release();
    drop_graphs();
    for (auto& kv : taps) kv.second.buf.release();
    if (h_plan) (void)hipHostFree(h_plan);
    if (h_io) (void)hipHostFree(h_io);
    prof.release();
    arena.release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
  void init();
  DeviceBuffer d_skws, d_skcnt;        // split-K workspace + tickets of the skinny / decode GEMM (per session: sessions may run concurrently)
  static constexpr int SK_CNT = 4096;
  static constexpr size_t SK_WS_BYTES = (size_t)16 << 20;
  void gemm(const GemmArgs& g0) {
    if (precision != ASR_PRECISION_BF16) { launch_gemm_f32(g0, stream); return; }
    if (!d_skws.ptr) { d_skws.reserve(SK_WS_BYTES, stream); d_skcnt.reserve((size_t)SK_CNT * 4, stream); }
    GemmArgs g = g0;
    g.sk_ws = d_skws.as<float>(); g.sk_ws_bytes = SK_WS_BYTES; g.sk_cnt = d_skcnt.as<int32_t>();
    launch_gemm_bf16(g, stream);
  }
  void* pinned(size_t bytes) {
    if (bytes > h_io_cap) {
      if (h_io) HIP_CHECK(hipHostFree(h_io));
      HIP_CHECK(hipHostMalloc(&h_io, bytes * 2, hipHostMallocDefault));
      h_io_cap = bytes * 2;
    }
    return h_io;
  }
  template <typename T> void encode(const float* audio, int audio_mem, const int64_t* offs, int B, int32_t* n_pos_out);
  template <typename T> void enqueue_step(const int32_t* ids_dev, int n, bool is_prefill, bool use_hist_dev);
  template <typename T> void step(const int32_t* ids_host, int n, bool is_prefill, int32_t* next_out, float* logits_out);
};

void WhSession::init() {
  const auto& c = cfg;
  ASR_REQUIRE(c.d_model == c.n_heads * c.d_head && c.d_head == 64, "whisper: head_dim must be 64 and d_model = heads * 64");
  ASR_REQUIRE(c.d_model % 128 == 0 && c.d_ffn % 128 == 0, "whisper: d_model and d_ffn must be multiples of 128");
  ASR_REQUIRE(c.nfft == 400 && c.hop_length == 160, "whisper: front-end is built for n_fft 400 / hop 160");
  ASR_REQUIRE(c.n_mels % 16 == 0 && (3 * c.n_mels) % 64 == 0, "whisper: n_mels must make 3*n_mels a multiple of 64");
  ASR_REQUIRE(c.max_target_positions <= 1536, "whisper: decoder context too long for the attention kernel");
  vpad = round_up(c.vocab, 128);
  n_bin_tiles = (c.nfft / 2 + 1 + 15) / 16;
  n_kchunks = c.nfft / 16;
  act = c.gelu_tanh ? ACT_GELU_TANH : ACT_GELU_ERF;
  const int wt = precision == ASR_PRECISION_BF16 ? ARENA_BF16 : ARENA_F32;
  const int d = c.d_model, dff = c.d_ffn, Ld = c.n_dec_layers;
  auto F = [&](const std::string& n, std::initializer_list<int64_t> sh) { return (const float*)arena.get(n, ARENA_F32, sh).ptr; };
  auto W = [&](const std::string& n, std::initializer_list<int64_t> sh) { return arena.get(n, wt, sh).ptr; };
  dft = F("fe.dft", {(int64_t)n_bin_tiles * 2 * n_kchunks * 64 * 4});
  melp = F("fe.mel", {(int64_t)(c.n_mels / 16) * n_bin_tiles * 64 * 4});
  conv1_w = W("enc.conv1_w", {d, 3 * c.n_mels});
  conv1_b = F("enc.conv1_b", {d});
  conv2_w = W("enc.conv2_w", {d, 3 * d});
  conv2_b = F("enc.conv2_b", {d});
  enc_pos = F("enc.pos", {c.max_source_positions, d});
  enc_ln_g = F("enc.ln_g", {d});
  enc_ln_b = F("enc.ln_b", {d});
  ckv_w = W("ckv.w", {2 * Ld * d, d});
  ckv_b = F("ckv.b", {2 * Ld * d});
  embed = W("dec.embed", {vpad, d});
  dec_pos = F("dec.pos", {c.max_target_positions, d});
  suppress = F("dec.suppress", {vpad});
  begin = F("dec.begin", {vpad});
  dec_ln_g = F("dec.ln_g", {d});
  dec_ln_b = F("dec.ln_b", {d});
  enc.resize(c.n_enc_layers);
  for (int i = 0; i < c.n_enc_layers; ++i) {
    const std::string p = "enc" + std::to_string(i) + ".";
    enc[i] = EncLayer{W(p + "wqkv", {3 * d, d}), W(p + "wo", {d, d}), W(p + "w1", {dff, d}), W(p + "w2", {d, dff}),
                      F(p + "bqkv", {3 * d}), F(p + "bo", {d}), F(p + "b1", {dff}), F(p + "b2", {d})};
  }
  dec.resize(Ld);
  for (int i = 0; i < Ld; ++i) {
    const std::string p = "dec" + std::to_string(i) + ".";
    dec[i] = DecLayer{W(p + "wqkv", {3 * d, d}), W(p + "wo", {d, d}), W(p + "wcq", {d, d}), W(p + "wco", {d, d}),
                      W(p + "w1", {dff, d}), W(p + "w2", {d, dff}),
                      F(p + "bqkv", {3 * d}), F(p + "bo", {d}), F(p + "bcq", {d}), F(p + "bco", {d}), F(p + "b1", {dff}), F(p + "b2", {d})};
  }
  if (fp8_mm) {
    ASR_REQUIRE(d % 256 == 0 && dff % 256 == 0, "whisper: FP8MM mode needs d_model and d_ffn to be multiples of 256");
    const size_t per = (size_t)2 * d * dff;
    d_ew8.reserve((size_t)c.n_enc_layers * per, stream); d_ewscale.reserve((size_t)c.n_enc_layers * (d + dff) * 4, stream);
    d_sat.reserve(8, stream); HIP_CHECK(hipMemsetAsync(d_sat.ptr, 0, 8, stream));
    enc8.resize(c.n_enc_layers);
    for (int i = 0; i < c.n_enc_layers; ++i) {
      unsigned char* w8 = d_ew8.as<unsigned char>() + i * per;
      float* sc = d_ewscale.as<float>() + (size_t)i * (d + dff);
      launch_quantize_rows_fp8((const bf16_t*)enc[i].w1, d, dff, d, w8, sc, nullptr, stream);
      launch_quantize_rows_fp8((const bf16_t*)enc[i].w2, dff, d, dff, w8 + (size_t)dff * d, sc + dff, nullptr, stream);
      enc8[i] = Enc8Layer{w8, w8 + (size_t)dff * d, sc, sc + dff};
    }
  }
  if (fp8 && fp8_weights) {
    ASR_REQUIRE(d % 256 == 0 && dff % 256 == 0, "whisper: FP8 / MXFP4 mode needs d_model and d_ffn to be multiples of 256");
    const size_t w_elems = (size_t)6 * d * d + (size_t)2 * d * dff, n_scales = (size_t)7 * d + dff;      // per layer
    // MXFP4W: d_w8 holds the nibbles (half a byte per element), d_wscale the e8m0 block scales (one byte per 32 elements)
    d_w8.reserve(fp4 ? Ld * w_elems / 2 : Ld * w_elems, stream); d_wscale.reserve(fp4 ? Ld * w_elems / 32 : Ld * n_scales * 4, stream); d_wdq.reserve(Ld * w_elems * 2, stream);
    dec8.resize(Ld);
    for (int i = 0; i < Ld; ++i) {
      DecLayer& L = dec[i];
      const void** slot[6] = {&L.wqkv, &L.wo, &L.wcq, &L.wco, &L.w1, &L.w2};
      const int Ns[6] = {3 * d, d, d, d, dff, d}, Ks[6] = {d, d, d, d, d, dff};
      unsigned char* w8 = d_w8.as<unsigned char>() + (fp4 ? i * w_elems / 2 : i * w_elems);
      bf16_t* dq = d_wdq.as<bf16_t>() + i * w_elems;
      float* sc = d_wscale.as<float>() + i * n_scales;
      unsigned char* sc4 = d_wscale.as<unsigned char>() + i * w_elems / 32;
      for (int j = 0; j < 6; ++j) {
        const size_t ne = (size_t)Ns[j] * Ks[j];
        if (fp4) launch_quantize_rows_mxfp4((const bf16_t*)*slot[j], Ks[j], Ns[j], Ks[j], w8, sc4, dq, stream);
        else launch_quantize_rows_fp8((const bf16_t*)*slot[j], Ks[j], Ns[j], Ks[j], w8, sc, dq, stream);
        dec8[i].w[j] = w8; dec8[i].s[j] = sc; dec8[i].s4[j] = sc4;
        *slot[j] = dq;                                     // from here on "the weights" are the dequantised copies
        w8 += fp4 ? ne / 2 : ne; dq += ne; sc += Ns[j]; sc4 += ne / 32;
      }
    }
  }
  if (precision == ASR_PRECISION_BF16 && use_decode_gemm) {      // column sums of the three LayerNorm-folded projections of every decoder layer: [3d | d | dff]
    const size_t per = (size_t)3 * d + d + dff;
    d_colsum.reserve((size_t)Ld * per * 4, stream);
    for (int i = 0; i < Ld; ++i) {
      float* c0 = d_colsum.as<float>() + i * per;
      launch_colsum_bf16((const bf16_t*)dec[i].wqkv, d, 3 * d, d, c0, stream);
      launch_colsum_bf16((const bf16_t*)dec[i].wcq, d, d, d, c0 + 3 * d, stream);
      launch_colsum_bf16((const bf16_t*)dec[i].w1, d, dff, d, c0 + 4 * d, stream);
    }
    HIP_CHECK(hipStreamSynchronize(stream));
  }
}

// ======================================================================================== encoder
template <typename T>
void WhSession::encode(const float* audio, int audio_mem, const int64_t* offs, int B, int32_t* n_pos_out) {
  const auto& c = cfg;
  ASR_REQUIRE(B > 0 && audio && offs, "whisper_encode: bad argument");
  HIP_CHECK(hipSetDevice(device));
  const int d = c.d_model, dff = c.d_ffn, Ld = c.n_dec_layers, H = c.n_heads;
  plan.assign(B, UttPlan{});
  std::vector<UttPlan> splan(B);                          // the conv stem's view: same utterances, row_off in the gapped layout's row space
  int r = 0, rg = 0, frames = 0, n_fb = 0, n_qb = 0, max_T = 0;
  const int64_t base0 = offs[0];
  for (int b = 0; b < B; ++b) {
    const int64_t n = offs[b + 1] - offs[b];
    ASR_REQUIRE(n >= c.nfft, "whisper: utterance %d has %lld samples (< n_fft %d)", b, (long long)n, c.nfft);
    ASR_REQUIRE(n <= c.max_audio_len, "whisper: utterance %d has %lld samples (> max_audio_len %d)", b, (long long)n, c.max_audio_len);
    UttPlan& p = plan[b];
    p.audio_off = offs[b] - base0;
    p.n_samples = (int)n;
    p.n_frames = (int)n / c.hop_length;                 // centred STFT with the last frame dropped (:96-103)
    p.frame_off = frames;
    p.T = (p.n_frames + 1) / 2;                          // conv2 stride 2, pad 1
    p.n_lfr = p.T;
    ASR_REQUIRE(p.T <= c.max_source_positions, "whisper: %d encoder positions exceed max_source_positions", p.T);
    p.row_off = r;
    p.lang = 0;
    p.blk0 = n_fb;
    frames += p.n_frames;
    splan[b] = p;
    splan[b].row_off = rg;
    r += round_up(p.T, 16);                              // encoder stream: compact, 16-row aligned (8 s: 400 rows per utterance)
    rg += round_up(p.T + 1, 16);                         // conv stem: +1 = room for the right zero-pad frame
    n_fb += (p.n_frames + 63) / 64;
    max_T = std::max(max_T, p.T);
    if (n_pos_out) n_pos_out[b] = p.T;
  }
  batch = B; rows = r; Mpad = round_up(r, 128); hist = 0; max_T_enc = max_T;
  const int rows_g = rg, Mg = round_up(rg, 128);
  int att_qt = 0, att_nw = 4, q_rows = 64;
  if (precision == ASR_PRECISION_BF16) { attention_geometry(max_T, c.d_head, &att_qt, &att_nw); q_rows = 16 * att_qt * att_nw; }
  for (int b = 0; b < B; ++b) n_qb += (plan[b].T + q_rows - 1) / q_rows;
  const int R = 2 * Mg;                                  // gapped (frame-rate) rows
  const int64_t total_samples = offs[B] - base0;

  // plan blob: [UttPlan B (encoder rows)][UttPlan B (stem rows)][blk_utt][blk_f0][qb_utt][qb_q0][row_utt Mpad][pos_rows Mg][grow_utt R]
  const size_t plan_bytes = 2 * sizeof(UttPlan) * B + 4 * (2 * (size_t)n_fb + 2 * (size_t)n_qb + (size_t)Mpad + (size_t)Mg + R);
  if (plan_bytes > h_plan_cap) {
    if (h_plan) HIP_CHECK(hipHostFree(h_plan));
    HIP_CHECK(hipHostMalloc(&h_plan, plan_bytes * 2, hipHostMallocDefault));
    h_plan_cap = plan_bytes * 2;
  }
  unsigned char* hp = (unsigned char*)h_plan;
  memcpy(hp, plan.data(), sizeof(UttPlan) * B);
  memcpy(hp + sizeof(UttPlan) * B, splan.data(), sizeof(UttPlan) * B);
  int32_t* blk_utt = (int32_t*)(hp + 2 * sizeof(UttPlan) * B);
  int32_t* blk_f0 = blk_utt + n_fb;
  int32_t* qb_utt = blk_f0 + n_fb;
  int32_t* qb_q0 = qb_utt + n_qb;
  int32_t* row_utt = qb_q0 + n_qb;
  int32_t* pos_rows = row_utt + Mpad;
  int32_t* grow_utt = pos_rows + Mg;
  for (int i = 0; i < Mpad; ++i) row_utt[i] = -1;
  for (int i = 0; i < Mg; ++i) pos_rows[i] = 0;
  for (int i = 0; i < R; ++i) grow_utt[i] = -1;
  {
    int fi = 0, qi = 0;
    for (int b = 0; b < B; ++b) {
      for (int f0 = 0; f0 < plan[b].n_frames; f0 += 64) { blk_utt[fi] = b; blk_f0[fi++] = f0; }
      for (int q0 = 0; q0 < plan[b].T; q0 += q_rows) { qb_utt[qi] = b; qb_q0[qi++] = q0; }
      const int rb = round_up(plan[b].T + 1, 16), rc = round_up(plan[b].T, 16);
      for (int t = 0; t < rc; ++t) row_utt[plan[b].row_off + t] = b;
      for (int t = 0; t < rb; ++t) {
        pos_rows[splan[b].row_off + t] = t < plan[b].T ? t : 0;
        grow_utt[2 * (splan[b].row_off + t)] = b;
        grow_utt[2 * (splan[b].row_off + t) + 1] = b;
      }
    }
  }
  { void* before = d_plan.ptr; d_plan.reserve(plan_bytes, stream); if (d_plan.ptr != before) ++ws_epoch; }
  HIP_CHECK(hipMemcpyAsync(d_plan.ptr, h_plan, plan_bytes, hipMemcpyHostToDevice, stream));
  const UttPlan* dp = d_plan.as<UttPlan>();
  const UttPlan* dps = dp + B;                            // stem rows
  const int32_t* d_blk_utt = (const int32_t*)((unsigned char*)d_plan.ptr + 2 * sizeof(UttPlan) * B);
  const int32_t* d_blk_f0 = d_blk_utt + n_fb;
  const int32_t* d_qb_utt = d_blk_f0 + n_fb;
  const int32_t* d_qb_q0 = d_qb_utt + n_qb;
  const int32_t* d_row_utt = d_qb_q0 + n_qb;
  const int32_t* d_pos_rows = d_row_utt + Mpad;
  const int32_t* d_grow_utt = d_pos_rows + Mg;

  const size_t eT = sizeof(T);
  const float* d_aud;
  if (audio_mem == ASR_MEM_HOST) {
    d_audio.reserve((size_t)total_samples * 4, stream);
    HIP_CHECK(hipMemcpyAsync(d_audio.ptr, audio + base0, (size_t)total_samples * 4, hipMemcpyHostToDevice, stream));
    d_aud = d_audio.as<float>();
  } else {
    d_aud = audio + base0;
  }
  const int Rpad = R + 256;                                        // tile-edge + halo rows of the strided conv views
  d_mel.reserve((size_t)frames * c.n_mels * 4, stream);
  d_blkmax.reserve((size_t)n_fb * 4, stream);
  d_x0.reserve((size_t)(Rpad + 1) * c.n_mels * eT, stream);
  d_h1.reserve((size_t)Rpad * d * eT, stream);
  d_xa.reserve((size_t)Mpad * d * 4, stream);
  d_xb.reserve((size_t)Mpad * d * 4, stream);
  d_h.reserve((size_t)Mpad * d * eT, stream);
  d_qk.reserve((size_t)Mpad * 2 * d * eT, stream);
  d_vt.reserve((size_t)Mpad * d * eT, stream);
  d_ctx.reserve((size_t)Mpad * d * eT, stream);
  d_ffn.reserve(std::max((size_t)Mpad * dff * eT, (size_t)Mg * d * 4), stream);
  if (fp8_mm) { d_h8.reserve((size_t)Mpad * d, stream); d_ffn8.reserve((size_t)Mpad * dff, stream); }
  { void* before = d_cross.ptr; d_cross.reserve((size_t)2 * Ld * H * Mpad * 64 * eT, stream); if (d_cross.ptr != before) ++ws_epoch; }

  // ---- STFT power -> mel -> log10 (STFT_Process.py:224-246, Export_Whisper.py:424-425)
  {
    ProfScope ps(prof, "logmel", stream);
    FbankArgs fa;
    fa.audio = d_aud; fa.plan = dp; fa.blk_utt = d_blk_utt; fa.blk_f0 = d_blk_f0; fa.dft_packed = dft; fa.mel_packed = melp;
    fa.mel_out = d_mel.as<float>(); fa.n_bin_tiles = n_bin_tiles; fa.n_kchunks = n_kchunks; fa.n_mel_tiles = c.n_mels / 16;
    fa.n_mels = c.n_mels; fa.win = c.nfft; fa.hop = c.hop_length; fa.log_floor = 1e-10f; fa.whisper = 1;
    fa.blk_max = d_blkmax.as<float>();
    launch_fbank(fa, n_fb, stream);
    // per-utterance max clamp + (x+4)/4, written behind one leading zero row (the conv view of row j starts at j-1)
    T* x0 = d_x0.as<T>();
    HIP_CHECK(hipMemsetAsync(x0, 0, (size_t)c.n_mels * eT, stream));
    launch_whisper_mel_finish<T>(d_mel.as<float>(), d_blkmax.as<float>(), dps, d_grow_utt, R, c.n_mels, x0 + c.n_mels, stream);
  }
  if (taps_enabled) save_tap("mel_gapped", d_x0.as<T>() + c.n_mels, R, c.n_mels, c.n_mels, (int)eT);
  // ---- conv stem as two GEMMs over strided views (no im2col): conv1 row j = frames j-1..j+1, conv2 row m = rows 2m..2m+2
  {
    ProfScope ps(prof, "conv_stem", stream);
    GemmArgs g1;
    g1.A = d_x0.ptr; g1.lda = c.n_mels; g1.W = conv1_w; g1.ldw = 3 * c.n_mels; g1.M = R; g1.N = d; g1.K = 3 * c.n_mels;
    g1.bias = conv1_b; g1.act = act; g1.out_lo = d_h1.ptr; g1.ld_out_lo = d;
    gemm(g1);
    launch_zero_gap_rows<T>(d_h1.as<T>(), d, d, dps, d_grow_utt, R, stream);     // conv2's zero padding
    // conv2's output rows live in the stem's row space (row m = gapped rows 2m .. 2m + 2); they land in the (still unused) FFN buffer and are
    // moved to the encoder's compact rows: every encoder GEMM then sees 16-row-aligned utterances without the pad row (8 s: 400, not 416)
    float* stem_out = d_ffn.as<float>();
    GemmArgs g2;
    g2.A = d_h1.ptr; g2.lda = 2 * d; g2.W = conv2_w; g2.ldw = 3 * d; g2.M = rows_g; g2.N = d; g2.K = 3 * d; g2.bias = conv2_b;
    g2.act = act; g2.add2 = enc_pos; g2.ld_add2 = d; g2.add2_rows = d_pos_rows; g2.out_f32 = stem_out; g2.ld_out_f32 = d;
    gemm(g2);
    launch_compact_rows(stem_out, dps, dp, d_row_utt, rows, d, d_xa.as<float>(), stream);
  }
  if (taps_enabled) save_tap("stem", d_xa.ptr, rows, d, d, 4);
  // ---- encoder layers (:430-437)
  float* xa = d_xa.as<float>();
  float* xb = d_xb.as<float>();
  T* h = d_h.as<T>();
  T* qk = d_qk.as<T>();
  T* vt = d_vt.as<T>();
  T* ctx = d_ctx.as<T>();
  T* ffn = d_ffn.as<T>();
  for (int i = 0; i < c.n_enc_layers; ++i) {
    const EncLayer& L = enc[i];
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(xa, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream); }
    {
      ProfScope ps(prof, "gemm_qkv", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = L.wqkv; g.ldw = d; g.M = rows; g.N = 2 * d; g.K = d; g.bias = L.bqkv; g.out_lo = qk; g.ld_out_lo = 2 * d;
      gemm(g);
      GemmArgs gv;
      gv.A = h; gv.lda = d; gv.W = (const T*)L.wqkv + (size_t)2 * d * d; gv.ldw = d; gv.M = rows; gv.N = d; gv.K = d;
      gv.bias = L.bqkv + 2 * d; gv.out_t = vt; gv.ld_out_t = Mpad;
      gemm(gv);
    }
    {
      ProfScope ps(prof, "attention", stream);
      AttnArgs aa;
      aa.q = qk; aa.k = qk + d; aa.ld_qk = 2 * d; aa.vt = vt; aa.ld_vt = Mpad; aa.ctx = ctx; aa.ld_ctx = d; aa.plan = dp;
      aa.qb_utt = d_qb_utt; aa.qb_q0 = d_qb_q0; aa.n_qblocks = n_qb; aa.n_heads = H; aa.qt = att_qt; aa.n_waves = att_nw; aa.max_T = max_T;
      if (precision == ASR_PRECISION_BF16) launch_attention_bf16_hd64(aa, stream);
      else launch_attention_f32(aa, c.d_head, stream);
    }
    {
      ProfScope ps(prof, "gemm_out", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = L.wo; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.bias = L.bo; g.add = xa; g.ld_add = d;
      g.out_f32 = xb; g.ld_out_f32 = d;
      gemm(g);
    }
    if (fp8_mm && sizeof(T) == 2) {                       // FFN pair on the FP8 matrix pipe: e4m3 operand rows, twice the bf16 MFMA rate
      { ProfScope ps(prof, "layernorm", stream); launch_layernorm_fp8(xb, d, rows, d, 1e-5f, 1.0f, d_h8.as<unsigned char>(), d, stream); }
      {
        ProfScope ps(prof, "gemm_ffn1", stream);
        Fp8GemmArgs g;
        g.A = d_h8.as<unsigned char>(); g.lda = d; g.W = enc8[i].w1; g.ldw = d; g.M = rows; g.N = dff; g.K = d; g.w_scale = enc8[i].s1; g.bias = L.b1;
        g.act = act; g.out8 = d_ffn8.as<unsigned char>(); g.ld_out8 = dff;
        g.out_inv_scale = ldexpf(1.0f, -fp8_act_shift); g.sat_count = d_sat.as<unsigned long long>();
        launch_gemm_fp8(g, stream);
      }
      {
        ProfScope ps(prof, "gemm_ffn2", stream);
        Fp8GemmArgs g;
        g.A = d_ffn8.as<unsigned char>(); g.lda = dff; g.W = enc8[i].w2; g.ldw = dff; g.M = rows; g.N = d; g.K = dff; g.w_scale = enc8[i].s2; g.bias = L.b2;
        g.a_scale = ldexpf(1.0f, fp8_act_shift);
        g.add = xb; g.ld_add = d; g.out_f32 = xa; g.ld_out_f32 = d;
        launch_gemm_fp8(g, stream);
      }
      continue;
    }
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(xb, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream); }
    {
      ProfScope ps(prof, "gemm_ffn1", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = L.w1; g.ldw = d; g.M = rows; g.N = dff; g.K = d; g.bias = L.b1; g.act = act; g.out_lo = ffn; g.ld_out_lo = dff;
      gemm(g);
    }
    {
      ProfScope ps(prof, "gemm_ffn2", stream);
      GemmArgs g;
      g.A = ffn; g.lda = dff; g.W = L.w2; g.ldw = dff; g.M = rows; g.N = d; g.K = dff; g.bias = L.b2; g.add = xb; g.ld_add = d;
      g.out_f32 = xa; g.ld_out_f32 = d;
      gemm(g);
    }
  }
  // ---- final LayerNorm + fused cross-KV projection into [kv][layer][head] slabs of [row][64] (:438-447)
  if (taps_enabled) {
    launch_layernorm<float>(xa, d, rows, d, enc_ln_g, enc_ln_b, 1e-5f, xb, d, d, stream);
    save_tap("enc_out", xb, rows, d, d, 4);
  }
  { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(xa, d, rows, d, enc_ln_g, enc_ln_b, 1e-5f, h, d, d, stream); }
  {
    ProfScope ps(prof, "gemm_crosskv", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = ckv_w; g.ldw = d; g.M = rows; g.N = 2 * Ld * d; g.K = d; g.bias = ckv_b;
    g.out_lo = d_cross.ptr; g.lo_group = 64; g.ld_out_lo = Mpad * 64;
    gemm(g);
  }
  if constexpr (sizeof(T) == 2) {
    if (fp8 && fp8_kv) {
      ProfScope ps(prof, "quant_crosskv", stream);
      { void* before = d_cross8.ptr; void* before_s = d_cscale.ptr;      // either buffer moving invalidates the captured decode graph (the scales grow with B, the bytes with the rows)
        d_cross8.reserve((size_t)2 * Ld * H * Mpad * 64, stream); d_cscale.reserve((size_t)2 * Ld * H * B * 4, stream);
        if (d_cross8.ptr != before || d_cscale.ptr != before_s) ++ws_epoch; }
      launch_quantize_crosskv_fp8(d_cross.as<bf16_t>(), (size_t)Mpad * 64, 2 * Ld * H, d_plan.as<UttPlan>(), B, d_cross8.as<unsigned char>(), d_cscale.as<float>(),
                                  fp8_fake ? 1 : 0, stream);
    }
  }
  if (taps_enabled) save_tap("cross", d_cross.ptr, (int64_t)2 * Ld * H * Mpad, 64, 64, (int)eT);
  HIP_CHECK(hipStreamSynchronize(stream));
  if (prof.enabled) prof.collect();
}

// ======================================================================================== decoder step
// All launches of one step. `hist_dev` (device-resident history length) is what the kernels read, so the single-token
// step is position independent and ONE captured hipGraph replays for every decode position.
template <typename T>
void WhSession::enqueue_step(const int32_t* ids_dev, int n, bool is_prefill, bool use_hist_dev) {
  const auto& c = cfg;
  const int B = batch, d = c.d_model, dff = c.d_ffn, Ld = c.n_dec_layers, H = c.n_heads;
  const int R = B * n, Rp = round_up(R, 128);
  const int32_t* hd = use_hist_dev ? d_hist.as<int32_t>() : nullptr;
  float* xa = d_dx.as<float>();
  float* xb = xa + (size_t)Rp * d;
  float* xc = xb + (size_t)Rp * d;
  T* qkv = d_dqkv.as<T>();
  T* hh = qkv + (size_t)Rp * 3 * d;
  T* ctx = hh + (size_t)Rp * d;
  T* ffn = ctx + (size_t)Rp * d;
  T* cq = ffn + (size_t)Rp * dff;
  T* hl = cq + (size_t)Rp * d;
  const UttPlan* dp = d_plan.as<UttPlan>();
  const size_t cache_l = (size_t)B * H * c.max_target_positions * 64;
  // bf16 mode, M <= 64 rows: the (affine-less) LayerNorm runs inside the skinny GEMM's prologue
  const bool fuse_ln = precision == ASR_PRECISION_BF16 && R <= 32 && d % 256 == 0;   // above 32 rows a separate LayerNorm launch is cheaper
  auto ln_gemm = [&](const float* x, GemmArgs& g) {
    // the prologue's registers / LDS hold one workgroup per CU: outputs wider than the chip (N / 16 > 256 workgroups, fc1) are faster
    // behind a separate LayerNorm launch, with two plain workgroups sharing a CU
    if (fuse_ln && (R <= 16 || g.N / 16 <= 256)) { g.A = nullptr; g.ln_x = x; g.ld_ln_x = d; }
    else { ProfScope ps(prof, "dec_layernorm", stream); launch_layernorm<T>(x, d, R, d, nullptr, nullptr, 1e-5f, hh, d, d, stream); g.A = hh; g.lda = d; }
    ProfScope ps(prof, "dec_gemm", stream);
    gemm(g);
  };
  // bf16 decode steps of <= 64 rows: the decode GEMM (csrc/decode_gemm.hip) with the three LayerNorms folded into q|k|v, cross-q and fc1
  bool dgm = false;
  bf16_t *xa_lo = nullptr, *xb_lo = nullptr, *xc_lo = nullptr;
  const float* csum = d_colsum.as<float>();
  const size_t cs_l = (size_t)3 * d + d + dff;
  if constexpr (sizeof(T) == 2) {
    dgm = use_decode_gemm && R <= 64 && d_colsum.ptr != nullptr && d % 256 == 0 && dff % 256 == 0;
    xa_lo = d_dlo.as<bf16_t>(); xb_lo = xa_lo + (size_t)Rp * d; xc_lo = xb_lo + (size_t)Rp * d;
  }
  const bool w8 = fp8 && fp8_weights && !fp8_fake;
  // (Round 5 also cut the batch into sub-batches whose layer loops ran on one stream each ("decode chains"): 3.29 vs 3.31 ms per token at 64 sequences,
  //  slower at 32 -- profiles/r05_whisper_decode_chains.txt; removed in round 6.)
  if (!d_skws.ptr) { d_skws.reserve(SK_WS_BYTES, stream); d_skcnt.reserve((size_t)SK_CNT * 4, stream); }
  const int plan_rows = 0, NC = 1;
  auto dg = [&](hipStream_t st, int ci, int r0, int Rc, int layer, const void* A, int lda, const void* Wt, int wi, int N, int K, const float* bias, const float* colsum,
                const float* add, int act_, float* of32, void* olo, int ld_lo) {
    ProfScope ps(prof, "dec_gemm", st);
    DecGemmArgs g;
    g.A = (const bf16_t*)A + (size_t)r0 * lda; g.lda = lda; g.W = (const bf16_t*)Wt; g.ldw = K; g.M = Rc; g.plan_M = plan_rows; g.N = N; g.K = K; g.bias = bias; g.colsum = colsum;
    if (w8 && fp4) { g.W = nullptr; g.W4 = dec8[layer].w[wi]; g.w_scale4 = dec8[layer].s4[wi]; }
    else if (w8) { g.W = nullptr; g.W8 = dec8[layer].w[wi]; g.w_scale = dec8[layer].s[wi]; }
    g.add = add ? add + (size_t)r0 * d : nullptr; g.ld_add = d; g.act = act_; g.out_f32 = of32 ? of32 + (size_t)r0 * d : nullptr; g.ld_out_f32 = d;
    g.out_lo = olo ? (bf16_t*)olo + (size_t)r0 * ld_lo : nullptr; g.ld_out_lo = ld_lo;
    g.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(d_skws.ptr) + (size_t)ci * SK_WS_BYTES); g.ws_bytes = SK_WS_BYTES;
    g.cnt = d_skcnt.as<int32_t>() + (size_t)ci * SK_CNT;
    launch_decode_gemm(g, st);
  };
  // embedding + layer loop of sequences [b0, b0 + nb) on stream `st` (chain ci)
  auto run_chain = [&](hipStream_t st, int ci, int b0, int nb) {
    const int r0 = b0 * n, Rc = nb * n;
    { ProfScope ps(prof, "dec_embed", st);
      launch_embed_pos<T>(ids_dev + r0, Rc, n, hist, hd, (const T*)embed, dec_pos, d, xa + (size_t)r0 * d, st);
      if (dgm) launch_rows_to_bf16(xa + (size_t)r0 * d, xa_lo + (size_t)r0 * d, (size_t)(NC == 1 ? Rp : Rc) * d, st); }
    for (int l = 0; l < Ld; ++l) {
      const DecLayer& L = dec[l];
      if (dgm) dg(st, ci, r0, Rc, l, xa_lo, d, L.wqkv, 0, 3 * d, d, L.bqkv, csum + l * cs_l, nullptr, ACT_NONE, nullptr, qkv, 3 * d);
      else {
        GemmArgs g;
        g.W = L.wqkv; g.ldw = d; g.M = R; g.N = 3 * d; g.K = d; g.bias = L.bqkv; g.out_lo = qkv; g.ld_out_lo = 3 * d;
        ln_gemm(xa, g);
      }
      {
        ProfScope ps(prof, "dec_self_attn", st);
        DecAttnArgs a;
        a.q = qkv; a.ld_q = 3 * d; a.q_col0 = 0; a.kv_new = qkv; a.ld_new = 3 * d; a.k_col0 = d; a.v_col0 = 2 * d;
        if (kv_paged) {
          a.k_base = d_kvpool.as<T>() + (size_t)(l * 2) * H * KV_PAGE * 64; a.v_base = d_kvpool.as<T>() + (size_t)(l * 2 + 1) * H * KV_PAGE * 64;
          a.page_table = d_ptable.as<int32_t>(); a.pages_per_seq = (c.max_target_positions + KV_PAGE - 1) / KV_PAGE;
          a.page_stride = (int64_t)Ld * 2 * H * KV_PAGE * 64;
        } else {
          a.k_base = d_kc.as<T>() + l * cache_l; a.v_base = d_vc.as<T>() + l * cache_l;
          a.stride_b = (int64_t)H * c.max_target_positions * 64; a.stride_h = (int64_t)c.max_target_positions * 64;
        }
        a.plan = nullptr; a.hist = hist; a.hist_dev = hd; a.n = n; a.n_heads = H; a.causal = 1; a.out = ctx; a.ld_out = d;
        a.max_keys = c.max_target_positions;
        a.b0 = b0;
        launch_decode_attention<T>(a, nb, st);
      }
      if (dgm) {
        dg(st, ci, r0, Rc, l, ctx, d, L.wo, 1, d, d, L.bo, nullptr, xa, ACT_NONE, xb, xb_lo, d);
        dg(st, ci, r0, Rc, l, xb_lo, d, L.wcq, 2, d, d, L.bcq, csum + l * cs_l + 3 * d, nullptr, ACT_NONE, nullptr, cq, d);
      } else {
        {
          ProfScope ps(prof, "dec_gemm", stream);
          GemmArgs g;
          g.A = ctx; g.lda = d; g.W = L.wo; g.ldw = d; g.M = R; g.N = d; g.K = d; g.bias = L.bo; g.add = xa; g.ld_add = d; g.out_f32 = xb; g.ld_out_f32 = d;
          gemm(g);
        }
        GemmArgs g;
        g.W = L.wcq; g.ldw = d; g.M = R; g.N = d; g.K = d; g.bias = L.bcq; g.out_lo = cq; g.ld_out_lo = d;
        ln_gemm(xb, g);
      }
      {
        ProfScope ps(prof, "dec_cross_attn", st);
        DecAttnArgs a;
        a.q = cq; a.ld_q = d; a.q_col0 = 0; a.kv_new = nullptr; a.ld_new = 0; a.k_col0 = a.v_col0 = 0;
        a.k_base = d_cross.as<T>() + (size_t)(0 * Ld + l) * H * Mpad * 64;
        a.v_base = d_cross.as<T>() + (size_t)(1 * Ld + l) * H * Mpad * 64;
        a.stride_b = 0; a.stride_h = (int64_t)Mpad * 64; a.plan = dp; a.hist = 0; a.hist_dev = nullptr; a.n = n; a.n_heads = H; a.causal = 0;
        a.max_keys = max_T_enc;
        a.out = ctx; a.ld_out = d;
        if (fp8 && fp8_kv && !fp8_fake) {
          a.k_base = d_cross8.as<unsigned char>() + (size_t)(0 * Ld + l) * H * Mpad * 64;
          a.v_base = d_cross8.as<unsigned char>() + (size_t)(1 * Ld + l) * H * Mpad * 64;
          a.k_scale = d_cscale.as<float>() + (size_t)(0 * Ld + l) * H * B; a.v_scale = d_cscale.as<float>() + (size_t)(1 * Ld + l) * H * B;
        }
        a.b0 = b0; a.scale_ld = B;
        launch_decode_attention<T>(a, nb, st);
      }
      if (dgm) {
        dg(st, ci, r0, Rc, l, ctx, d, L.wco, 3, d, d, L.bco, nullptr, xb, ACT_NONE, xc, xc_lo, d);
        dg(st, ci, r0, Rc, l, xc_lo, d, L.w1, 4, dff, d, L.b1, csum + l * cs_l + 4 * d, nullptr, act, nullptr, ffn, dff);
        if (Rc <= 32 || w8) dg(st, ci, r0, Rc, l, ffn, dff, L.w2, 5, d, dff, L.b2, nullptr, xc, ACT_NONE, xa, xa_lo, d);
        else {               // 33..64 rows: the tiled split-K pass shares the activation rows across 64 columns (13.8 vs 17.9 us); it writes the bf16 copy too (one chain only: Rc = R)
          ProfScope ps(prof, "dec_gemm", stream);
          GemmArgs g2;
          g2.A = ffn; g2.lda = dff; g2.W = L.w2; g2.ldw = dff; g2.M = R; g2.N = d; g2.K = dff; g2.bias = L.b2; g2.add = xc; g2.ld_add = d;
          g2.out_f32 = xa; g2.ld_out_f32 = d; g2.out_lo = xa_lo; g2.ld_out_lo = d;
          gemm(g2);
        }
      } else {
        {
          ProfScope ps(prof, "dec_gemm", stream);
          GemmArgs g;
          g.A = ctx; g.lda = d; g.W = L.wco; g.ldw = d; g.M = R; g.N = d; g.K = d; g.bias = L.bco; g.add = xb; g.ld_add = d; g.out_f32 = xc; g.ld_out_f32 = d;
          gemm(g);
        }
        GemmArgs g;
        g.W = L.w1; g.ldw = d; g.M = R; g.N = dff; g.K = d; g.bias = L.b1; g.act = act; g.out_lo = ffn; g.ld_out_lo = dff;
        ln_gemm(xc, g);
        ProfScope ps(prof, "dec_gemm", stream);
        GemmArgs g2;
        g2.A = ffn; g2.lda = dff; g2.W = L.w2; g2.ldw = dff; g2.M = R; g2.N = d; g2.K = dff; g2.bias = L.b2; g2.add = xc; g2.ld_add = d;
        g2.out_f32 = xa; g2.ld_out_f32 = d;
        gemm(g2);
      }
    }
  };
  run_chain(stream, 0, 0, B);
  // final LayerNorm of the LAST position of every sequence, tied proj_out, -128 suppress penalty (:663-666)
  {
    ProfScope ps(prof, "dec_logits", stream);
    GemmArgs g;
    g.W = embed; g.ldw = d; g.M = B; g.N = vpad; g.K = d; g.bias = suppress; g.out_f32 = d_logits.as<float>(); g.ld_out_f32 = vpad;
    // up to 16 sequences: LayerNorm inside the weight-streaming GEMM; above, a separate LayerNorm feeds the 128 x 128 tiles (every 16-column
    // granule of the streaming kernel would re-read all B activation rows: 170 us for 32 x 51 866 x 1280 against ~45)
    if (precision == ASR_PRECISION_BF16 && B <= 16 && d % 256 == 0) {
      g.ln_x = xa + (size_t)(n - 1) * d; g.ld_ln_x = n * d; g.ln_gamma = dec_ln_g; g.ln_beta = dec_ln_b;
    } else {
      launch_layernorm<T>(xa + (size_t)(n - 1) * d, n * d, B, d, dec_ln_g, dec_ln_b, 1e-5f, hl, d, d, stream);
      g.A = hl; g.lda = d;
    }
    gemm(g);
    const bool penalised = penalty_value != 1.0f && !sampling;
    if (penalised && !is_prefill)          // APPLY_PENALTY over the saved ids; the history is empty at the prefill (:312-325)
      launch_apply_penalty(d_logits.as<float>(), vpad, B, d_save.as<int32_t>(), c.max_target_positions, d_nsaved.as<int32_t>(),
                           penalty_range, penalty_value, stream);
    if (sampling) {                        // TOPK_TOPP_SAMPLING (:263-308): BEGIN_SUPPRESS bias first, history = every sampled id
      SampleArgs sa;
      sa.logits = d_logits.as<float>(); sa.ld = vpad; sa.rows = B; sa.n_valid = c.vocab; sa.extra = is_prefill ? begin : nullptr;
      sa.save_ids = d_save.as<int32_t>(); sa.ld_save = c.max_target_positions; sa.n_saved = d_nsaved.as<int32_t>();
      sa.temperature = temperature; sa.top_p = top_p; sa.repetition_penalty = samp_rep_penalty; sa.top_k = top_k;
      sa.noise = noise_armed ? d_noise.as<float>() : nullptr; sa.seed = samp_seed; sa.next = d_next.as<int32_t>();
      launch_sample_topk_topp(sa, stream);
    } else {
      // BEGIN_SUPPRESS (-inf on begin_suppress_tokens) applies to the head after a prefill only (:228-240)
      launch_argmax_rows(d_logits.as<float>(), vpad, B, c.vocab, is_prefill ? begin : nullptr, d_next.as<int32_t>(), stream);
    }
    if (penalised || sampling || track_history) {   // GREEDY_SEARCH / the sampling head append their pick to the history (:243-251,306)
      launch_append_ids(d_next.as<int32_t>(), B, d_save.as<int32_t>(), c.max_target_positions, d_nsaved.as<int32_t>(), stream);
      launch_add_scalar(d_nsaved.as<int32_t>(), 1, stream);
    }
    launch_add_scalar(d_hist.as<int32_t>(), n, stream);
  }
}

// Block table + pool of the paged self-KV cache: make sure every sequence of the batch owns pages for `positions` positions. Growth appends whole
// generations (page ids j * B + slot, slot = b or -- tests -- a permutation of the batch), copies the old pool (a prefix of the new one), rewrites the table and
// invalidates the captured decode graph; with the first cut at prompt + 48 positions and doubling, a 448-position generation regrows at most three times.
void WhSession::ensure_kv_pages(int B, int positions, size_t elem_bytes) {
  const auto& c = cfg;
  const int P = (c.max_target_positions + KV_PAGE - 1) / KV_PAGE;
  const int need = std::min(P, (positions + KV_PAGE - 1) / KV_PAGE);
  if (B != kv_batch) { kv_gens = 0; kv_batch = B; }                       // another batch: the cache restarts with its prefill
  if (need <= kv_gens && d_kvpool.ptr && d_ptable.ptr) return;
  const int gens = std::min(P, std::max(need, kv_gens ? 2 * kv_gens : (positions + 48 + KV_PAGE - 1) / KV_PAGE));
  const size_t page_bytes = (size_t)c.n_dec_layers * 2 * c.n_heads * KV_PAGE * 64 * elem_bytes;
  const size_t bytes = (size_t)gens * B * page_bytes, old_bytes = std::min(d_kvpool.cap, (size_t)kv_gens * B * page_bytes);
  if (bytes > d_kvpool.cap) {
    DeviceBuffer fresh;
    fresh.reserve(bytes, stream);
    try {
      if (hist > 0 && old_bytes) HIP_CHECK(hipMemcpyAsync(fresh.ptr, d_kvpool.ptr, old_bytes, hipMemcpyDeviceToDevice, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    } catch (...) {
      fresh.release();                   // (DeviceBuffer has no destructor: a failed copy must not leak the new pool)
      throw;
    }
    d_kvpool.release();
    d_kvpool = fresh;
  }
  d_ptable.reserve((size_t)B * P * 4, stream);
  int32_t* tab = (int32_t*)pinned((size_t)B * P * 4 + 64);
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < P; ++j) tab[(size_t)b * P + j] = j < gens ? j * B + (kv_shuffle ? (B - 1 - b + j) % B : b) : -1;
  HIP_CHECK(hipMemcpyAsync(d_ptable.ptr, tab, (size_t)B * P * 4, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipStreamSynchronize(stream));                                 // (the pinned staging buffer is reused by the step that follows)
  kv_gens = gens;
  ++ws_epoch;
}

template <typename T>
void WhSession::step(const int32_t* ids_host, int n, bool is_prefill, int32_t* next_out, float* logits_out) {
  const auto& c = cfg;
  ASR_REQUIRE(batch > 0, "whisper: encode a batch before prefill / decode");
  ASR_REQUIRE(n >= 1 && n <= 8, "whisper: %d tokens per step (1..8)", n);
  if (is_prefill) hist = 0;                      // the reference always prefills with an empty self-KV (:476-480)
  ASR_REQUIRE(hist + n <= c.max_target_positions, "whisper: %d positions exceed max_target_positions %d", hist + n, c.max_target_positions);
  HIP_CHECK(hipSetDevice(device));
  const int B = batch, d = c.d_model, dff = c.d_ffn, Ld = c.n_dec_layers, H = c.n_heads;
  const int R = B * n, Rp = round_up(R, 128), Bp = round_up(B, 128);
  const size_t eT = sizeof(T);
  const size_t cache_elems = (size_t)Ld * B * H * c.max_target_positions * 64;
  auto grow = [&](DeviceBuffer& buf, size_t bytes) { void* before = buf.ptr; buf.reserve(bytes, stream); if (buf.ptr != before) ++ws_epoch; };
  if (kv_paged) ensure_kv_pages(B, hist + n, eT);
  else { grow(d_kc, cache_elems * eT); grow(d_vc, cache_elems * eT); }
  grow(d_ids, (size_t)B * 8 * 4);
  grow(d_next, (size_t)B * 4);
  grow(d_hist, 256);
  grow(d_nsaved, 256);
  grow(d_save, (size_t)B * c.max_target_positions * 4);
  grow(d_logits, (size_t)Bp * vpad * 4);
  grow(d_dx, (size_t)3 * Rp * d * 4);                  // three f32 residual-stream buffers
  if (precision == ASR_PRECISION_BF16) grow(d_dlo, (size_t)3 * Rp * d * 2);     // ... and their bf16 copies (operands of the LayerNorm-folded projections)
  grow(d_dqkv, (size_t)Rp * (3 * d + d + d + dff + d) * eT + (size_t)Bp * d * eT);
  const int32_t* ids_dev;
  if (ids_host) {
    int32_t* stage = (int32_t*)pinned((size_t)R * 4 + 64);
    for (int i = 0; i < R; ++i) {
      ASR_REQUIRE(ids_host[i] >= 0 && ids_host[i] < c.vocab, "whisper: token id %d out of range", ids_host[i]);
      stage[i] = ids_host[i];
    }
    HIP_CHECK(hipMemcpyAsync(d_ids.ptr, stage, (size_t)R * 4, hipMemcpyHostToDevice, stream));
    ids_dev = d_ids.as<int32_t>();
  } else {
    ASR_REQUIRE(n == 1, "whisper: device-resident ids feed single-token decode steps only");
    ids_dev = d_next.as<int32_t>();
  }
  if (is_prefill) {
    HIP_CHECK(hipMemsetAsync(d_hist.ptr, 0, 4, stream));
    HIP_CHECK(hipMemsetAsync(d_nsaved.ptr, 0, 4, stream));
  }
  // single-token steps fed from the device are position independent => one graph for all of them
  const bool graphable = use_graph && !ids_host && n == 1 && !taps_enabled && !prof.enabled && !noise_armed;
  const uint64_t key = ((uint64_t)B << 32) ^ (uint64_t)Mpad ^ (ws_epoch << 48) ^ (uint64_t)(uintptr_t)stream;
  auto capture = [&](hipStream_t cs) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    try {
      enqueue_step<T>(ids_dev, 1, false, true);
    } catch (...) {
      (void)hipStreamEndCapture(cs, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    HIP_CHECK(hipStreamEndCapture(cs, &graph));
    HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    return exec;
  };
  if (graphable && dec_graph && key == dec_key) {
    HIP_CHECK(hipGraphLaunch(dec_graph, stream));
  } else if (graphable && key == dec_eager_key) {
    drop_graphs();
    dec_graph = capture(stream);
    dec_key = key;
    HIP_CHECK(hipGraphLaunch(dec_graph, stream));
  } else {
    enqueue_step<T>(ids_dev, n, is_prefill, true);
    if (graphable) dec_eager_key = key;
  }
  hist += n;
  noise_armed = false;                   // caller-supplied uniforms serve exactly one step
  if (taps_enabled) save_tap("logits", d_logits.ptr, B, c.vocab, vpad, 4);
  if (next_out || logits_out) {
    unsigned char* st = (unsigned char*)pinned((size_t)B * 4 + (logits_out ? (size_t)B * c.vocab * 4 : 0));
    if (next_out) HIP_CHECK(hipMemcpyAsync(st, d_next.ptr, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
    if (logits_out)
      HIP_CHECK(hipMemcpy2DAsync(st + (size_t)B * 4, (size_t)c.vocab * 4, d_logits.ptr, (size_t)vpad * 4, (size_t)c.vocab * 4, B,
                                 hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (next_out) memcpy(next_out, st, (size_t)B * 4);
    if (logits_out) memcpy(logits_out, st + (size_t)B * 4, (size_t)B * c.vocab * 4);
    if (prof.enabled) prof.collect();
  }
}

}  // namespace

extern "C" int asr_whisper_create(const asr_whisper_config* cfg, const void* arena, size_t arena_bytes, int arena_mem,
                                  int device_id, int precision, asr_session** out) {
  return asr_guard([&] {
    ASR_REQUIRE(cfg && arena && out, "whisper_create: null argument");
    ASR_REQUIRE(precision == ASR_PRECISION_BF16 || precision == ASR_PRECISION_F32 || precision == ASR_PRECISION_FP8W || precision == ASR_PRECISION_FP8MM ||
                precision == ASR_PRECISION_MXFP4W, "whisper_create: bad precision %d", precision);
    asr_require_device(device_id);
    WhSession* s = new WhSession();
    try {
      s->kind = 2;
      s->device = device_id;
      asr_tenant_attach(s);
      s->fp4 = precision == ASR_PRECISION_MXFP4W;
      s->fp8 = precision == ASR_PRECISION_FP8W || precision == ASR_PRECISION_FP8MM || s->fp4;
      s->fp8_mm = precision == ASR_PRECISION_FP8MM;
      s->precision = s->fp8 ? ASR_PRECISION_BF16 : precision;        // FP8 mode = bf16 mode with byte-wide decoder weights and cross-K/V
      s->cfg = *cfg;
      gemm_reload_env();
      if (const char* e = getenv("ASR_FP8_FAKE")) s->fp8_fake = e[0] == '1';
      if (const char* e = getenv("ASR_FP8MM_ACT_SHIFT")) s->fp8_act_shift = std::min(std::max(atoi(e), 0), 16);
      if (const char* e = getenv("ASR_FP8_WEIGHTS")) s->fp8_weights = !(e[0] == '0');
      if (const char* e = getenv("ASR_FP8_KV")) s->fp8_kv = !(e[0] == '0');
      if (const char* e = getenv("ASR_NO_GRAPH")) s->use_graph = !(e[0] == '1');
      if (const char* e = getenv("ASR_KV_PAGED")) s->kv_paged = !(e[0] == '0');
      if (const char* e = getenv("ASR_KV_PAGE_SHUFFLE")) s->kv_shuffle = e[0] == '1';
      if (const char* e = getenv("ASR_DECODE_GEMM")) s->use_decode_gemm = !(e[0] == '0');
      HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
      s->own_stream = true;
      s->arena.load(arena, arena_bytes, arena_mem, s->stream);
      s->init();
    } catch (...) {
      delete s;
      throw;
    }
    *out = s;
  });
}

extern "C" int asr_whisper_encode(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                                  int32_t* n_positions_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_encode: not a Whisper session");
    TenantScope tenant(s);
    WhSession* w = static_cast<WhSession*>(s);
    if (w->precision == ASR_PRECISION_BF16) w->encode<bf16_t>(audio, audio_mem, audio_offsets, batch, n_positions_out);
    else w->encode<float>(audio, audio_mem, audio_offsets, batch, n_positions_out);
  });
}

extern "C" int asr_whisper_prefill(asr_session* s, const int32_t* ids, int n, int32_t* next_ids_out, float* logits_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2 && ids, "whisper_prefill: bad argument");
    TenantScope tenant(s);
    WhSession* w = static_cast<WhSession*>(s);
    if (w->precision == ASR_PRECISION_BF16) w->step<bf16_t>(ids, n, true, next_ids_out, logits_out);
    else w->step<float>(ids, n, true, next_ids_out, logits_out);
  });
}

extern "C" int asr_whisper_decode(asr_session* s, const int32_t* ids, int32_t* next_ids_out, float* logits_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_decode: not a Whisper session");
    TenantScope tenant(s);
    WhSession* w = static_cast<WhSession*>(s);
    if (w->precision == ASR_PRECISION_BF16) w->step<bf16_t>(ids, 1, false, next_ids_out, logits_out);
    else w->step<float>(ids, 1, false, next_ids_out, logits_out);
  });
}

extern "C" int asr_whisper_set_penalty(asr_session* s, float repeat_penalty, int penalty_range) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_set_penalty: not a Whisper session");
    ASR_REQUIRE(repeat_penalty > 0.0f && penalty_range >= 1 && penalty_range <= 64, "whisper_set_penalty: value %g range %d", repeat_penalty, penalty_range);
    WhSession* w = static_cast<WhSession*>(s);
    if (w->penalty_value != repeat_penalty || w->penalty_range != penalty_range) {
      w->penalty_value = repeat_penalty;
      w->penalty_range = penalty_range;
      ++w->ws_epoch;                       // the captured decode graph bakes the head in: re-capture
    }
  });
}

extern "C" int asr_whisper_no_speech_prob(asr_session* s, int no_speech_id, float* prob_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2 && prob_out, "whisper_no_speech_prob: bad argument");
    WhSession* w = static_cast<WhSession*>(s);
    ASR_REQUIRE(w->batch > 0 && w->hist > 0 && w->d_logits.ptr, "whisper_no_speech_prob: run a prefill first (the probe's logits are the input)");
    HIP_CHECK(hipSetDevice(w->device));
    const int B = w->batch;
    w->d_nsp.reserve((size_t)std::max(B, 64) * 4, w->stream);
    launch_no_speech_prob(w->d_logits.as<float>(), w->vpad, B, w->cfg.vocab, w->suppress, no_speech_id, w->d_nsp.as<float>(), w->stream);
    HIP_CHECK(hipMemcpyAsync(prob_out, w->d_nsp.ptr, (size_t)B * 4, hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
  });
}

extern "C" int asr_whisper_set_fp8_act_shift(asr_session* s, int shift) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_set_fp8_act_shift: not a Whisper session");
    ASR_REQUIRE(shift >= 0 && shift <= 16, "whisper_set_fp8_act_shift: shift %d outside 0 .. 16", shift);
    static_cast<WhSession*>(s)->fp8_act_shift = shift;          // (the encoder is not graph-captured: the next encode uses it)
  });
}

extern "C" int asr_whisper_fp8_stats(asr_session* s, uint64_t stats[2]) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2 && stats, "whisper_fp8_stats: not a Whisper session");
    WhSession* w = static_cast<WhSession*>(s);
    HIP_CHECK(hipSetDevice(w->device));
    stats[0] = 0; stats[1] = (uint64_t)w->fp8_act_shift;
    if (w->fp8_mm && w->d_sat.ptr) {
      unsigned long long n = 0;
      HIP_CHECK(hipStreamSynchronize(w->stream));
      HIP_CHECK(hipMemcpy(&n, w->d_sat.ptr, 8, hipMemcpyDeviceToHost));
      stats[0] = n;
    }
  });
}

extern "C" int asr_whisper_track_history(asr_session* s, int enable) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_track_history: not a Whisper session");
    WhSession* w = static_cast<WhSession*>(s);
    if (w->track_history != (enable != 0)) {
      w->track_history = enable != 0;
      ++w->ws_epoch;                       // the captured decode graph bakes the head in: re-capture
    }
  });
}

extern "C" int asr_whisper_set_sampling(asr_session* s, int enable, float temperature, int top_k, float top_p,
                                        float repetition_penalty, uint64_t seed) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2, "whisper_set_sampling: not a Whisper session");
    WhSession* w = static_cast<WhSession*>(s);
    if (enable) {
      ASR_REQUIRE(temperature > 0.0f && top_k >= 1 && top_k <= 64 && top_p > 0.0f && repetition_penalty > 0.0f && w->cfg.max_target_positions <= 512,
                  "whisper_set_sampling: temperature %g top_k %d top_p %g penalty %g", temperature, top_k, top_p, repetition_penalty);
      w->temperature = temperature; w->top_k = top_k; w->top_p = top_p; w->samp_rep_penalty = repetition_penalty; w->samp_seed = seed;
    }
    w->sampling = enable != 0;
    w->noise_armed = false;
    ++w->ws_epoch;                         // the captured decode graph bakes the head in: re-capture
  });
}

extern "C" int asr_whisper_set_sampling_noise(asr_session* s, const float* uniforms, int count) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2 && uniforms, "whisper_set_sampling_noise: bad argument");
    WhSession* w = static_cast<WhSession*>(s);
    ASR_REQUIRE(w->sampling && w->batch > 0 && count == w->batch * w->top_k, "whisper_set_sampling_noise: expects batch x top_k = %d uniforms for the next step",
                w->batch * w->top_k);
    HIP_CHECK(hipSetDevice(w->device));
    w->d_noise.reserve((size_t)count * 4, w->stream);
    HIP_CHECK(hipMemcpyAsync(w->d_noise.ptr, uniforms, (size_t)count * 4, hipMemcpyHostToDevice, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    w->noise_armed = true;
  });
}

extern "C" int asr_whisper_generate(asr_session* s, int max_new, int eos_id, int32_t* tokens_out, int32_t* n_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 2 && tokens_out && n_out && max_new >= 1, "whisper_generate: bad argument");
    TenantScope tenant(s);
    WhSession* w = static_cast<WhSession*>(s);
    ASR_REQUIRE(w->hist > 0, "whisper_generate: prefill first");
    const int B = w->batch;
    std::vector<int32_t> cur(B);
    HIP_CHECK(hipSetDevice(w->device));
    HIP_CHECK(hipMemcpyAsync(cur.data(), w->d_next.ptr, (size_t)B * 4, hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    std::vector<char> done(B, 0);
    for (int b = 0; b < B; ++b) n_out[b] = 0;
    for (int t = 0; t < max_new; ++t) {
      bool all_done = true;
      for (int b = 0; b < B; ++b) {
        if (!done[b]) {
          if (cur[b] == eos_id) done[b] = 1;                         // the stop token itself is not emitted (:648)
          else tokens_out[(size_t)b * max_new + n_out[b]++] = cur[b];
        }
        all_done = all_done && done[b];
      }
      if (all_done || t + 1 == max_new || w->hist + 1 > w->cfg.max_target_positions) break;
      // token ids stay on the device between steps; one small D2H per step for the stop test
      if (w->precision == ASR_PRECISION_BF16) w->step<bf16_t>(nullptr, 1, false, cur.data(), nullptr);
      else w->step<float>(nullptr, 1, false, cur.data(), nullptr);
    }
  });
}
