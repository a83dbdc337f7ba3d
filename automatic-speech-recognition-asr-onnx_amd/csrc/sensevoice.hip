// SenseVoiceSmall hot path on one MI355X: packed ragged batch -> fbank -> LFR/CMVN -> SANM blocks ->
// CTC arg-max + collapse. Follows SENSE_VOICE.forward (SenseVoice/Export_SenseVoice.py:271-296).
#include <cstring>

#include "../../include/asr_mi355x.h"
#include "engine.h"
#include "gemm.h"
#include "kernels.h"

namespace {

struct SvBlock {
  const float *ln1_g, *ln1_b, *bqkv, *wfsmn, *bfsmn, *ln2_g, *ln2_b, *b1, *b2;
  const void *wqkv, *wout, *w1, *w2;
  int in_size, kpad;
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct SvSession : asr_session {
  asr_sensevoice_config cfg;
  int feat = 0, kpad0 = 0, vpad = 0, max_lfr = 0;
  int n_bin_tiles = 0, n_kchunks = 0;
  std::vector<SvBlock> blocks;
  const float *dft = nullptr, *melp = nullptr, *cmvn_means = nullptr, *cmvn_vars = nullptr, *speech_pos = nullptr;
  const float *language_embed = nullptr, *system_embed = nullptr;
  const float *after_g = nullptr, *after_b = nullptr, *tp_g = nullptr, *tp_b = nullptr, *ctc_b = nullptr;
  const void* ctc_w = nullptr;

  // workspace (grow-only)
  DeviceBuffer d_plan, d_audio, d_mel, d_x0, d_xa, d_xb, d_h, d_qk, d_vt, d_ctx, d_mem, d_ffn, d_amax_v, d_amax_i, d_ids,
      d_tok, d_num, d_logits;
  void* h_plan = nullptr;   // pinned staging
  size_t h_plan_cap = 0;
  void* h_out = nullptr;
  size_t h_out_cap = 0;

  ~SvSession() override {
    for (DeviceBuffer* b : {&d_plan, &d_audio, &d_mel, &d_x0, &d_xa, &d_xb, &d_h, &d_qk, &d_vt, &d_ctx, &d_mem, &d_ffn,
                            &d_amax_v, &d_amax_i, &d_ids, &d_tok, &d_num, &d_logits})
      b->release();
    for (auto& kv : taps) kv.second.buf.release();
    if (h_plan) (void)hipHostFree(h_plan);
    if (h_out) (void)hipHostFree(h_out);
    prof.release();
    arena.release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  void init();
  template <typename T> void run(const float* audio, int audio_mem, const int64_t* offs, int batch, const int32_t* lang,
                                 int32_t* tok_out, int max_tokens, int32_t* num_out);
  void gemm(const GemmArgs& g) { precision == ASR_PRECISION_BF16 ? launch_gemm_bf16(g, stream) : launch_gemm_f32(g, stream); }
};

void SvSession::init() {
  const auto& c = cfg;
  ASR_REQUIRE(c.d_model == c.n_heads * c.d_head, "sensevoice: d_model != n_heads * d_head");
  ASR_REQUIRE(c.d_model % 128 == 0 && c.d_ffn % 128 == 0, "sensevoice: d_model and d_ffn must be multiples of 128");
  ASR_REQUIRE(c.d_head == 128 || (precision == ASR_PRECISION_F32 && c.d_head <= 128), "sensevoice: head_dim %d unsupported", c.d_head);
  ASR_REQUIRE(c.win_length == 400 && c.hop_length == 160, "sensevoice: front-end is built for 25 ms / 10 ms frames");
  ASR_REQUIRE(c.n_mels % 16 == 0, "sensevoice: n_mels must be a multiple of 16");
  ASR_REQUIRE(c.n_blocks >= 1 && c.n_main >= 1 && c.n_main <= c.n_blocks, "sensevoice: bad block counts");
  feat = c.n_mels * c.lfr_m;
  kpad0 = round_up(feat, 64);
  vpad = round_up(c.vocab, 128);
  n_bin_tiles = (c.nfft / 2 + 1 + 15) / 16;
  n_kchunks = c.win_length / 16;
  const int n_frames_max = (c.max_audio_len - c.win_length) / c.hop_length + 1;
  max_lfr = (n_frames_max + c.lfr_n - 1) / c.lfr_n;
  const int wt = precision == ASR_PRECISION_BF16 ? ARENA_BF16 : ARENA_F32;
  const int d = c.d_model, dff = c.d_ffn;

  dft = (const float*)arena.get("fe.dft", ARENA_F32, {(int64_t)n_bin_tiles * 2 * n_kchunks * 64 * 4}).ptr;
  melp = (const float*)arena.get("fe.mel", ARENA_F32, {(int64_t)(c.n_mels / 16) * n_bin_tiles * 64 * 4}).ptr;
  cmvn_means = (const float*)arena.get("fe.cmvn_means", ARENA_F32, {feat}).ptr;
  cmvn_vars = (const float*)arena.get("fe.cmvn_vars", ARENA_F32, {feat}).ptr;
  speech_pos = (const float*)arena.get("fe.speech_pos", ARENA_F32, {max_lfr, feat}).ptr;
  language_embed = (const float*)arena.get("fe.language_embed", ARENA_F32, {c.n_languages, feat}).ptr;
  system_embed = (const float*)arena.get("fe.system_embed", ARENA_F32, {c.n_prompt - 1, feat}).ptr;
  after_g = (const float*)arena.get("after_norm_g", ARENA_F32, {d}).ptr;
  after_b = (const float*)arena.get("after_norm_b", ARENA_F32, {d}).ptr;
  tp_g = (const float*)arena.get("tp_norm_g", ARENA_F32, {d}).ptr;
  tp_b = (const float*)arena.get("tp_norm_b", ARENA_F32, {d}).ptr;
  ctc_w = arena.get("ctc.w", wt, {vpad, d}).ptr;
  ctc_b = (const float*)arena.get("ctc.b", ARENA_F32, {vpad}).ptr;
  blocks.resize(c.n_blocks);
  for (int i = 0; i < c.n_blocks; ++i) {
    SvBlock& b = blocks[i];
    const std::string p = "blk" + std::to_string(i) + ".";
    b.in_size = (int)arena.get(p + "ln1_g").shape[0];
    ASR_REQUIRE(b.in_size == d || b.in_size == feat, "sensevoice: block %d has input size %d", i, b.in_size);
    b.kpad = round_up(b.in_size, 64);
    b.ln1_g = (const float*)arena.get(p + "ln1_g", ARENA_F32, {b.in_size}).ptr;
    b.ln1_b = (const float*)arena.get(p + "ln1_b", ARENA_F32, {b.in_size}).ptr;
    b.wqkv = arena.get(p + "wqkv", wt, {3 * d, b.kpad}).ptr;
    b.bqkv = (const float*)arena.get(p + "bqkv", ARENA_F32, {3 * d}).ptr;
    b.wfsmn = (const float*)arena.get(p + "wfsmn", ARENA_F32, {d, c.fsmn_kernel}).ptr;
    b.bfsmn = (const float*)arena.get(p + "bfsmn", ARENA_F32, {d}).ptr;
    b.wout = arena.get(p + "wout", wt, {d, d}).ptr;
    b.ln2_g = (const float*)arena.get(p + "ln2_g", ARENA_F32, {d}).ptr;
    b.ln2_b = (const float*)arena.get(p + "ln2_b", ARENA_F32, {d}).ptr;
    b.w1 = arena.get(p + "w1", wt, {dff, d}).ptr;
    b.b1 = (const float*)arena.get(p + "b1", ARENA_F32, {dff}).ptr;
    b.w2 = arena.get(p + "w2", wt, {d, dff}).ptr;
    b.b2 = (const float*)arena.get(p + "b2", ARENA_F32, {d}).ptr;
  }
}

template <typename T>
void SvSession::run(const float* audio, int audio_mem, const int64_t* offs, int batch, const int32_t* lang, int32_t* tok_out,
                    int max_tokens, int32_t* num_out) {
  const auto& c = cfg;
  ASR_REQUIRE(batch > 0, "sensevoice: empty batch");
  ASR_REQUIRE(audio && offs && lang && tok_out && num_out, "sensevoice: null argument");
  HIP_CHECK(hipSetDevice(device));
  const int d = c.d_model, dff = c.d_ffn;

  // ---- host plan -------------------------------------------------------------------------
  std::vector<UttPlan> plan(batch);
  int rows = 0, frames = 0, n_fb = 0, n_qb = 0, max_T = 0;
  const int64_t base0 = offs[0];
  for (int b = 0; b < batch; ++b) {
    const int64_t n = offs[b + 1] - offs[b];
    ASR_REQUIRE(n >= c.win_length, "sensevoice: utterance %d has %lld samples (< one %d-sample frame)", b, (long long)n, c.win_length);
    ASR_REQUIRE(n <= c.max_audio_len, "sensevoice: utterance %d has %lld samples (> max_audio_len %d)", b, (long long)n, c.max_audio_len);
    ASR_REQUIRE(lang[b] >= 0 && lang[b] < c.n_languages, "sensevoice: language_idx %d out of range", lang[b]);
    UttPlan& p = plan[b];
    p.audio_off = offs[b] - base0;
    p.n_samples = (int)n;
    p.n_frames = ((int)n - c.win_length) / c.hop_length + 1;
    p.frame_off = frames;
    p.n_lfr = (p.n_frames + c.lfr_n - 1) / c.lfr_n;
    p.T = p.n_lfr + c.n_prompt;
    p.row_off = rows;
    p.lang = lang[b];
    p.blk0 = n_fb;
    frames += p.n_frames;
    rows += round_up(p.T, 16);
    n_fb += (p.n_frames + 63) / 64;
    n_qb += (p.T + 63) / 64;
    max_T = std::max(max_T, p.T);
  }
  ASR_REQUIRE(max_tokens >= 1, "sensevoice: max_tokens must be positive");
  const int Mpad = round_up(rows, 128);
  const int64_t total_samples = offs[batch] - base0;

  // plan blob: [UttPlan x B][blk_utt n_fb][blk_f0 n_fb][qb_utt n_qb][qb_q0 n_qb][row_utt Mpad]
  const size_t plan_bytes = sizeof(UttPlan) * batch + sizeof(int32_t) * (2 * (size_t)n_fb + 2 * (size_t)n_qb + Mpad);
  if (plan_bytes > h_plan_cap) {
    if (h_plan) HIP_CHECK(hipHostFree(h_plan));
    HIP_CHECK(hipHostMalloc(&h_plan, plan_bytes * 2, hipHostMallocDefault));
    h_plan_cap = plan_bytes * 2;
  }
  unsigned char* hp = (unsigned char*)h_plan;
  memcpy(hp, plan.data(), sizeof(UttPlan) * batch);
  int32_t* blk_utt = (int32_t*)(hp + sizeof(UttPlan) * batch);
  int32_t* blk_f0 = blk_utt + n_fb;
  int32_t* qb_utt = blk_f0 + n_fb;
  int32_t* qb_q0 = qb_utt + n_qb;
  int32_t* row_utt = qb_q0 + n_qb;
  {
    int fi = 0, qi = 0;
    for (int b = 0; b < batch; ++b) {
      for (int f0 = 0; f0 < plan[b].n_frames; f0 += 64) { blk_utt[fi] = b; blk_f0[fi++] = f0; }
      for (int q0 = 0; q0 < plan[b].T; q0 += 64) { qb_utt[qi] = b; qb_q0[qi++] = q0; }
      const int r16 = round_up(plan[b].T, 16);
      for (int r = 0; r < r16; ++r) row_utt[plan[b].row_off + r] = b;
    }
    for (int r = rows; r < Mpad; ++r) row_utt[r] = -1;
  }
  d_plan.reserve(plan_bytes, stream);
  HIP_CHECK(hipMemcpyAsync(d_plan.ptr, h_plan, plan_bytes, hipMemcpyHostToDevice, stream));
  const UttPlan* dp = d_plan.as<UttPlan>();
  const int32_t* d_blk_utt = (const int32_t*)((unsigned char*)d_plan.ptr + sizeof(UttPlan) * batch);
  const int32_t* d_blk_f0 = d_blk_utt + n_fb;
  const int32_t* d_qb_utt = d_blk_f0 + n_fb;
  const int32_t* d_qb_q0 = d_qb_utt + n_qb;
  const int32_t* d_row_utt = d_qb_q0 + n_qb;

  // ---- workspace -------------------------------------------------------------------------
  const size_t eT = sizeof(T);
  const float* d_aud = nullptr;
  if (audio_mem == ASR_MEM_HOST) {
    d_audio.reserve((size_t)total_samples * 4, stream);
    HIP_CHECK(hipMemcpyAsync(d_audio.ptr, audio + base0, (size_t)total_samples * 4, hipMemcpyHostToDevice, stream));
    d_aud = d_audio.as<float>();
  } else {
    d_aud = audio + base0;
  }
  d_mel.reserve((size_t)frames * c.n_mels * 4, stream);
  d_x0.reserve((size_t)Mpad * kpad0 * 4, stream);
  d_xa.reserve((size_t)Mpad * d * 4, stream);
  d_xb.reserve((size_t)Mpad * d * 4, stream);
  d_h.reserve((size_t)Mpad * std::max(kpad0, d) * eT, stream);
  d_qk.reserve((size_t)Mpad * 2 * d * eT, stream);
  d_vt.reserve((size_t)Mpad * d * eT, stream);
  d_ctx.reserve((size_t)Mpad * d * eT, stream);
  d_mem.reserve((size_t)Mpad * d * 4, stream);
  d_ffn.reserve((size_t)Mpad * dff * eT, stream);
  const int n_slabs = vpad / 64;
  d_amax_v.reserve((size_t)Mpad * n_slabs * 4, stream);
  d_amax_i.reserve((size_t)Mpad * n_slabs * 4, stream);
  d_ids.reserve((size_t)Mpad * 4, stream);
  d_tok.reserve((size_t)batch * max_tokens * 4, stream);
  d_num.reserve((size_t)batch * 4, stream);

  // ---- 1. Kaldi fbank (Export_SenseVoice.py:275-278) ---------------------------------------
  {
    ProfScope ps(prof, "fbank", stream);
    FbankArgs fa;
    fa.audio = d_aud; fa.plan = dp; fa.blk_utt = d_blk_utt; fa.blk_f0 = d_blk_f0;
    fa.dft_packed = dft; fa.mel_packed = melp; fa.mel_out = d_mel.as<float>();
    fa.n_bin_tiles = n_bin_tiles; fa.n_kchunks = n_kchunks; fa.n_mel_tiles = c.n_mels / 16; fa.n_mels = c.n_mels;
    fa.win = c.win_length; fa.hop = c.hop_length; fa.log_floor = 1.1920928955078125e-07f; fa.whisper = 0; fa.blk_max = nullptr;
    launch_fbank(fa, n_fb, stream);
  }
  save_tap("mel", d_mel.ptr, frames, c.n_mels, c.n_mels, 4);
  // ---- 2./3. LFR + CMVN + positions + prompts (Export_SenseVoice.py:280-287) ---------------
  {
    ProfScope ps(prof, "lfr_cmvn", stream);
    LfrArgs la;
    la.mel = d_mel.as<float>(); la.plan = dp; la.row_utt = d_row_utt; la.cmvn_means = cmvn_means; la.cmvn_vars = cmvn_vars;
    la.speech_pos = speech_pos; la.language_embed = language_embed; la.system_embed = system_embed;
    la.out = d_x0.as<float>(); la.ld_out = kpad0; la.feat = feat; la.n_mels = c.n_mels; la.lfr_m = c.lfr_m; la.lfr_n = c.lfr_n;
    la.n_prompt = c.n_prompt; la.n_rows = Mpad;
    launch_lfr_cmvn(la, stream);
  }
  save_tap("enc_in", d_x0.ptr, rows, feat, kpad0, 4);

  // ---- 4. SANM blocks (Export_SenseVoice.py:227-269) ---------------------------------------
  const float* x_in = d_x0.as<float>();
  int ld_in = kpad0;
  float* xa = d_xa.as<float>();
  float* xb = d_xb.as<float>();
  T* h = d_h.as<T>();
  T* qk = d_qk.as<T>();
  T* vt = d_vt.as<T>();
  T* ctx = d_ctx.as<T>();
  float* mem = d_mem.as<float>();
  T* ffn = d_ffn.as<T>();
  for (int i = 0; i < c.n_blocks; ++i) {
    const SvBlock& b = blocks[i];
    {
      ProfScope ps(prof, "layernorm", stream);
      launch_layernorm<T>(x_in, ld_in, rows, b.in_size, b.ln1_g, b.ln1_b, 1e-5f, h, b.kpad, b.kpad, stream);
    }
    {
      ProfScope ps(prof, "gemm_qkv", stream);
      GemmArgs g;                               // q | k, row-major
      g.A = h; g.lda = b.kpad; g.W = b.wqkv; g.ldw = b.kpad; g.M = rows; g.N = 2 * d; g.K = b.kpad; g.bias = b.bqkv;
      g.out_lo = qk; g.ld_out_lo = 2 * d;
      gemm(g);
      GemmArgs gv;                              // v, stored transposed (time-contiguous) for the P.V operand and the FSMN
      gv.A = h; gv.lda = b.kpad; gv.W = (const T*)b.wqkv + (size_t)2 * d * b.kpad; gv.ldw = b.kpad; gv.M = rows; gv.N = d;
      gv.K = b.kpad; gv.bias = b.bqkv + 2 * d; gv.out_t = vt; gv.ld_out_t = Mpad;
      gemm(gv);
    }
    {
      ProfScope ps(prof, "fsmn", stream);
      launch_fsmn<T>(vt, Mpad, b.wfsmn, b.bfsmn, d, c.fsmn_kernel, dp, d_row_utt, Mpad, mem, d, stream);
    }
    {
      ProfScope ps(prof, "attention", stream);
      AttnArgs aa;
      aa.q = qk; aa.k = qk + d; aa.ld_qk = 2 * d; aa.vt = vt; aa.ld_vt = Mpad; aa.ctx = ctx; aa.ld_ctx = d;
      aa.plan = dp; aa.qb_utt = d_qb_utt; aa.qb_q0 = d_qb_q0; aa.n_qblocks = n_qb; aa.n_heads = c.n_heads;
      if (precision == ASR_PRECISION_BF16) launch_attention_bf16_hd128(aa, stream);
      else launch_attention_f32(aa, c.d_head, stream);
    }
    {
      ProfScope ps(prof, "gemm_out", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = b.wout; g.ldw = d; g.M = rows; g.N = d; g.K = d;
      g.add = mem; g.ld_add = d;                                  // FSMN memory rides in as the GEMM's additive term (:244)
      if (b.in_size == d) { g.add2 = x_in; g.ld_add2 = ld_in; }    // residual only when in/out sizes match (:246-256)
      g.out_f32 = xb; g.ld_out_f32 = d;
      gemm(g);
    }
    {
      ProfScope ps(prof, "layernorm", stream);
      launch_layernorm<T>(xb, d, rows, d, b.ln2_g, b.ln2_b, 1e-5f, h, d, d, stream);
    }
    {
      ProfScope ps(prof, "gemm_ffn1", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = b.w1; g.ldw = d; g.M = rows; g.N = dff; g.K = d; g.bias = b.b1; g.act = ACT_RELU;
      g.out_lo = ffn; g.ld_out_lo = dff;
      gemm(g);
    }
    {
      ProfScope ps(prof, "gemm_ffn2", stream);
      GemmArgs g;
      g.A = ffn; g.lda = dff; g.W = b.w2; g.ldw = dff; g.M = rows; g.N = d; g.K = dff; g.bias = b.b2;
      g.add = xb; g.ld_add = d; g.out_f32 = xa; g.ld_out_f32 = d;
      gemm(g);
    }
    x_in = xa;
    ld_in = d;
    if (i == 0) save_tap("block0", xa, rows, d, d, 4);
    if (i == c.n_main - 1) {
      ProfScope ps(prof, "layernorm", stream);
      launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, xa, d, d, stream);
    }
  }
  // tp_norm -> operand dtype for the CTC GEMM (f32 copy kept only for the tap)
  if (taps_enabled) {
    launch_layernorm<float>(xa, d, rows, d, tp_g, tp_b, 1e-5f, xb, d, d, stream);
    save_tap("enc_out", xb, rows, d, d, 4);
  }
  {
    ProfScope ps(prof, "layernorm", stream);
    launch_layernorm<T>(xa, d, rows, d, tp_g, tp_b, 1e-5f, h, d, d, stream);
  }
  // ---- 5. CTC head: GEMM with fused row arg-max, then circular collapse (:290-296) ----------
  {
    ProfScope ps(prof, "gemm_ctc", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = ctc_w; g.ldw = d; g.M = rows; g.N = vpad; g.K = d; g.bias = ctc_b;
    g.amax_val = d_amax_v.as<float>(); g.amax_idx = d_amax_i.as<int32_t>(); g.n_valid = c.vocab;
    if (taps_enabled) {
      d_logits.reserve((size_t)Mpad * vpad * 4, stream);
      g.out_f32 = d_logits.as<float>(); g.ld_out_f32 = vpad;
    }
    gemm(g);
  }
  {
    ProfScope ps(prof, "ctc_tail", stream);
    launch_argmax_reduce(d_amax_v.as<float>(), d_amax_i.as<int32_t>(), rows, n_slabs, d_ids.as<int32_t>(), stream);
    launch_ctc_collapse(d_ids.as<int32_t>(), dp, batch, c.blank_id, d_tok.as<int32_t>(), max_tokens, d_num.as<int32_t>(), stream);
  }
  if (taps_enabled) {
    save_tap("logits", d_logits.ptr, rows, c.vocab, vpad, 4);
    save_tap("frame_ids", d_ids.ptr, rows, 1, 1, 4);
  }
  // ---- outputs ---------------------------------------------------------------------------
  const size_t out_bytes = (size_t)batch * max_tokens * 4 + (size_t)batch * 4;
  if (out_bytes > h_out_cap) {
    if (h_out) HIP_CHECK(hipHostFree(h_out));
    HIP_CHECK(hipHostMalloc(&h_out, out_bytes * 2, hipHostMallocDefault));
    h_out_cap = out_bytes * 2;
  }
  HIP_CHECK(hipMemcpyAsync(h_out, d_tok.ptr, (size_t)batch * max_tokens * 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)batch * max_tokens * 4, d_num.ptr, (size_t)batch * 4,
                           hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  if (prof.enabled) prof.collect();
  memcpy(num_out, (unsigned char*)h_out + (size_t)batch * max_tokens * 4, (size_t)batch * 4);
  const int32_t* ht = (const int32_t*)h_out;
  for (int b = 0; b < batch; ++b) {
    const int n = std::min(num_out[b], max_tokens);
    memcpy(tok_out + (size_t)b * max_tokens, ht + (size_t)b * max_tokens, (size_t)n * 4);
  }
}

}  // namespace

extern "C" int asr_sensevoice_create(const asr_sensevoice_config* cfg, const void* arena, size_t arena_bytes, int arena_mem,
                                     int device_id, int precision, asr_session** out) {
  return asr_guard([&] {
    ASR_REQUIRE(cfg && arena && out, "sensevoice_create: null argument");
    ASR_REQUIRE(precision == ASR_PRECISION_BF16 || precision == ASR_PRECISION_F32, "sensevoice_create: bad precision %d", precision);
    asr_require_device(device_id);
    SvSession* s = new SvSession();
    try {
      s->kind = 1;
      s->device = device_id;
      s->precision = precision;
      s->cfg = *cfg;
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

extern "C" int asr_sensevoice_run(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                                  const int32_t* language_idx, int32_t* token_ids_out, int max_tokens, int32_t* num_id_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 1, "sensevoice_run: not a SenseVoice session");
    SvSession* sv = static_cast<SvSession*>(s);
    if (sv->precision == ASR_PRECISION_BF16)
      sv->run<bf16_t>(audio, audio_mem, audio_offsets, batch, language_idx, token_ids_out, max_tokens, num_id_out);
    else
      sv->run<float>(audio, audio_mem, audio_offsets, batch, language_idx, token_ids_out, max_tokens, num_id_out);
  });
}

extern "C" int asr_sensevoice_seq_len(const asr_sensevoice_config* cfg, int n_samples, int* seq_len) {
  return asr_guard([&] {
    ASR_REQUIRE(cfg && seq_len, "seq_len: null argument");
    ASR_REQUIRE(n_samples >= cfg->win_length, "seq_len: fewer samples than one frame");
    const int frames = (n_samples - cfg->win_length) / cfg->hop_length + 1;
    *seq_len = (frames + cfg->lfr_n - 1) / cfg->lfr_n + cfg->n_prompt;
  });
}
