// SenseVoiceSmall hot path on one MI355X: packed ragged batch -> fbank -> LFR/CMVN -> SANM blocks ->
// CTC arg-max + collapse. Follows SENSE_VOICE.forward (SenseVoice/Export_SenseVoice.py:271-296).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/asr_mi355x.h"
#include "engine.h"
#include "gemm.h"
#include "kernels.h"

namespace {

struct SvBlock {
  const float *ln1_g, *ln1_b, *bqkv, *wfsmn, *bfsmn, *ln2_g, *ln2_b, *b1, *b2;
  const float *cqkv, *c1;       // column sums of wqkv / w1 (bf16 arenas with folded LayerNorm affines): LayerNorm inside the GEMM
  const void *wqkv, *wout, *w1, *w2;
  int in_size, kpad;
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct SvRunCtx;

// Paraformer decoder block (Export_Paraformer.py:536-552): FFN (norm folded) -> LayerNorm -> FSMN + residual -> cross-attention
struct PfDecLayer {
  const void *w1, *w2, *wq, *wkv, *wo;
  const float *b1, *b2, *n2_g, *n2_b, *wfsmn, *bq, *bkv, *bo;
  bool full;     // false: FFN-only block (decoders3)
};

struct SvSession : asr_session {
  bool paraformer = false;
  asr_paraformer_config pcfg{};
  std::vector<PfDecLayer> pdec;
  const void *cif_conv_w = nullptr, *pf_out_w = nullptr;
  const float *cif_conv_b = nullptr, *cif_out_w = nullptr, *cif_out_b = nullptr, *pf_out_b = nullptr;
  DeviceBuffer d_enc_lo, d_ck, d_cifa, d_alpha, d_dec, d_x2, d_sa, d_ffn32, d_tplan;
  // the cross-attention K / V projections of ALL decoder layers as two GEMMs over the encoder output (they depend on nothing the decoder computes): the layers' wkv halves
  // gathered once into [n_full d][d] images (ASR_PF_KV_BATCH=0: one pair of launches per layer, the round-2..5 form)
  DeviceBuffer d_wk_all, d_wv_all, d_bk_all, d_bv_all; int pf_n_full = 0; bool pf_kv_batch = true;
  template <typename T> void enqueue_paraformer_tail(const struct SvRunCtx& r);

  asr_sensevoice_config cfg;
  int feat = 0, kpad0 = 0, vpad = 0, max_lfr = 0;
  int n_bin_tiles = 0, n_kchunks = 0;
  std::vector<SvBlock> blocks;
  const float *dft = nullptr, *melp = nullptr, *cmvn_means = nullptr, *cmvn_vars = nullptr, *speech_pos = nullptr;
  const float *language_embed = nullptr, *system_embed = nullptr;
  const float *after_g = nullptr, *after_b = nullptr, *tp_g = nullptr, *tp_b = nullptr, *ctc_b = nullptr;
  const void* ctc_w = nullptr;

  // workspace (grow-only)
  DeviceBuffer d_ctplan, d_mdev, d_trow;   // Paraformer: compact token plan, device-side token-row count, row -> utterance map of the token rows
  DeviceBuffer d_sta, d_stb;               // per-row (sum, sum of squares) partials of those copies, 32-column groups
  DeviceBuffer d_x0lo, d_xalo, d_xblo;     // bf16 copies of the residual stream (operands of the LayerNorm-fused projections)
  DeviceBuffer d_plan, d_audio, d_mel, d_x0, d_xa, d_xb, d_h, d_qk, d_vt, d_ctx, d_mem, d_ffn, d_amax_v, d_amax_i, d_ids,
      d_tok, d_num, d_logits;
  void* h_plan = nullptr;   // pinned staging
  size_t h_plan_cap = 0;
  void* h_out = nullptr;
  size_t h_out_cap = 0;

  ~SvSession() override {
    for (DeviceBuffer* b : {&d_dft_split, &d_times, &d_flags, &d_tpack, &d_tlayer_tab, &d_tkv, &d_ctplan, &d_mdev, &d_trow, &d_skws, &d_skcnt, &d_sta, &d_stb, &st_segs, &st_shadow, &st_wpack, &st_layer_tab, &st_flags, &st_times, &st_dpack, &st_dlayer_tab, &st_enk, &st_env, &st_dek, &st_dev, &st_defsmn, &st_prev, &st_cifh, &st_cifa, &st_enlen, &st_delen, &st_start, &d_sqkv, &d_skv,
                            &d_x0lo, &d_xalo, &d_xblo, &d_plan, &d_audio, &d_mel, &d_x0, &d_xa, &d_xb, &d_h, &d_qk, &d_vt, &d_ctx, &d_mem, &d_ffn,
                            &d_amax_v, &d_amax_i, &d_ids, &d_tok, &d_num, &d_logits, &d_enc_lo, &d_ck, &d_wk_all, &d_wv_all, &d_bk_all, &d_bv_all, &d_cifa, &d_alpha, &d_dec, &d_x2,
                            &d_sa, &d_ffn32, &d_tplan})
      b->release();
    for (auto& kv : taps) kv.second.buf.release();
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (hipGraphExec_t g : st_graph) if (g) (void)hipGraphExecDestroy(g);
    if (h_plan) (void)hipHostFree(h_plan);
    if (h_out) (void)hipHostFree(h_out);
    prof.release();
    arena.release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  // hipGraph replay of the forward pass (one graph per batch geometry)
  // ---- streaming Paraformer (kind 4): per-stream recurrent state in HBM, every step advances n streams by one chunk
  int st_chunk = 0, st_B = 0, st_C = 0, st_en_cap = 0, st_de_cap = 0, st_max = 0, st_frames = 0;
  DeviceBuffer st_enk, st_env, st_dek, st_dev, st_defsmn, st_prev, st_cifh, st_cifa, st_enlen, st_delen, st_start, d_sqkv, d_skv;
  // streaming encoder layers 1 .. as one launch (stream_layers.hip): fragment-major weight copies, the device layer table, exchange counters (+ err)
  DeviceBuffer st_wpack, st_layer_tab, st_flags, st_times, st_dpack, st_dlayer_tab;     // (st_d*: the decoder launch, stream_dec.hip)
  bool st_dec_fused = false;
  int st_times_layer = -1;              // ASR_STREAM_TIMES=<layer>: phase clocks of that fused layer, printed to stderr after every step (tuning)
  bool st_fused = false;
  int st_opt = 0;                       // ASR_STREAM_OPT: tuning switches of the fused launch (kernels.h: StreamLayersArgs::opt)
  int st_fused_env = 2;                 // ASR_STREAM_FUSED: 0 = every layer on the per-launch path, 1 = encoder launch only, 2 = encoder and decoder launches
  // Which steps take the cluster launches (round 5). (i) Above st_fused_max active streams the per-launch GEMMs amortise a layer's weights over all the rows
  // while a cluster launch streams them once per CU-load of clusters: 256 streams ran 10.4 ms fused against 9.0 per launch -- default from the
  // CU count, ASR_STREAM_FUSED_MAX overrides. (ii) While ANOTHER session of this process is computing on this GPU (asr_tenant_busy_others) the step takes the
  // per-launch path: (streams x 4) 150-KB workgroups hold every CU for a whole launch and starved a co-tenant's chain of small kernels 10x, and a cluster that is
  // split across dispatch waves by someone else's kernels can give up. ASR_STREAM_SHARE=0 disables the rule. (iii) A fused step is RECOVERABLE when a snapshot of
  // the active streams' recurrent state was taken in front of it (launch_stream_state_copy): on a give-up the state is restored, the step is redone on the per-launch
  // path and the session stays there for st_cooldown_steps steps. Snapshots are taken when other sessions exist on this GPU (they may start computing at any time)
  // or always with ASR_STREAM_SNAPSHOT=1 (=0: never; a give-up then fails the step and the caller has to reset its streams, as in round 4).
  int st_fused_max = 0, st_share_rule = 1, st_snapshot_env = -1, st_fault = 0, st_cooldown = 0;
  int st_giveups = 0, st_shared_steps = 0, st_snapshots = 0;      // counters (asr_paraformer_stream_stats)
  static constexpr int st_cooldown_steps = 64;
  DeviceBuffer st_segs, st_shadow;      // StreamStateSeg table + the shadow copies
  int st_n_segs = 0, st_n_items = 0;
  void ensure_stream_shadow();
  void stream_init(int chunk, int look_back_encoder, int look_back_decoder, int max_streams);
  void stream_reset(int sid);
  template <typename T> void stream_step(const float* audio, int audio_mem, const int32_t* stream_ids, int n, int32_t* tok_out, int max_tokens,
                                         int32_t* num_out);
  bool use_graph = true;
  bool use_ln_alg = true;       // LayerNorm evaluated inside the projections from row statistics (ASR_LN_FUSED=0 disables)
  bool use_fused = true;        // fused q|k|v + attention + FSMN kernel for windows of <= 144 rows (ASR_SANM_FUSED=0 disables)
  bool use_block = true;        // one launch per SANM block (clusters of four workgroups per window; ASR_SANM_BLOCK=0 disables)
  bool use_fbank_split = true;  // ASR_FBANK_SPLIT=0: exact-f32 MFMA DFT in bf16 sessions too
  DeviceBuffer d_dft_split;
  int block8_opt = 0;           // ASR_SANM_BLOCK8_OPT: tuning switches of the 8-wave kernel (SanmBlockArgs::opt)
  bool block_persist = true;    // ASR_SANM_BLOCK_PERSIST=0: the 8-wave kernel is launched once per block instead of once per run of blocks
  DeviceBuffer d_layer_tab;     // SanmBlockLayer[n_blocks]: the per-block constants a launch of the 8-wave kernel walks
  // small batches: a workgroup per (16-row tile, head), csrc/sanm_tiles.hip (ASR_SANM_TILES=0: four launches per block as before)
  bool use_tiles = true, tpack_ready = false;
  int tiles_opt = 0, tiles_dbg = -1;       // ASR_SANM_TILES_OPT (SanmTilesArgs::opt), ASR_SANM_TILES_DBG=<block>: phase clock of that block on stderr
  DeviceBuffer d_tpack, d_tlayer_tab, d_tkv;
  void ensure_tiles_pack();
  DeviceBuffer d_wpack;         // 8-wave form: fragment-major copy of every 512 -> 512 block's weights, made once per session (ensure_block_pack)
  bool wpack_ready = false;
  void ensure_block_pack();
  int block_fault = 0;          // ASR_SANM_BLOCK_FAULT=1 (tests): one workgroup of the first block launch withholds an exchange count
  int block_giveups = 0;        // forward passes redone on the four-launch path because a cluster gave up (see run())
  bool block_ffnk = true;       // ASR_SANM_BLOCK_FFNK=0: the round-4 form of the block kernel's FFN pair (hid exchanged); default: FFN-2 split over K, f16 partials exchanged (round 6)
  bool foreign_now = false;     // a foreign section (RCCL collective, engine.h: ClusterScope) is open on this GPU: this pass launches no cluster kernel
  int foreign_diverted = 0;     // passes that took the cluster-free path for that reason
  int block_cooldown = 0;       // batches left on the four-launch path after a give-up (other sessions are holding CUs: do not walk into the same wait again)
  void load_env() {             // debug / ablation switches, re-read at every session creation
    gemm_reload_env();
    if (const char* e = getenv("ASR_NO_GRAPH")) use_graph = !(e[0] == '1');
    if (const char* e = getenv("ASR_SANM_FUSED")) use_fused = !(e[0] == '0');
    if (const char* e = getenv("ASR_SANM_BLOCK")) use_block = !(e[0] == '0');
    if (const char* e = getenv("ASR_FBANK_SPLIT")) use_fbank_split = !(e[0] == '0');
    if (const char* e = getenv("ASR_SANM_BLOCK8_OPT")) block8_opt = atoi(e);
    if (const char* e = getenv("ASR_SANM_BLOCK_FFNK")) block_ffnk = !(e[0] == '0');
    if ((block8_opt >> 4) & 15) block_ffnk = false;          // the timing-only ablations exist for the round-4 loops
    if (const char* e = getenv("ASR_SANM_TILES")) use_tiles = !(e[0] == '0');
    if (const char* e = getenv("ASR_SANM_TILES_OPT")) tiles_opt = atoi(e);
    if (const char* e = getenv("ASR_SANM_TILES_DBG")) tiles_dbg = atoi(e);
    if (const char* e = getenv("ASR_SANM_BLOCK_PERSIST")) block_persist = !(e[0] == '0');
    if (const char* e = getenv("ASR_SANM_BLOCK_SCATTER")) block_scatter = e[0] == '1';
    if (const char* e = getenv("ASR_SANM_BLOCK_FAULT")) block_fault = e[0] == '1';
    if (const char* e = getenv("ASR_SANM_BLOCK_DBG")) block_dbg = atoi(e);
    if (const char* e = getenv("ASR_SANM_BLOCK_MIN")) block_min_utts = atoi(e);
    if (const char* e = getenv("ASR_LN_FUSED")) use_ln_alg = !(e[0] == '0');
  }
  int block_scatter = 0;        // ASR_SANM_BLOCK_SCATTER=1: test placement, every cluster spread over four XCDs
  int block_min_utts = 12;      // ASR_SANM_BLOCK_MIN=<windows>: smallest batch that takes the block kernel (8-wave form: faster from ~10 windows on; tools/probes/block_min_sweep.sh)
  DeviceBuffer d_times; int block_dbg = -1;   // ASR_SANM_BLOCK_DBG=<block index>: phase clock of that block's launch on stderr
  DeviceBuffer d_flags;         // exchange counters of the block kernel: [n_blocks][batch][4] + the error word at the end
  hipGraphExec_t graph_exec = nullptr;
  uint64_t graph_key = 0, eager_key = 0, ws_epoch = 1;
  hipGraphExec_t st_graph[3] = {nullptr, nullptr, nullptr}; uint64_t st_graph_key[3] = {0, 0, 0}, st_eager_key[3] = {0, 0, 0};     // streaming chunk step: per-launch / fused / fused + snapshot

  void init();
  void copy_block_status(const struct SvRunCtx& r);   // the block kernel's error word rides home behind the token counts
  template <typename T> void enqueue(const struct SvRunCtx& r);
  template <typename T> void run(const float* audio, int audio_mem, const int64_t* offs, int batch, const int32_t* lang,
                                 int32_t* tok_out, int max_tokens, int32_t* num_out);
  DeviceBuffer d_skws, d_skcnt;        // split-K workspace + tickets of the skinny GEMM (per session: sessions may run concurrently)
  void gemm(const GemmArgs& g0) {
    if (!d_skws.ptr) { d_skws.reserve((size_t)16 << 20, stream); d_skcnt.reserve(4096 * 4, stream); }
    GemmArgs g = g0;
    g.sk_ws = d_skws.as<float>(); g.sk_ws_bytes = d_skws.cap; g.sk_cnt = d_skcnt.as<int32_t>();
    if (precision != ASR_PRECISION_BF16) { launch_gemm_f32(g, stream); return; }      // (the workspace lets single windows split K)
    launch_gemm_bf16(g, stream);
  }
};

void SvSession::init() {
  const auto& c = cfg;
  ASR_REQUIRE(c.d_model == c.n_heads * c.d_head, "sensevoice: d_model != n_heads * d_head");
  ASR_REQUIRE(c.d_model % 128 == 0 && c.d_ffn % 128 == 0, "sensevoice: d_model and d_ffn must be multiples of 128");
  ASR_REQUIRE(c.d_head == 128 || (precision == ASR_PRECISION_F32 && c.d_head <= 128), "sensevoice: head_dim %d unsupported", c.d_head);
  ASR_REQUIRE(c.win_length == 400 && c.hop_length == 160, "sensevoice: front-end is built for 25 ms / 10 ms frames");
  ASR_REQUIRE(c.n_mels % 16 == 0, "sensevoice: n_mels must be a multiple of 16");
  ASR_REQUIRE(c.n_blocks >= 1 && c.n_main >= 1 && c.n_main <= c.n_blocks, "sensevoice: bad block counts");
  auto opt = [&](const std::string& n, std::initializer_list<int64_t> sh) -> const float* {
    return arena.has(n) ? (const float*)arena.get(n, ARENA_F32, sh).ptr : nullptr;
  };
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
  if (precision == ASR_PRECISION_BF16 && use_fbank_split) {       // bf16 sessions: the DFT on the bf16 pipe with split operands (kernels.hip: fbank_split_kernel)
    d_dft_split.reserve(fbank_split_table_bytes(n_bin_tiles, cfg.win_length), stream);
    launch_fbank_split_table(dft, n_bin_tiles, n_kchunks, d_dft_split.ptr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  melp = (const float*)arena.get("fe.mel", ARENA_F32, {(int64_t)(c.n_mels / 16) * n_bin_tiles * 64 * 4}).ptr;
  cmvn_means = opt("fe.cmvn_means", {feat});
  cmvn_vars = (const float*)arena.get("fe.cmvn_vars", ARENA_F32, {feat}).ptr;
  speech_pos = (const float*)arena.get("fe.speech_pos", ARENA_F32, {max_lfr, feat}).ptr;
  language_embed = c.n_prompt > 0 ? (const float*)arena.get("fe.language_embed", ARENA_F32, {c.n_languages, feat}).ptr : nullptr;
  system_embed = c.n_prompt > 1 ? (const float*)arena.get("fe.system_embed", ARENA_F32, {c.n_prompt - 1, feat}).ptr : nullptr;
  after_g = (const float*)arena.get("after_norm_g", ARENA_F32, {d}).ptr;
  after_b = (const float*)arena.get("after_norm_b", ARENA_F32, {d}).ptr;
  if (!paraformer) {
    tp_g = (const float*)arena.get("tp_norm_g", ARENA_F32, {d}).ptr;
    tp_b = (const float*)arena.get("tp_norm_b", ARENA_F32, {d}).ptr;
    ctc_w = arena.get("ctc.w", wt, {vpad, d}).ptr;
    ctc_b = (const float*)arena.get("ctc.b", ARENA_F32, {vpad}).ptr;
  } else {
    const int dd = pcfg.d_dec_ffn;
    ASR_REQUIRE(dd % 128 == 0 && dd <= 2048 && (3 * d) % 64 == 0 && pcfg.cif_kernel == 3, "paraformer: unsupported decoder geometry");
    cif_conv_w = arena.get("cif.conv_w", wt, {d, 3 * d}).ptr;
    cif_conv_b = (const float*)arena.get("cif.conv_b", ARENA_F32, {d}).ptr;
    cif_out_w = (const float*)arena.get("cif.out_w", ARENA_F32, {d}).ptr;
    cif_out_b = (const float*)arena.get("cif.out_b", ARENA_F32, {1}).ptr;
    pf_out_w = arena.get("out.w", wt, {vpad, d}).ptr;
    pf_out_b = (const float*)arena.get("out.b", ARENA_F32, {vpad}).ptr;
    pdec.resize(pcfg.n_dec + pcfg.n_dec3);
    for (int j = 0; j < (int)pdec.size(); ++j) {
      PfDecLayer& L = pdec[j];
      const std::string q = "dec" + std::to_string(j) + ".";
      L.full = j < pcfg.n_dec;
      L.w1 = arena.get(q + "w1", wt, {dd, d}).ptr;
      L.b1 = (const float*)arena.get(q + "b1", ARENA_F32, {dd}).ptr;
      L.w2 = arena.get(q + "w2", wt, {d, dd}).ptr;
      L.b2 = (const float*)arena.get(q + "b2", ARENA_F32, {d}).ptr;
      if (L.full) {
        L.n2_g = (const float*)arena.get(q + "n2_g", ARENA_F32, {d}).ptr;
        L.n2_b = (const float*)arena.get(q + "n2_b", ARENA_F32, {d}).ptr;
        L.wfsmn = (const float*)arena.get(q + "wfsmn", ARENA_F32, {d, c.fsmn_kernel}).ptr;
        L.wq = arena.get(q + "wq", wt, {d, d}).ptr;
        L.bq = (const float*)arena.get(q + "bq", ARENA_F32, {d}).ptr;
        L.wkv = arena.get(q + "wkv", wt, {2 * d, d}).ptr;
        L.bkv = (const float*)arena.get(q + "bkv", ARENA_F32, {2 * d}).ptr;
        L.wo = arena.get(q + "wo", wt, {d, d}).ptr;
        L.bo = (const float*)arena.get(q + "bo", ARENA_F32, {d}).ptr;
      }
    }
    if (const char* e = getenv("ASR_PF_KV_BATCH")) pf_kv_batch = !(e[0] == '0');
    pf_n_full = 0;
    for (const PfDecLayer& L : pdec) pf_n_full += L.full ? 1 : 0;
    if (pf_kv_batch && pf_n_full > 0) {
      const size_t ew = wt == ARENA_BF16 ? 2 : 4, wbytes = (size_t)d * d * ew;
      d_wk_all.reserve(pf_n_full * wbytes, stream); d_wv_all.reserve(pf_n_full * wbytes, stream);
      d_bk_all.reserve((size_t)pf_n_full * d * 4, stream); d_bv_all.reserve((size_t)pf_n_full * d * 4, stream);
      int li = 0;
      for (const PfDecLayer& L : pdec) {
        if (!L.full) continue;
        HIP_CHECK(hipMemcpyAsync((unsigned char*)d_wk_all.ptr + li * wbytes, L.wkv, wbytes, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync((unsigned char*)d_wv_all.ptr + li * wbytes, (const unsigned char*)L.wkv + wbytes, wbytes, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_bk_all.as<float>() + (size_t)li * d, L.bkv, (size_t)d * 4, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_bv_all.as<float>() + (size_t)li * d, L.bkv + d, (size_t)d * 4, hipMemcpyDeviceToDevice, stream));
        ++li;
      }
    }
  }
  blocks.resize(c.n_blocks);
  for (int i = 0; i < c.n_blocks; ++i) {
    SvBlock& b = blocks[i];
    const std::string p = "blk" + std::to_string(i) + ".";
    b.kpad = (int)arena.get(p + "wqkv").shape[1];
    b.in_size = b.kpad == d ? d : feat;
    ASR_REQUIRE(b.kpad == round_up(b.in_size, 64), "sensevoice: block %d has K = %d", i, b.kpad);
    b.ln1_g = opt(p + "ln1_g", {b.in_size});             // absent when the affine is folded into the Linear (Paraformer)
    b.ln1_b = opt(p + "ln1_b", {b.in_size});
    b.wqkv = arena.get(p + "wqkv", wt, {3 * d, b.kpad}).ptr;
    b.bqkv = (const float*)arena.get(p + "bqkv", ARENA_F32, {3 * d}).ptr;
    b.wfsmn = (const float*)arena.get(p + "wfsmn", ARENA_F32, {d, c.fsmn_kernel}).ptr;
    b.bfsmn = (const float*)arena.get(p + "bfsmn", ARENA_F32, {d}).ptr;
    b.wout = arena.get(p + "wout", wt, {d, d}).ptr;
    b.cqkv = opt(p + "cqkv", {3 * d});
    b.c1 = opt(p + "c1", {dff});
    b.ln2_g = opt(p + "ln2_g", {d});
    b.ln2_b = opt(p + "ln2_b", {d});
    b.w1 = arena.get(p + "w1", wt, {dff, d}).ptr;
    b.b1 = (const float*)arena.get(p + "b1", ARENA_F32, {dff}).ptr;
    b.w2 = arena.get(p + "w2", wt, {d, dff}).ptr;
    b.b2 = (const float*)arena.get(p + "b2", ARENA_F32, {d}).ptr;
  }
}

// Everything a forward pass needs once the host plan is uploaded; captured into a hipGraph for replay.
struct SvRunCtx {
  int batch, rows, Mpad, frames, n_fb, n_qb, max_T, max_tokens, att_qt, att_nw;
  const float* d_aud;
  const UttPlan* dp;
  const int32_t *d_blk_utt, *d_blk_f0, *d_qb_utt, *d_qb_q0, *d_row_utt;
  const int32_t *d_tile_win = nullptr, *d_tile_idx = nullptr;    // small batches (sanm_tiles.hip): per 16-row tile its window / its index inside the window
  int n_tiles = 0;
  bool tiles = false;
};

// fragment-major weight copies for the 8-wave block kernel: one pass over the arena's bf16 matrices per session, outside any graph capture
void SvSession::ensure_block_pack() {
  if (wpack_ready) return;
  const size_t per = sanm_block8_pack_bytes();
  d_wpack.reserve(per * cfg.n_blocks, stream);
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const SvBlock& b = blocks[i];
    if (b.in_size != cfg.d_model) continue;              // (block 0 maps 560 -> 512: it keeps the separate launches)
    launch_sanm_block8_pack((const bf16_t*)b.wqkv, (const bf16_t*)b.wout, (const bf16_t*)b.w1, (const bf16_t*)b.w2, (unsigned char*)d_wpack.ptr + per * i, block_ffnk, stream);
  }
  std::vector<SanmBlockLayer> tab(cfg.n_blocks);
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const SvBlock& b = blocks[i];
    tab[i] = SanmBlockLayer{b.bqkv, b.cqkv, b.wfsmn, b.bfsmn, b.b1, b.c1, b.b2, (const unsigned char*)d_wpack.ptr + per * i};
  }
  d_layer_tab.reserve(tab.size() * sizeof(SanmBlockLayer), stream);
  HIP_CHECK(hipMemcpyAsync(d_layer_tab.ptr, tab.data(), tab.size() * sizeof(SanmBlockLayer), hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  wpack_ready = true;
}

void SvSession::copy_block_status(const SvRunCtx& r) {
  const size_t flag_words = (size_t)cfg.n_blocks * r.batch * 4;
  HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)r.batch * r.max_tokens * 4 + (size_t)r.batch * 4, d_flags.as<unsigned>() + flag_words, 4,
                           hipMemcpyDeviceToHost, stream));
}

// the same for the tile kernel of small batches: the streaming encoder's image format and table entries
void SvSession::ensure_tiles_pack() {
  if (tpack_ready) return;
  const size_t per = stream_layers_pack_bytes();
  d_tpack.reserve(per * cfg.n_blocks, stream);
  std::vector<StreamLayer> tab(cfg.n_blocks);
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const SvBlock& b = blocks[i];
    tab[i] = StreamLayer{};
    if (b.in_size != cfg.d_model) continue;
    unsigned char* dst = (unsigned char*)d_tpack.ptr + per * i;
    launch_stream_layers_pack((const bf16_t*)b.wqkv, (const bf16_t*)b.wout, (const bf16_t*)b.w1, (const bf16_t*)b.w2, dst, stream);
    tab[i].wpack = dst; tab[i].bqkv = b.bqkv; tab[i].wfsmn = b.wfsmn; tab[i].bfsmn = b.bfsmn; tab[i].b1 = b.b1; tab[i].b2 = b.b2;
  }
  d_tlayer_tab.reserve(tab.size() * sizeof(StreamLayer), stream);
  HIP_CHECK(hipMemcpyAsync(d_tlayer_tab.ptr, tab.data(), tab.size() * sizeof(StreamLayer), hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  tpack_ready = true;
}

template <typename T>
void SvSession::enqueue(const SvRunCtx& r) {
  const auto& c = cfg;
  const int d = c.d_model, dff = c.d_ffn, rows = r.rows, Mpad = r.Mpad;
  const int n_slabs = vpad / 64;
  // ---- 1. Kaldi fbank (Export_SenseVoice.py:275-278) ---------------------------------------
  {
    ProfScope ps(prof, "fbank", stream);
    FbankArgs fa;
    fa.audio = r.d_aud; fa.plan = r.dp; fa.blk_utt = r.d_blk_utt; fa.blk_f0 = r.d_blk_f0;
    fa.dft_packed = dft; fa.mel_packed = melp; fa.mel_out = d_mel.as<float>();
    fa.n_bin_tiles = n_bin_tiles; fa.n_kchunks = n_kchunks; fa.n_mel_tiles = c.n_mels / 16; fa.n_mels = c.n_mels;
    fa.win = c.win_length; fa.hop = c.hop_length; fa.log_floor = 1.1920928955078125e-07f; fa.whisper = 0; fa.blk_max = nullptr;
    fa.dft_split = d_dft_split.ptr;
    launch_fbank(fa, r.n_fb, stream);
  }
  save_tap("mel", d_mel.ptr, r.frames, c.n_mels, c.n_mels, 4);
  // LayerNorm inside the projections (bf16 mode, 8 s windows): the residual stream is kept in f32 AND as its bf16 rounding;
  // the fused attention kernel and the FFN-1 GEMM read the raw bf16 rows, accumulate the row statistics from their LDS tiles
  // and apply rstd (x W^T - mean colsum(W)) + b in the epilogue -- no LayerNorm launches, no normalised copy in memory.
  bool alg = false;
  if constexpr (sizeof(T) == 2) {
    const SvBlock& b1 = blocks[c.n_blocks - 1];
    GemmArgs probe;
    probe.M = rows; probe.N = dff; probe.K = d; probe.ln_dim = d; probe.ln_colsum = b1.c1; probe.act = ACT_RELU; probe.bias = b1.b1;
    probe.out_lo = d_ffn.ptr; probe.A = d_xblo.ptr; probe.W = b1.w1;
    // a single window is weight-streaming bound: its GEMMs take the skinny split-K kernel, which wants separately normalised rows
    const bool skinny144 = gemm_skinny144_enabled();
    alg = (rows > 144 || !skinny144) && use_ln_alg && use_fused && b1.cqkv && b1.c1 && blocks[0].cqkv &&
          sanm_fused_supported(r.max_T, c.d_head, c.n_heads, d, c.fsmn_kernel, blocks[0].kpad) && gemm_ln_fusable(probe);
  }
  bf16_t* x0lo = d_x0lo.as<bf16_t>();
  bf16_t* xalo = d_xalo.as<bf16_t>();
  bf16_t* xblo = d_xblo.as<bf16_t>();
  // ---- 2./3. LFR + CMVN + positions + prompts (Export_SenseVoice.py:280-287) ---------------
  {
    ProfScope ps(prof, "lfr_cmvn", stream);
    LfrArgs la;
    if (alg) la.out_lo = x0lo;
    la.mel = d_mel.as<float>(); la.plan = r.dp; la.row_utt = r.d_row_utt; la.cmvn_means = cmvn_means; la.cmvn_vars = cmvn_vars;
    la.speech_pos = speech_pos; la.language_embed = language_embed; la.system_embed = system_embed;
    la.out = d_x0.as<float>(); la.ld_out = kpad0; la.feat = feat; la.n_mels = c.n_mels; la.lfr_m = c.lfr_m; la.lfr_n = c.lfr_n;
    la.n_prompt = c.n_prompt; la.n_rows = Mpad; la.affine_mode = paraformer ? 1 : 0;
    launch_lfr_cmvn(la, stream);
  }
  save_tap("enc_in", d_x0.ptr, rows, feat, kpad0, 4);

  // ---- 4. SANM blocks (Export_SenseVoice.py:227-269) ---------------------------------------
  // one launch per block (csrc/sanm_block.hip) when every window fits a 144-row tile and the LayerNorms are folded into the projections
  // (it has its own tiling, so unlike `alg` it does not need batches of near-full windows: ragged batches qualify too)
  bool blk = false;
  if constexpr (sizeof(T) == 2)
    blk = use_block && block_cooldown == 0 && !foreign_now && use_ln_alg && use_fused && blocks[c.n_blocks - 1].cqkv && blocks[c.n_blocks - 1].c1 && c.n_blocks > 1 &&
          r.batch >= block_min_utts &&          // four workgroups per window: a small batch leaves most CUs idle (one window: 4 of 256), the tiled GEMMs do not
          sanm_block_supported(r.max_T, c.d_head, c.n_heads, d, dff, c.fsmn_kernel);
  const size_t flag_words = (size_t)c.n_blocks * r.batch * 4;
  const size_t tile_flag0 = flag_words + 4 + (size_t)c.n_blocks * r.batch, tile_flag_stride = ((size_t)r.n_tiles + r.batch) * 4;       // the tile kernel's counters: per block [tile][4] + [window][4]
  const bool tiles = r.tiles && block_cooldown == 0 && !foreign_now;
  HIP_CHECK(hipMemsetAsync(d_flags.ptr, 0, (tile_flag0 + (tiles ? (size_t)c.n_blocks * tile_flag_stride : 0)) * 4, stream));      // per (block, window, exchange) counters + the error word + per (launch, window) placement words
  const float* x_in = d_x0.as<float>();
  const bf16_t* x_in_lo = x0lo;
  const float2* st_in = nullptr;            // statistics of x_in_lo's rows when its producer wrote them
  bool st_in_block8 = false;                // ... by the 8-wave block kernel: one record per row and workgroup instead of one per 32 columns
  float2* sta = d_sta.as<float2>();
  float2* stb = d_stb.as<float2>();
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
    if constexpr (sizeof(T) == 2) {
      if (blk && b.in_size == d && x_in == xa) {             // (block 0 maps 560 -> 512 without a residual: it keeps the separate launches)
        if (!alg && i == 1) {                                // block 0 ran without the bf16 copy / statistics outputs: make the copy, derive the statistics in the kernel
          ProfScope ps0(prof, "layernorm", stream);
          launch_rows_to_bf16(xa, xalo, (size_t)Mpad * d, stream);
          st_in = nullptr;
        }
        const int per = sanm_block_max_utts();
        // 8-wave kernel: one launch walks every block up to the next stand-alone LayerNorm (blocks 1 .. n_main - 1, then n_main .. n_blocks - 1): a window's
        // blocks depend on its own cluster only (ASR_SANM_BLOCK_PERSIST=0: one launch per block)
        const int run_end = (!paraformer && i < c.n_main) ? c.n_main : c.n_blocks;
        const int n_run = block_persist ? run_end - i : 1;
        for (int u0 = 0; u0 < r.batch; u0 += per) {
          ProfScope ps(prof, "sanm_block", stream);
          SanmBlockArgs ba{};
          ba.wqkv = (const bf16_t*)b.wqkv; ba.bqkv = b.bqkv; ba.cqkv = b.cqkv; ba.wfsmn = b.wfsmn; ba.bfsmn = b.bfsmn;
          ba.wout = (const bf16_t*)b.wout; ba.w1 = (const bf16_t*)b.w1; ba.b1 = b.b1; ba.c1 = b.c1; ba.w2 = (const bf16_t*)b.w2; ba.b2 = b.b2;
          ba.x_lo = xalo; ba.st_in = st_in; ba.x = xa; ba.x_lo_out = xalo; ba.st_out = sta;
          ba.ctx = (bf16_t*)ctx; ba.x1_lo = xblo; ba.st1 = stb; ba.hid = (bf16_t*)ffn;
          ba.plan = r.dp; ba.utt0 = u0; ba.n_utts = std::min(per, r.batch - u0);
          ba.flags = d_flags.as<unsigned>() + ((size_t)i * r.batch + u0) * 4; ba.err = d_flags.as<unsigned>() + flag_words;
          ba.flag_stride = r.batch * 4; ba.place = d_flags.as<unsigned>() + flag_words + 4 + (size_t)i * r.batch + u0;
          ba.n_rows_alloc = Mpad; ba.ln_eps = 1e-5f; ba.scatter = block_scatter; ba.fault = (block_fault && i == 1) ? 1 : 0;
          if (block_dbg >= i && block_dbg < i + n_run && u0 == 0) {
            d_times.reserve(256 * 16 * 8, stream); HIP_CHECK(hipMemsetAsync(d_times.ptr, 0, 256 * 16 * 8, stream)); ba.times = d_times.as<unsigned long long>();
            ba.times_layer = block_dbg - i;
          }
          ba.layers = d_layer_tab.as<SanmBlockLayer>() + i; ba.n_layers = n_run; ba.opt = block8_opt; ba.ffnk = block_ffnk ? 1 : 0;
          ba.st_in_n = st_in_block8 ? 4 : 16;
          launch_sanm_block8(ba, stream);
        }
        i += n_run - 1;                                        // (the loop header steps over the last block of the run)
        st_in = sta;
        st_in_block8 = true;
        if (i == c.n_main - 1 && !paraformer) {
          ProfScope ps2(prof, "layernorm", stream);
          launch_layernorm<bf16_t>(xa, d, rows, d, after_g, after_b, 1e-5f, xalo, d, d, stream); st_in = nullptr;
          launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, xa, d, d, stream);
        }
        continue;
      }
    }
    if constexpr (sizeof(T) == 2) {
      if (tiles && !blk && b.in_size == d && x_in == xa) {   // small batch: every block up to the next stand-alone LayerNorm in one launch of (tile, head) workgroups
        const int run_end = (!paraformer && i < c.n_main) ? c.n_main : c.n_blocks;
        {
          ProfScope ps(prof, "sanm_tiles", stream);
          SanmTilesArgs ta;
          ta.plan = r.dp; ta.tile_win = r.d_tile_win; ta.tile_idx = r.d_tile_idx; ta.n_tiles = r.n_tiles; ta.n_windows = r.batch; ta.n_layers = run_end - i;
          ta.ln_eps = 1e-5f; ta.layers = d_tlayer_tab.as<StreamLayer>() + i;
          ta.x = xa; ta.xb = xb; ta.ctx = (bf16_t*)ctx; ta.hid = (bf16_t*)ffn; ta.kv = d_tkv.as<bf16_t>(); ta.kv_parity_stride = (size_t)Mpad * 2 * d;
          ta.flags = d_flags.as<unsigned>() + tile_flag0 + (size_t)i * tile_flag_stride; ta.flag_stride = (int)tile_flag_stride;
          ta.err = d_flags.as<unsigned>() + flag_words; ta.opt = tiles_opt;
          if (tiles_dbg >= i && tiles_dbg < run_end) {
            d_times.reserve(256 * 16 * 8, stream); HIP_CHECK(hipMemsetAsync(d_times.ptr, 0, 256 * 16 * 8, stream)); ta.times = d_times.as<unsigned long long>();
            ta.times_layer = tiles_dbg - i;
          }
          launch_sanm_tiles(ta, stream);
        }
        i = run_end - 1;
        st_in = nullptr;
        if (i == c.n_main - 1 && !paraformer) {
          ProfScope ps2(prof, "layernorm", stream);
          launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, xa, d, d, stream);
        }
        continue;
      }
    }
    if (!alg) {
      ProfScope ps(prof, "layernorm", stream);
      launch_layernorm<T>(x_in, ld_in, rows, b.in_size, b.ln1_g, b.ln1_b, 1e-5f, h, b.kpad, b.kpad, stream);
    }
    const bool fused = use_fused && precision == ASR_PRECISION_BF16 &&
                       sanm_fused_supported(r.max_T, c.d_head, c.n_heads, d, c.fsmn_kernel, b.kpad);
    if (fused) {
      ProfScope ps(prof, "sanm_fused", stream);
      SanmFusedArgs fa;
      fa.h = h; fa.ld_h = b.kpad; fa.K = b.kpad;
      if (alg) { fa.h = x_in_lo; fa.ln_colsum = b.cqkv; fa.ln_dim = b.in_size; fa.ln_stats_in = st_in; fa.ln_slots = d / 32; } fa.wqkv = b.wqkv; fa.ldw = b.kpad; fa.bqkv = b.bqkv;
      fa.wfsmn = b.wfsmn; fa.bfsmn = b.bfsmn; fa.plan = r.dp; fa.n_utts = r.batch; fa.n_heads = c.n_heads; fa.d = d;
      fa.ctx = ctx; fa.ld_ctx = d; fa.mem = mem; fa.ld_mem = d; fa.n_rows_alloc = Mpad;
      launch_sanm_qkv_attn(fa, stream);
    } else {
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
        launch_fsmn<T>(vt, Mpad, b.wfsmn, b.bfsmn, d, c.fsmn_kernel, r.dp, r.d_row_utt, Mpad, mem, d, stream);
      }
      {
        ProfScope ps(prof, "attention", stream);
        AttnArgs aa;
        aa.q = qk; aa.k = qk + d; aa.ld_qk = 2 * d; aa.vt = vt; aa.ld_vt = Mpad; aa.ctx = ctx; aa.ld_ctx = d;
        aa.plan = r.dp; aa.qb_utt = r.d_qb_utt; aa.qb_q0 = r.d_qb_q0; aa.n_qblocks = r.n_qb; aa.n_heads = c.n_heads;
        aa.qt = r.att_qt; aa.n_waves = r.att_nw; aa.max_T = r.max_T;
        if (precision == ASR_PRECISION_BF16) launch_attention_bf16_hd128(aa, stream);
        else launch_attention_f32(aa, c.d_head, stream);
      }
    }
    {
      ProfScope ps(prof, "gemm_out", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = b.wout; g.ldw = d; g.M = rows; g.N = d; g.K = d;
      g.add = mem; g.ld_add = d;                                  // FSMN memory rides in as the GEMM's additive term (:244)
      if (b.in_size == d) { g.add2 = x_in; g.ld_add2 = ld_in; }    // residual only when in/out sizes match (:246-256)
      g.out_f32 = xb; g.ld_out_f32 = d;
      if (alg) { g.out_lo = xblo; g.ld_out_lo = d; g.st_out = stb; }
      gemm(g);
    }
    if (!alg) {
      ProfScope ps(prof, "layernorm", stream);
      launch_layernorm<T>(xb, d, rows, d, b.ln2_g, b.ln2_b, 1e-5f, h, d, d, stream);
    }
    {
      ProfScope ps(prof, "gemm_ffn1", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = b.w1; g.ldw = d; g.M = rows; g.N = dff; g.K = d; g.bias = b.b1; g.act = ACT_RELU;
      g.out_lo = ffn; g.ld_out_lo = dff;
      if (alg) { g.A = xblo; g.ln_colsum = b.c1; g.ln_dim = d; g.ln_stats_in = stb; g.ln_slots = d / 32; }
      gemm(g);
    }
    {
      ProfScope ps(prof, "gemm_ffn2", stream);
      GemmArgs g;
      g.A = ffn; g.lda = dff; g.W = b.w2; g.ldw = dff; g.M = rows; g.N = d; g.K = dff; g.bias = b.b2;
      g.add = xb; g.ld_add = d; g.out_f32 = xa; g.ld_out_f32 = d;
      if (alg) { g.out_lo = xalo; g.ld_out_lo = d; g.st_out = sta; }
      gemm(g);
    }
    x_in = xa;
    x_in_lo = xalo;
    st_in = sta;
    ld_in = d;
    if (i == 0) save_tap("block0", xa, rows, d, d, 4);
    if (i == c.n_main - 1 && !paraformer) {
      ProfScope ps(prof, "layernorm", stream);
      if (alg) { launch_layernorm<bf16_t>(xa, d, rows, d, after_g, after_b, 1e-5f, xalo, d, d, stream); st_in = nullptr; }   // operand copy of the normed stream
      launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, xa, d, d, stream);
    }
  }
  if (paraformer) { copy_block_status(r); enqueue_paraformer_tail<T>(r); return; }
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
    if (taps_enabled) { g.out_f32 = d_logits.as<float>(); g.ld_out_f32 = vpad; }
    gemm(g);
  }
  {
    ProfScope ps(prof, "ctc_tail", stream);
    launch_argmax_reduce(d_amax_v.as<float>(), d_amax_i.as<int32_t>(), rows, n_slabs, d_ids.as<int32_t>(), stream);
    launch_ctc_collapse(d_ids.as<int32_t>(), r.dp, r.batch, c.blank_id, d_tok.as<int32_t>(), r.max_tokens, d_num.as<int32_t>(), stream);
  }
  if (taps_enabled) {
    save_tap("logits", d_logits.ptr, rows, c.vocab, vpad, 4);
    save_tap("frame_ids", d_ids.ptr, rows, 1, 1, 4);
  }
  // ---- outputs (pinned host staging) ---------------------------------------------------------
  HIP_CHECK(hipMemcpyAsync(h_out, d_tok.ptr, (size_t)r.batch * r.max_tokens * 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)r.batch * r.max_tokens * 4, d_num.ptr, (size_t)r.batch * 4,
                           hipMemcpyDeviceToHost, stream));
  copy_block_status(r);
}


// ---- Paraformer: CIF predictor + non-autoregressive decoder (Export_Paraformer.py:497-563) ----------------------
template <typename T>
void SvSession::enqueue_paraformer_tail(const SvRunCtx& r) {
  const auto& c = cfg;
  const int d = c.d_model, dd = pcfg.d_dec_ffn, rows = r.rows, Mpad = r.Mpad, n_slabs = vpad / 64;
  float* xa = d_xa.as<float>();
  float* enc32 = d_xb.as<float>();                 // after_norm output, f32: CIF integrals
  T* enc_lo = d_enc_lo.as<T>();                    // same in the operand dtype: GEMM input (predictor conv, cross-KV)
  T* h = d_h.as<T>();
  T* q = d_qk.as<T>();
  T* ck = d_ck.as<T>();
  T* cvt = d_vt.as<T>();
  T* ctx = d_ctx.as<T>();
  T* ffn = d_ffn.as<T>();
  float* ffn32 = d_ffn32.as<float>();
  float* x1 = d_mem.as<float>();
  float* dec = d_dec.as<float>();
  float* x2 = d_x2.as<float>();
  float* x2_own = d_sa.as<float>();                    // scratch for the un-packed CIF output (sa is not live yet)
  float* sa = d_sa.as<float>();
  UttPlan* own_plan = d_tplan.as<UttPlan>();            // token rows inside each utterance's own row range (CIF scan output)
  UttPlan* tplan = d_ctplan.as<UttPlan>();              // compact token rows: the decoder works on these
  const int32_t* m_dev = d_mdev.as<int32_t>();          // their total count, device side
  const int32_t* trow_utt = d_trow.as<int32_t>();
  {
    ProfScope ps(prof, "layernorm", stream);
    launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, enc32, d, d, stream);
    launch_layernorm<T>(xa, d, rows, d, after_g, after_b, 1e-5f, enc_lo, d, d, stream);
  }
  save_tap("enc_out", enc32, rows, d, d, 4);
  {
    ProfScope ps(prof, "cif", stream);
    launch_shift3<T>(enc_lo, d, r.dp, r.d_row_utt, Mpad, d_cifa.as<T>(), stream);     // conv k=3 (pad folded) as one GEMM
    GemmArgs g;
    g.A = d_cifa.ptr; g.lda = 3 * d; g.W = cif_conv_w; g.ldw = 3 * d; g.M = rows; g.N = d; g.K = 3 * d; g.bias = cif_conv_b; g.act = ACT_RELU;
    g.out_lo = ctx; g.ld_out_lo = d;
    gemm(g);
    launch_alpha<T>(ctx, d, cif_out_w, cif_out_b, rows, d_alpha.as<float>(), stream);
    // fired frames land in the utterance's own rows first; then they are packed (16-row aligned per utterance) so that the decoder
    // touches ~ the token count of the batch instead of every encoder row -- the count stays on the device (GemmArgs::m_dev)
    launch_cif_scan(d_alpha.as<float>(), enc32, d, r.dp, r.batch, pcfg.tail_threshold, x2_own, own_plan, d_num.as<int32_t>(), stream);
    launch_token_compact(own_plan, r.batch, Mpad, tplan, d_trow.as<int32_t>(), d_mdev.as<int32_t>(), stream);
    launch_compact_rows(x2_own, own_plan, tplan, r.batch, d, dec, stream);
  }
  save_tap("alphas", d_alpha.ptr, rows, 1, 1, 4);
  auto ffn_block = [&](const PfDecLayer& L, float* out, const float* res) {
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(dec, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream, m_dev); }
    ProfScope ps(prof, "gemm_dec", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = L.w1; g.ldw = d; g.M = rows; g.N = dd; g.K = d; g.bias = L.b1; g.act = ACT_RELU; g.out_f32 = ffn32; g.ld_out_f32 = dd; g.m_dev = m_dev;
    gemm(g);
    launch_layernorm<T>(ffn32, dd, rows, dd, nullptr, nullptr, 1e-5f, ffn, dd, dd, stream, m_dev);   // ff.norm, affine folded into w_2
    GemmArgs g2;
    g2.A = ffn; g2.lda = dd; g2.W = L.w2; g2.ldw = dd; g2.M = rows; g2.N = d; g2.K = dd; g2.bias = L.b2; g2.out_f32 = out; g2.ld_out_f32 = d; g2.m_dev = m_dev;
    if (res) { g2.add = res; g2.ld_add = d; }
    gemm(g2);
  };
  const bool kv_all = pf_kv_batch && pf_n_full > 0;
  const int ld_k = kv_all ? pf_n_full * d : d;
  if (kv_all) {                                                       // every layer's cross K (row-major, layer l at columns l d ..) and V (transposed, layer l at rows l d ..) from the memory
    ProfScope ps(prof, "gemm_dec", stream);
    GemmArgs gk;
    gk.A = enc_lo; gk.lda = d; gk.W = d_wk_all.ptr; gk.ldw = d; gk.M = rows; gk.N = pf_n_full * d; gk.K = d; gk.bias = d_bk_all.as<float>(); gk.out_lo = ck; gk.ld_out_lo = ld_k;
    gemm(gk);
    GemmArgs gv;
    gv.A = enc_lo; gv.lda = d; gv.W = d_wv_all.ptr; gv.ldw = d; gv.M = rows; gv.N = pf_n_full * d; gv.K = d; gv.bias = d_bv_all.as<float>();
    gv.out_t = cvt; gv.ld_out_t = Mpad;
    gemm(gv);
  }
  int full_i = 0;
  for (const PfDecLayer& L : pdec) {
    if (!L.full) { ffn_block(L, dec, nullptr); continue; }           // decoders3: dec = FFN(dec), no residual (:553-555)
    T* ck_l = kv_all ? ck + (size_t)full_i * d : ck;
    T* cvt_l = kv_all ? cvt + (size_t)full_i * d * Mpad : cvt;
    ++full_i;
    if (!kv_all) {
      ProfScope ps(prof, "gemm_dec", stream);                        // this layer's cross K (row-major) and V (transposed) from the memory
      GemmArgs gk;
      gk.A = enc_lo; gk.lda = d; gk.W = L.wkv; gk.ldw = d; gk.M = rows; gk.N = d; gk.K = d; gk.bias = L.bkv; gk.out_lo = ck; gk.ld_out_lo = d;
      gemm(gk);
      GemmArgs gv;
      gv.A = enc_lo; gv.lda = d; gv.W = (const T*)L.wkv + (size_t)d * d; gv.ldw = d; gv.M = rows; gv.N = d; gv.K = d; gv.bias = L.bkv + d;
      gv.out_t = cvt; gv.ld_out_t = Mpad;
      gemm(gv);
    }
    ffn_block(L, x1, nullptr);                                       // x = w_2(norm(relu(w_1(norm1(dec)))))
    {
      ProfScope ps(prof, "fsmn", stream);
      launch_layernorm<float>(x1, d, rows, d, L.n2_g, L.n2_b, 1e-5f, sa, d, d, stream, m_dev);
      launch_fsmn_rows(sa, dec, L.wfsmn, d, c.fsmn_kernel, tplan, trow_utt, Mpad, x2, stream, m_dev);   // x = dec + fsmn(norm2(x))
    }
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(x2, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream, m_dev); }
    {
      ProfScope ps(prof, "gemm_dec", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = L.wq; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.bias = L.bq; g.out_lo = q; g.ld_out_lo = d; g.m_dev = m_dev;
      gemm(g);
    }
    {
      ProfScope ps(prof, "attention", stream);
      AttnArgs aa;
      aa.q = q; aa.ld_q = d; aa.k = ck_l; aa.ld_qk = ld_k; aa.vt = cvt_l; aa.ld_vt = Mpad; aa.ctx = ctx; aa.ld_ctx = d;
      aa.plan = r.dp; aa.q_plan = tplan; aa.qb_utt = r.d_qb_utt; aa.qb_q0 = r.d_qb_q0; aa.n_qblocks = r.n_qb; aa.n_heads = c.n_heads;
      aa.qt = r.att_qt; aa.n_waves = r.att_nw; aa.max_T = r.max_T;
      if (precision == ASR_PRECISION_BF16) launch_attention_bf16_hd128(aa, stream);
      else launch_attention_f32(aa, c.d_head, stream);
    }
    {
      ProfScope ps(prof, "gemm_dec", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = L.wo; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.bias = L.bo; g.add = x2; g.ld_add = d; g.out_f32 = dec; g.ld_out_f32 = d; g.m_dev = m_dev;
      gemm(g);
    }
  }
  { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(dec, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream, m_dev); }
  {
    ProfScope ps(prof, "gemm_out", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = pf_out_w; g.ldw = d; g.M = rows; g.N = vpad; g.K = d; g.bias = pf_out_b;
    g.amax_val = d_amax_v.as<float>(); g.amax_idx = d_amax_i.as<int32_t>(); g.n_valid = c.vocab; g.m_dev = m_dev;
    if (taps_enabled) { g.out_f32 = d_logits.as<float>(); g.ld_out_f32 = vpad; }
    gemm(g);
    launch_argmax_reduce(d_amax_v.as<float>(), d_amax_i.as<int32_t>(), rows, n_slabs, d_ids.as<int32_t>(), stream);
    launch_gather_tokens(d_ids.as<int32_t>(), tplan, r.batch, d_tok.as<int32_t>(), r.max_tokens, stream);
  }
  if (taps_enabled) save_tap("logits", d_logits.ptr, rows, c.vocab, vpad, 4);
  HIP_CHECK(hipMemcpyAsync(h_out, d_tok.ptr, (size_t)r.batch * r.max_tokens * 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)r.batch * r.max_tokens * 4, d_num.ptr, (size_t)r.batch * 4,
                           hipMemcpyDeviceToHost, stream));
}

template <typename T>
void SvSession::run(const float* audio, int audio_mem, const int64_t* offs, int batch, const int32_t* lang, int32_t* tok_out,
                    int max_tokens, int32_t* num_out) {
  const auto& c = cfg;
  ASR_REQUIRE(batch > 0, "sensevoice: empty batch");
  ASR_REQUIRE(audio && offs && (lang || paraformer) && tok_out && num_out, "sensevoice: null argument");
  HIP_CHECK(hipSetDevice(device));
  // no cluster kernel beside a foreign (RCCL) section: this whole call -- it returns with the stream drained -- is one cluster pass, or takes the four-launch path
  ClusterScope cluster(device);
  foreign_now = !cluster.ok;
  if (foreign_now) ++foreign_diverted;
  const int d = c.d_model, dff = c.d_ffn;

  // ---- host plan -------------------------------------------------------------------------
  std::vector<UttPlan> plan(batch);
  SvRunCtx r{};
  int rows = 0, frames = 0, n_fb = 0, n_qb = 0, max_T = 0;
  const int64_t base0 = offs[0];
  uint64_t key = 1469598103934665603ull;                 // FNV-1a over everything that shapes the launch sequence
  auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
  for (int b = 0; b < batch; ++b) {
    const int64_t n = offs[b + 1] - offs[b];
    ASR_REQUIRE(n >= c.win_length, "sensevoice: utterance %d has %lld samples (< one %d-sample frame)", b, (long long)n, c.win_length);
    ASR_REQUIRE(n <= c.max_audio_len, "sensevoice: utterance %d has %lld samples (> max_audio_len %d)", b, (long long)n, c.max_audio_len);
    ASR_REQUIRE(paraformer || (lang[b] >= 0 && lang[b] < c.n_languages), "sensevoice: language_idx %d out of range", lang[b]);
    UttPlan& p = plan[b];
    p.audio_off = offs[b] - base0;
    p.n_samples = (int)n;
    p.n_frames = ((int)n - c.win_length) / c.hop_length + 1;
    p.frame_off = frames;
    p.n_lfr = (p.n_frames + c.lfr_n - 1) / c.lfr_n;
    p.T = p.n_lfr + c.n_prompt;
    p.row_off = rows;
    p.lang = lang ? lang[b] : 0;
    p.blk0 = n_fb;
    frames += p.n_frames;
    rows += round_up(p.T, 16);
    n_fb += (p.n_frames + 63) / 64;
    max_T = std::max(max_T, p.T);
    mix((uint64_t)n);
  }
  ASR_REQUIRE(max_tokens >= 1, "sensevoice: max_tokens must be positive");
  int att_qt = 0, att_nw = 4, q_rows = 64;             // f32 kernel: fixed 64-row query blocks
  if (precision == ASR_PRECISION_BF16) { attention_geometry(max_T, c.d_head, &att_qt, &att_nw); q_rows = 16 * att_qt * att_nw; }
  for (int b = 0; b < batch; ++b) n_qb += (plan[b].T + q_rows - 1) / q_rows;
  const int Mpad = round_up(rows, 128);
  const int64_t total_samples = offs[batch] - base0;

  // plan blob: [UttPlan x B][blk_utt n_fb][blk_f0 n_fb][qb_utt n_qb][qb_q0 n_qb][row_utt Mpad]
  const int n_tiles = rows / 16;
  const size_t plan_bytes = sizeof(UttPlan) * batch + sizeof(int32_t) * (2 * (size_t)n_fb + 2 * (size_t)n_qb + Mpad + 2 * (size_t)n_tiles);
  if (plan_bytes > h_plan_cap) {
    if (h_plan) HIP_CHECK(hipHostFree(h_plan));
    HIP_CHECK(hipHostMalloc(&h_plan, plan_bytes * 2, hipHostMallocDefault));
    h_plan_cap = plan_bytes * 2;
    ++ws_epoch;
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
      for (int q0 = 0; q0 < plan[b].T; q0 += q_rows) { qb_utt[qi] = b; qb_q0[qi++] = q0; }
      const int r16 = round_up(plan[b].T, 16);
      for (int t = 0; t < r16; ++t) row_utt[plan[b].row_off + t] = b;
    }
    for (int t = rows; t < Mpad; ++t) row_utt[t] = -1;
    int32_t* tile_win = row_utt + Mpad;
    int32_t* tile_idx = tile_win + n_tiles;
    int ti = 0;
    for (int b = 0; b < batch; ++b)
      for (int t = 0; t < round_up(plan[b].T, 16) / 16; ++t) { tile_win[ti] = b; tile_idx[ti++] = t; }
  }
  // small batches take the tile kernel (bf16, LayerNorm-folded arenas of the standard geometry, every workgroup resident at once)
  const bool tiles = sizeof(T) == 2 && use_tiles && c.n_blocks > 1 && blocks[c.n_blocks - 1].cqkv && blocks[c.n_blocks - 1].c1 &&
                     !(use_block && batch >= block_min_utts) && n_tiles <= sanm_tiles_max_tiles() &&
                     sanm_tiles_supported(max_T, d, dff, c.n_heads, c.d_head, c.fsmn_kernel);
  mix((uint64_t)tiles);
  // ---- workspace (grow-only; any re-allocation invalidates the captured graph) -----------------
  const size_t eT = sizeof(T);
  auto grow = [&](DeviceBuffer& buf, size_t bytes) { void* before = buf.ptr; buf.reserve(bytes, stream); if (buf.ptr != before) ++ws_epoch; };
  grow(d_plan, plan_bytes);
  if (audio_mem == ASR_MEM_HOST) grow(d_audio, (size_t)total_samples * 4);
  grow(d_mel, (size_t)frames * c.n_mels * 4);
  grow(d_x0, (size_t)Mpad * kpad0 * 4);
  grow(d_xa, (size_t)Mpad * d * 4);
  grow(d_xb, (size_t)Mpad * d * 4);
  grow(d_h, (size_t)Mpad * std::max(kpad0, d) * eT);
  if (precision == ASR_PRECISION_BF16) {
    grow(d_x0lo, (size_t)Mpad * kpad0 * 2);
    grow(d_xalo, (size_t)Mpad * d * 2);
    grow(d_xblo, (size_t)Mpad * d * 2);
    grow(d_sta, (size_t)Mpad * (d / 32) * 8);
    grow(d_stb, (size_t)Mpad * (d / 32) * 8);
  }
  grow(d_qk, (size_t)Mpad * 2 * d * eT);
  grow(d_vt, (size_t)Mpad * d * eT);
  grow(d_ctx, (size_t)Mpad * d * eT);
  grow(d_mem, (size_t)Mpad * d * 4);
  grow(d_ffn, (size_t)Mpad * dff * eT);
  const int n_slabs = vpad / 64;
  grow(d_amax_v, (size_t)Mpad * n_slabs * 4);
  grow(d_amax_i, (size_t)Mpad * n_slabs * 4);
  grow(d_ids, (size_t)Mpad * 4);
  grow(d_flags, ((size_t)c.n_blocks * batch * 5 + 4 + (tiles ? (size_t)c.n_blocks * (n_tiles + batch) * 4 : 0)) * 4);
  if (tiles) grow(d_tkv, (size_t)Mpad * 2 * d * 2 * 2);          // [block][window][4] exchange counters + error word (4) + [block][window] placement words
  grow(d_tok, (size_t)batch * max_tokens * 4);
  grow(d_num, (size_t)batch * 4);
  if (taps_enabled) grow(d_logits, (size_t)Mpad * vpad * 4);
  if (paraformer) {
    const int dd = pcfg.d_dec_ffn;
    grow(d_enc_lo, (size_t)Mpad * d * eT);
    grow(d_ck, (size_t)Mpad * d * eT * (pf_kv_batch ? std::max(pf_n_full, 1) : 1));
    if (pf_kv_batch) grow(d_vt, (size_t)Mpad * d * eT * std::max(pf_n_full, 1));
    grow(d_cifa, (size_t)Mpad * 3 * d * eT);
    grow(d_alpha, (size_t)Mpad * 4);
    grow(d_dec, (size_t)Mpad * d * 4);
    grow(d_x2, (size_t)Mpad * d * 4);
    grow(d_sa, (size_t)Mpad * d * 4);
    grow(d_ffn32, (size_t)Mpad * dd * 4);
    grow(d_ffn, (size_t)Mpad * std::max(dd, dff) * eT);
    grow(d_tplan, sizeof(UttPlan) * batch);
    grow(d_ctplan, sizeof(UttPlan) * batch);
    grow(d_mdev, 256);
    grow(d_trow, (size_t)Mpad * 4);
  }
  const size_t out_bytes = (size_t)batch * max_tokens * 4 + (size_t)batch * 4 + 16;
  if (out_bytes > h_out_cap) {
    if (h_out) HIP_CHECK(hipHostFree(h_out));
    HIP_CHECK(hipHostMalloc(&h_out, out_bytes * 2, hipHostMallocDefault));
    h_out_cap = out_bytes * 2;
    ++ws_epoch;
  }

  HIP_CHECK(hipMemcpyAsync(d_plan.ptr, h_plan, plan_bytes, hipMemcpyHostToDevice, stream));
  if (audio_mem == ASR_MEM_HOST) {
    HIP_CHECK(hipMemcpyAsync(d_audio.ptr, audio + base0, (size_t)total_samples * 4, hipMemcpyHostToDevice, stream));
    r.d_aud = d_audio.as<float>();
  } else {
    r.d_aud = audio + base0;
  }
  r.batch = batch; r.rows = rows; r.Mpad = Mpad; r.frames = frames; r.n_fb = n_fb; r.n_qb = n_qb; r.max_T = max_T;
  r.max_tokens = max_tokens; r.att_qt = att_qt; r.att_nw = att_nw;
  r.dp = d_plan.as<UttPlan>();
  r.n_tiles = n_tiles; r.tiles = tiles;
  r.d_blk_utt = (const int32_t*)((unsigned char*)d_plan.ptr + sizeof(UttPlan) * batch);
  r.d_blk_f0 = r.d_blk_utt + n_fb;
  r.d_qb_utt = r.d_blk_f0 + n_fb;
  r.d_qb_q0 = r.d_qb_utt + n_qb;
  r.d_row_utt = r.d_qb_q0 + n_qb;
  r.d_tile_win = r.d_row_utt + Mpad;
  r.d_tile_idx = r.d_tile_win + n_tiles;
  mix((uint64_t)batch); mix((uint64_t)max_tokens); mix((uint64_t)(uintptr_t)r.d_aud); mix(ws_epoch); mix((uint64_t)(uintptr_t)stream);
  mix((uint64_t)(block_cooldown > 0 || foreign_now));        // a session cooling down after a cluster give-up replays the four-launch capture, not the block one

  if (sizeof(T) == 2 && use_block && cfg.n_blocks > 1 && blocks[cfg.n_blocks - 1].cqkv && blocks[cfg.n_blocks - 1].c1 && batch >= block_min_utts &&
      sanm_block_supported(max_T, cfg.d_head, cfg.n_heads, cfg.d_model, cfg.d_ffn, cfg.fsmn_kernel))
    ensure_block_pack();
  if (tiles) ensure_tiles_pack();
  // ---- launch: eager the first time a geometry is seen (allocations settle), then capture once and replay ----
  // 570 launches per forward are host-launch-bound when issued eagerly (~13 us each); replay costs ~1 us per node.
  const bool graphable = use_graph && !taps_enabled && !prof.enabled;
  if (graphable && graph_exec && key == graph_key) {
    HIP_CHECK(hipGraphLaunch(graph_exec, stream));
  } else if (graphable && key == eager_key) {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    try {
      enqueue<T>(r);
    } catch (...) {
      (void)hipStreamEndCapture(stream, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    HIP_CHECK(hipStreamEndCapture(stream, &graph));
    HIP_CHECK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    graph_key = key;
    HIP_CHECK(hipGraphLaunch(graph_exec, stream));
  } else {
    enqueue<T>(r);
    eager_key = key;
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  if (prof.enabled) prof.collect();
  if (tiles_dbg >= 0 && tiles && d_times.ptr) {          // tuning: mean phase intervals of one block of the tile kernel over its workgroups (100 MHz clock -> us)
    std::vector<unsigned long long> t(256 * 16);
    HIP_CHECK(hipMemcpy(t.data(), d_times.ptr, t.size() * 8, hipMemcpyDeviceToHost));
    double sum[14] = {}; int cnt = 0;
    for (int w = 0; w < 256; ++w) {
      if (!t[(size_t)w * 16]) continue;
      ++cnt;
      for (int k = 1; k <= 13; ++k) sum[k] += (double)(t[(size_t)w * 16 + k] - t[(size_t)w * 16 + k - 1]) * 0.01;
    }
    fprintf(stderr, "sanm_tiles block %d (us, mean of %d workgroups):", tiles_dbg, cnt);
    for (int k = 1; k <= 13; ++k) fprintf(stderr, " %.2f", sum[k] / std::max(cnt, 1));
    fprintf(stderr, "\n");
  }
  if (block_dbg >= 0 && d_times.ptr) {
    std::vector<unsigned long long> t(256 * 16);
    HIP_CHECK(hipMemcpy(t.data(), d_times.ptr, t.size() * 8, hipMemcpyDeviceToHost));
    static const char* names8[14] = {"A loop (4 chunks)", "stats+images+attention", "ctx store+publish", "fsmn", "B prologue+wait 0", "B loop", "B epilogue+publish", "C wait 1+stats",
                                     "C loop", "C epilogue (hid)", "publish 2", "D own chunk+wait 2", "D loop (rest)", "D epilogue"};
    const int nk = 14;
    const char* const* names = names8;
    unsigned long long t_first = ~0ull, t_last = 0;
    int n = 0;
    double sum[14] = {0}, mx[14] = {0};
    for (int w = 0; w < 256; ++w) {
      if (!t[w * 16]) continue;
      ++n;
      t_first = std::min(t_first, t[w * 16]); t_last = std::max(t_last, t[w * 16 + nk]);
      for (int k = 0; k < nk; ++k) { const double us = (double)(t[w * 16 + k + 1] - t[w * 16 + k]) * 0.01; sum[k] += us; mx[k] = std::max(mx[k], us); }
    }
    if (n) {
      fprintf(stderr, "[sanm_block %d] %d workgroups, first start -> last end %.1f us\n", block_dbg, n, (double)(t_last - t_first) * 0.01);
      for (int k = 0; k < nk; ++k) fprintf(stderr, "  %-22s avg %6.2f us  max %6.2f us\n", names[k], sum[k] / n, mx[k]);
      if (block8_opt & 2048) {            // stamps 4..7 were taken inside phase A (the rows "B prologue .. B epilogue" above are then meaningless)
        static const char* inside[5] = {"A: statistics", "A: q/k/v images", "A: own attention tile", "A: shared 9th tile", "A: closing barrier"};
        const int from[5] = {1, 4, 5, 6, 7}, to[5] = {4, 5, 6, 7, 2};
        for (int q = 0; q < 5; ++q) {
          double sm = 0.0, m2 = 0.0;
          for (int w = 0; w < 256; ++w) if (t[w * 16]) { const double us = (double)(t[w * 16 + to[q]] - t[w * 16 + from[q]]) * 0.01; sm += us; m2 = std::max(m2, us); }
          fprintf(stderr, "  %-22s avg %6.2f us  max %6.2f us\n", inside[q], sm / n, m2);
        }
      }
      int n_plain = 0;
      for (int w = 0; w < 256; ++w) if (t[w * 16]) n_plain += (int)t[w * 16 + 15];
      fprintf(stderr, "  (debug word 15: %d of %d workgroups)\n", n_plain, n);
    }
  }
  {
    // The four workgroups of a cluster must be resident together. That holds when the launch has the chip to itself; when another stream's
    // kernels hold CUs (several sessions in flight) a cluster can be split across dispatch waves and, in the worst case, wait in a circle
    // with another session's clusters until the bounded spin gives up. Such a forward pass is void: it is redone here on the four-launch
    // path (no cross-workgroup waits), still on the GPU; the error is raised only if that fails too.
    unsigned blk_err = 0;
    memcpy(&blk_err, (unsigned char*)h_out + (size_t)batch * max_tokens * 4 + (size_t)batch * 4, 4);
    if (blk_err != 0 && (use_block || use_tiles)) {
      if (block_giveups++ == 0)
        fprintf(stderr, "[asr_mi355x] sanm_block: a workgroup gave up waiting for its cluster; the batch is redone on the four-launch path\n");
      block_cooldown = 16;                       // this batch and the next ones stay on the four-launch path
      enqueue<T>(r);
      HIP_CHECK(hipStreamSynchronize(stream));
      memcpy(&blk_err, (unsigned char*)h_out + (size_t)batch * max_tokens * 4 + (size_t)batch * 4, 4);
    }
    else if (block_cooldown > 0) --block_cooldown;
    ASR_REQUIRE(blk_err == 0, "sensevoice: a SANM block workgroup gave up waiting for its cluster (results are invalid)");
  }
  memcpy(num_out, (unsigned char*)h_out + (size_t)batch * max_tokens * 4, (size_t)batch * 4);
  const int32_t* ht = (const int32_t*)h_out;
  for (int b = 0; b < batch; ++b) {
    const int n = std::min(num_out[b], max_tokens);
    memcpy(tok_out + (size_t)b * max_tokens, ht + (size_t)b * max_tokens, (size_t)n * 4);
  }
}

// ---- streaming Paraformer (Paraformer/Streaming/Export_Paraformer_Streaming.py:386-553; host loop Inference_..._Streaming_ONNX.py:401-449)
void SvSession::stream_init(int chunk, int look_back_encoder, int look_back_decoder, int max_streams) {
  const auto& c = cfg;
  ASR_REQUIRE(paraformer, "streaming: needs a Paraformer session");
  ASR_REQUIRE(chunk >= c.win_length && max_streams >= 1 && look_back_encoder >= 1 && look_back_decoder >= 1, "streaming: bad geometry");
  st_chunk = chunk;
  st_frames = (chunk - c.win_length) / c.hop_length + 1;
  st_B = ((c.lfr_m - 1) / 2 + st_frames) / c.lfr_n + 1;
  st_C = st_B / 2;
  st_en_cap = look_back_encoder * st_B;
  st_de_cap = look_back_decoder * st_B;
  st_max = max_streams;
  ASR_REQUIRE(st_B + st_C <= 16 && st_frames <= 64 && st_en_cap + st_B + st_C <= 64 && st_de_cap + st_B + st_C <= 64 && c.d_head == 128,
              "streaming: chunk of %d samples gives %d + %d rows per step (16-row slots, 64-key attention)", chunk, st_C, st_B);
  const size_t T = precision == ASR_PRECISION_BF16 ? 2 : 4;
  const size_t head_row = (size_t)c.n_heads * 128;
  st_enk.reserve((size_t)c.n_blocks * max_streams * st_en_cap * head_row * T, stream);
  st_env.reserve((size_t)c.n_blocks * max_streams * st_en_cap * head_row * T, stream);
  st_dek.reserve((size_t)pcfg.n_dec * max_streams * st_de_cap * head_row * T, stream);
  st_dev.reserve((size_t)pcfg.n_dec * max_streams * st_de_cap * head_row * T, stream);
  st_defsmn.reserve((size_t)pcfg.n_dec * max_streams * (c.fsmn_kernel - 1) * c.d_model * 4, stream);
  st_prev.reserve((size_t)max_streams * st_C * kpad0 * 4, stream);
  st_cifh.reserve((size_t)max_streams * c.d_model * 4, stream);
  st_cifa.reserve((size_t)max_streams * 4, stream);
  st_enlen.reserve((size_t)max_streams * 4, stream);
  st_delen.reserve((size_t)max_streams * 4, stream);
  st_start.reserve((size_t)max_streams * 4, stream);
  // bf16 sessions of the standard geometry: layers 1 .. n - 1 of a chunk step run as one launch (layer 0 has the 560-wide input and no residual)
  if (const char* e = getenv("ASR_STREAM_FUSED")) st_fused_env = atoi(e);
  if (const char* e = getenv("ASR_STREAM_TIMES")) st_times_layer = atoi(e);
  if (const char* e = getenv("ASR_STREAM_OPT")) st_opt = atoi(e);
  st_fused_max = gemm_env_cus() * 3 / 4;               // 192 streams on 256 CUs: tools/probes/stream_fused_sweep.sh (profiles/r05_stream_fused_sweep.txt) -- the cluster launches step by CU-loads of 64 streams
                                                       // (2.5 / 5.2 / 7.8 / 10.4 ms), the per-launch path grows smoothly (5.3 ms at 64, 8.1 at 192, 8.9 at 256) and is ahead from 193 on
  if (const char* e = getenv("ASR_STREAM_FUSED_MAX")) st_fused_max = atoi(e);
  if (const char* e = getenv("ASR_STREAM_SHARE")) st_share_rule = atoi(e);
  if (const char* e = getenv("ASR_STREAM_SNAPSHOT")) st_snapshot_env = atoi(e);
  if (const char* e = getenv("ASR_STREAM_FAULT")) st_fault = atoi(e);
  st_fused = st_fused_env != 0 && precision == ASR_PRECISION_BF16 && c.n_blocks > 1 &&
             stream_layers_supported(c.d_model, c.d_ffn, c.n_heads, st_en_cap, st_B + st_C, c.fsmn_kernel) && blocks[1].kpad == c.d_model;
  if (st_fused) {
    const int nl = c.n_blocks - 1;
    const size_t pk = stream_layers_pack_bytes(), en_layer = (size_t)st_max * c.n_heads * st_en_cap * 128;
    st_wpack.reserve(pk * nl, stream);
    st_layer_tab.reserve(sizeof(StreamLayer) * nl, stream);
    st_flags.reserve(((size_t)nl * st_max * 4 + 4 + (size_t)pdec.size() * st_max * 8) * 4, stream);       // encoder counters, err, decoder counters
    std::vector<StreamLayer> tab(nl);
    for (int i = 0; i < nl; ++i) {
      const SvBlock& b = blocks[i + 1];
      unsigned char* dst = (unsigned char*)st_wpack.ptr + pk * i;
      launch_stream_layers_pack((const bf16_t*)b.wqkv, (const bf16_t*)b.wout, (const bf16_t*)b.w1, (const bf16_t*)b.w2, dst, stream);
      tab[i].wpack = dst; tab[i].bqkv = b.bqkv; tab[i].wfsmn = b.wfsmn; tab[i].bfsmn = b.bfsmn; tab[i].b1 = b.b1; tab[i].b2 = b.b2;
      tab[i].cache_k = st_enk.as<bf16_t>() + (size_t)(i + 1) * en_layer;
      tab[i].cache_v = st_env.as<bf16_t>() + (size_t)(i + 1) * en_layer;
    }
    HIP_CHECK(hipMemcpyAsync(st_layer_tab.ptr, tab.data(), sizeof(StreamLayer) * nl, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  // ... and so does the decoder (every block of it)
  st_dec_fused = st_fused && st_fused_env >= 2 && !pdec.empty() &&
                 stream_dec_supported(c.d_model, pcfg.d_dec_ffn, c.n_heads, st_de_cap, st_B + st_C, c.fsmn_kernel);
  if (st_dec_fused) {
    const int ndl = (int)pdec.size();
    const size_t pk = stream_dec_pack_bytes(), de_layer = (size_t)st_max * c.n_heads * st_de_cap * 128, fs_layer = (size_t)st_max * (c.fsmn_kernel - 1) * c.d_model;
    st_dpack.reserve(pk * ndl, stream);
    st_dlayer_tab.reserve(sizeof(StreamDecLayer) * ndl, stream);
    std::vector<StreamDecLayer> tab(ndl);
    int li = 0;
    for (int j = 0; j < ndl; ++j) {
      const PfDecLayer& L = pdec[j];
      unsigned char* dst = (unsigned char*)st_dpack.ptr + pk * j;
      launch_stream_dec_pack((const bf16_t*)L.w1, (const bf16_t*)L.w2, (const bf16_t*)L.wq, (const bf16_t*)L.wkv, (const bf16_t*)L.wo, L.full, dst, stream);
      StreamDecLayer& t = tab[j];
      t = StreamDecLayer{};
      t.wpack = dst; t.b1 = L.b1; t.b2 = L.b2; t.full = L.full ? 1 : 0;
      if (L.full) {
        t.n2_g = L.n2_g; t.n2_b = L.n2_b; t.wfsmn = L.wfsmn; t.bq = L.bq; t.bkv = L.bkv; t.bo = L.bo;
        t.fsmn_hist = st_defsmn.as<float>() + (size_t)li * fs_layer;
        t.cache_k = st_dek.as<bf16_t>() + (size_t)li * de_layer;
        t.cache_v = st_dev.as<bf16_t>() + (size_t)li * de_layer;
        ++li;
      }
    }
    HIP_CHECK(hipMemcpyAsync(st_dlayer_tab.ptr, tab.data(), sizeof(StreamDecLayer) * ndl, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  stream_reset(-1);
}

void SvSession::ensure_stream_shadow() {
  if (st_n_segs) return;
  const auto& c = cfg;
  const size_t T = precision == ASR_PRECISION_BF16 ? 2 : 4, head_row = (size_t)c.n_heads * 128;
  struct Spec { DeviceBuffer* buf; size_t per_stream; int n_outer; };
  const Spec specs[] = {
      {&st_enk, st_en_cap * head_row * T, c.n_blocks}, {&st_env, st_en_cap * head_row * T, c.n_blocks},
      {&st_dek, st_de_cap * head_row * T, pcfg.n_dec}, {&st_dev, st_de_cap * head_row * T, pcfg.n_dec},
      {&st_defsmn, (size_t)(c.fsmn_kernel - 1) * c.d_model * 4, pcfg.n_dec}, {&st_prev, (size_t)st_C * kpad0 * 4, 1},
      {&st_cifh, (size_t)c.d_model * 4, 1}, {&st_cifa, 4, 1}, {&st_enlen, 4, 1}, {&st_delen, 4, 1}, {&st_start, 4, 1}};
  size_t total = 0;
  for (const Spec& sp : specs) total += (sp.per_stream * st_max * sp.n_outer + 255) / 256 * 256;
  st_shadow.reserve(total, stream);
  std::vector<StreamStateSeg> segs;
  size_t off = 0;
  int item = 0;
  for (const Spec& sp : specs) {
    if (sp.per_stream == 0 || sp.n_outer == 0) continue;
    StreamStateSeg g;
    g.live = (unsigned char*)sp.buf->ptr; g.shadow = (unsigned char*)st_shadow.ptr + off;
    g.per_stream = sp.per_stream; g.outer_stride = sp.per_stream * st_max; g.n_outer = sp.n_outer; g.first_item = item;
    segs.push_back(g);
    item += sp.n_outer;
    off += (sp.per_stream * st_max * sp.n_outer + 255) / 256 * 256;
  }
  st_segs.reserve(sizeof(StreamStateSeg) * segs.size(), stream);
  HIP_CHECK(hipMemcpyAsync(st_segs.ptr, segs.data(), sizeof(StreamStateSeg) * segs.size(), hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  st_n_segs = (int)segs.size(); st_n_items = item;
}

void SvSession::stream_reset(int sid) {
  const auto& c = cfg;
  ASR_REQUIRE(st_max > 0 && sid >= -1 && sid < st_max, "streaming: stream id %d out of range", sid);
  HIP_CHECK(hipSetDevice(device));
  const int lo = sid < 0 ? 0 : sid, cnt = sid < 0 ? st_max : 1;
  // K/V histories are guarded by their length counters; the additive state must really be zero
  HIP_CHECK(hipMemsetAsync(st_enlen.as<int32_t>() + lo, 0, (size_t)cnt * 4, stream));
  HIP_CHECK(hipMemsetAsync(st_delen.as<int32_t>() + lo, 0, (size_t)cnt * 4, stream));
  HIP_CHECK(hipMemsetAsync(st_start.as<int32_t>() + lo, 0, (size_t)cnt * 4, stream));
  HIP_CHECK(hipMemsetAsync(st_cifa.as<float>() + lo, 0, (size_t)cnt * 4, stream));
  HIP_CHECK(hipMemsetAsync(st_cifh.as<float>() + (size_t)lo * c.d_model, 0, (size_t)cnt * c.d_model * 4, stream));
  HIP_CHECK(hipMemsetAsync(st_prev.as<float>() + (size_t)lo * st_C * kpad0, 0, (size_t)cnt * st_C * kpad0 * 4, stream));
  const size_t hist = (size_t)(c.fsmn_kernel - 1) * c.d_model;
  for (int l = 0; l < pcfg.n_dec; ++l)
    HIP_CHECK(hipMemsetAsync(st_defsmn.as<float>() + ((size_t)l * st_max + lo) * hist, 0, (size_t)cnt * hist * 4, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
}

template <typename T>
void SvSession::stream_step(const float* audio, int audio_mem, const int32_t* stream_ids, int n, int32_t* tok_out, int max_tokens,
                            int32_t* num_out) {
  const auto& c = cfg;
  ASR_REQUIRE(st_max > 0, "streaming: session was not created with asr_paraformer_stream_create");
  ASR_REQUIRE(audio && stream_ids && tok_out && num_out && n >= 1 && n <= st_max && max_tokens >= st_B + 1, "streaming: bad argument (n = %d)", n);
  HIP_CHECK(hipSetDevice(device));
  ClusterScope cluster(device);                          // (as in run(): fused = cluster kernels; a foreign section open on this GPU sends the step down the per-launch path)
  foreign_now = !cluster.ok;
  if (foreign_now) ++foreign_diverted;
  const int d = c.d_model, dff = c.d_ffn, dd = pcfg.d_dec_ffn, H = c.n_heads, n_cur = st_B + st_C;
  const int rows = n * 16, Mpad = round_up(rows, 128), frames = n * st_frames, n_slabs = vpad / 64;
  std::vector<char> seen(st_max, 0);
  // ---- plan: one 16-row slot / one fbank workgroup per active stream
  const size_t plan_bytes = sizeof(UttPlan) * n + sizeof(int32_t) * (2 * (size_t)n + Mpad);
  if (plan_bytes > h_plan_cap) {
    if (h_plan) HIP_CHECK(hipHostFree(h_plan));
    HIP_CHECK(hipHostMalloc(&h_plan, plan_bytes * 2, hipHostMallocDefault));
    h_plan_cap = plan_bytes * 2;
  }
  UttPlan* hp = (UttPlan*)h_plan;
  int32_t* blk_utt = (int32_t*)((unsigned char*)h_plan + sizeof(UttPlan) * n);
  int32_t* blk_f0 = blk_utt + n;
  int32_t* row_utt = blk_f0 + n;
  for (int i = 0; i < n; ++i) {
    ASR_REQUIRE(stream_ids[i] >= 0 && stream_ids[i] < st_max && !seen[stream_ids[i]], "streaming: stream id %d invalid or repeated", stream_ids[i]);
    seen[stream_ids[i]] = 1;
    UttPlan& p = hp[i];
    p.audio_off = (int64_t)i * st_chunk; p.n_samples = st_chunk; p.n_frames = st_frames; p.frame_off = i * st_frames;
    p.n_lfr = st_B; p.T = n_cur; p.row_off = i * 16; p.lang = stream_ids[i]; p.blk0 = i;
    blk_utt[i] = i; blk_f0[i] = 0;
    for (int t = 0; t < 16; ++t) row_utt[i * 16 + t] = i;
  }
  for (int t = rows; t < Mpad; ++t) row_utt[t] = -1;
  const size_t eT = sizeof(T);
  auto grow = [&](DeviceBuffer& buf, size_t bytes) { buf.reserve(bytes, stream); };
  grow(d_plan, plan_bytes);
  if (audio_mem == ASR_MEM_HOST) grow(d_audio, (size_t)n * st_chunk * 4);
  grow(d_mel, (size_t)frames * c.n_mels * 4);
  grow(d_x0, (size_t)Mpad * kpad0 * 4);
  grow(d_xa, (size_t)Mpad * d * 4);
  grow(d_xb, (size_t)Mpad * d * 4);
  grow(d_h, (size_t)Mpad * std::max(kpad0, d) * eT);
  grow(d_sqkv, (size_t)Mpad * 3 * d * eT);
  grow(d_skv, (size_t)Mpad * 2 * d * eT);
  grow(d_qk, (size_t)Mpad * 2 * d * eT);
  grow(d_ctx, (size_t)Mpad * d * eT);
  grow(d_mem, (size_t)Mpad * d * 4);
  grow(d_ffn, (size_t)Mpad * std::max(dd, dff) * eT);
  grow(d_amax_v, (size_t)Mpad * n_slabs * 4);
  grow(d_amax_i, (size_t)Mpad * n_slabs * 4);
  grow(d_ids, (size_t)Mpad * 4);
  grow(d_tok, (size_t)n * max_tokens * 4);
  grow(d_num, (size_t)n * 4);
  grow(d_logits, (size_t)Mpad * vpad * 4);
  grow(d_enc_lo, (size_t)Mpad * d * eT);
  grow(d_cifa, (size_t)Mpad * 3 * d * eT);
  grow(d_alpha, (size_t)Mpad * 4);
  grow(d_dec, (size_t)Mpad * d * 4);
  grow(d_x2, (size_t)Mpad * d * 4);
  grow(d_sa, (size_t)Mpad * d * 4);
  grow(d_ffn32, (size_t)Mpad * dd * 4);
  grow(d_tplan, sizeof(UttPlan) * n);
  if (precision == ASR_PRECISION_BF16) {                // LayerNorm inside the per-launch layers' projections (see the layer loop): bf16 copies of the residual stream + their row statistics
    grow(d_xblo, (size_t)Mpad * d * 2);
    grow(d_stb, (size_t)Mpad * (d / 32) * 8);
  }
  const size_t out_bytes = (size_t)n * max_tokens * 4 + (size_t)n * 4;
  if (out_bytes + 16 > h_out_cap) {
    if (h_out) HIP_CHECK(hipHostFree(h_out));
    HIP_CHECK(hipHostMalloc(&h_out, out_bytes * 2 + 16, hipHostMallocDefault));
    h_out_cap = out_bytes * 2 + 16;
  }
  HIP_CHECK(hipMemcpyAsync(d_plan.ptr, h_plan, plan_bytes, hipMemcpyHostToDevice, stream));
  const float* d_aud = audio;
  if (audio_mem == ASR_MEM_HOST) {
    HIP_CHECK(hipMemcpyAsync(d_audio.ptr, audio, (size_t)n * st_chunk * 4, hipMemcpyHostToDevice, stream));
    d_aud = d_audio.as<float>();
  }
  // ---- which path this step takes (see the comment at st_fused_max)
  bool step_fused = st_fused && std::is_same<T, bf16_t>::value && n <= st_fused_max && st_cooldown == 0 && !foreign_now;
  bool snapshot = false;
  if (st_cooldown > 0) --st_cooldown;
  if (step_fused && st_share_rule && asr_tenant_busy_others(this, 5.0) > 0) { step_fused = false; ++st_shared_steps; }
  if (step_fused) snapshot = st_snapshot_env == 1 || (st_snapshot_env < 0 && asr_tenant_live_others(this) > 0);
  if (snapshot) { ensure_stream_shadow(); ++st_snapshots; }
  const bool inject_fault = step_fused && st_fault == 1;
  if (inject_fault) st_fault = 2;                        // once per session
  const UttPlan* dp = d_plan.as<UttPlan>();
  const int32_t* d_blk_utt = (const int32_t*)((unsigned char*)d_plan.ptr + sizeof(UttPlan) * n);
  const int32_t* d_blk_f0 = d_blk_utt + n;
  const int32_t* d_row_utt = d_blk_f0 + n;

  // Everything below reads its geometry from the uploaded plan and the device-side state (history lengths, fired-token counts), so
  // the launch sequence depends on n only: one captured hipGraph replays every chunk step of a given set size.
  auto enqueue = [&]() {
  if (snapshot && step_fused)
    launch_stream_state_copy(st_segs.as<StreamStateSeg>(), st_n_segs, st_n_items, dp, n, false, stream);
  // ---- front-end: fbank of the chunk, LFR rows, carried rows in front (:386-399)
  {
    ProfScope ps(prof, "fbank", stream);
    FbankArgs fa;
    fa.audio = d_aud; fa.plan = dp; fa.blk_utt = d_blk_utt; fa.blk_f0 = d_blk_f0;
    fa.dft_packed = dft; fa.mel_packed = melp; fa.mel_out = d_mel.as<float>();
    fa.n_bin_tiles = n_bin_tiles; fa.n_kchunks = n_kchunks; fa.n_mel_tiles = c.n_mels / 16; fa.n_mels = c.n_mels;
    fa.win = c.win_length; fa.hop = c.hop_length; fa.log_floor = 1.1920928955078125e-07f; fa.whisper = 0; fa.blk_max = nullptr;
    fa.dft_split = d_dft_split.ptr;
    launch_fbank(fa, n, stream);
  }
  {
    ProfScope ps(prof, "lfr_cmvn", stream);
    StreamLfrArgs la;
    la.mel = d_mel.as<float>(); la.plan = dp; la.cmvn_vars = cmvn_vars; la.pos_bias = speech_pos; la.prev = st_prev.as<float>();
    la.start = st_start.as<int32_t>(); la.out = d_x0.as<float>(); la.ld = kpad0; la.feat = feat; la.n_mels = c.n_mels; la.lfr_m = c.lfr_m;
    la.lfr_n = c.lfr_n; la.n_prev = st_C; la.n_new = st_B; la.n_rows = rows; la.n_frames = st_frames; la.pos_rows = max_lfr;
    launch_stream_lfr(la, stream);
    launch_stream_carry(d_x0.as<float>(), kpad0, dp, n, st_C, st_B, st_prev.as<float>(), st_start.as<int32_t>(), stream);
  }
  save_tap("enc_in", d_x0.ptr, rows, feat, kpad0, 4);
  // ---- encoder layers with K/V history (:400-435)
  const float* x_in = d_x0.as<float>();
  int ld_in = kpad0;
  float* xa = d_xa.as<float>();
  float* xb = d_xb.as<float>();
  T* h = d_h.as<T>();
  T* qkv = d_sqkv.as<T>();
  T* ctx = d_ctx.as<T>();
  float* mem = d_mem.as<float>();
  T* ffn = d_ffn.as<T>();
  const size_t en_layer = (size_t)st_max * H * st_en_cap * 128;
  const bool fused = step_fused;
  // Round 5: the per-launch layers evaluate their SECOND LayerNorm inside FFN-1, as the offline four-launch path does (rstd (x W^T - mean colsum(W)) + b over the bf16 copy of
  // the out-projection's result, row statistics handed over by that GEMM's epilogue) -- at 256 streams the ~100 stand-alone LayerNorm launches of a chunk step were 10 % of it at
  // their 5.4 us launch floor; this removes half of them. The first LayerNorm stays a launch: the LayerNorm-folded GEMM has no instance for the q|k|v epilogue (the offline path
  // folds it inside sanm_qkv_attn_kernel). ASR_LN_FUSED=0 restores the launches.
  bool st_alg = false;
  bf16_t* xblo = nullptr;
  float2* stb = nullptr;
  if constexpr (sizeof(T) == 2) {
    const SvBlock& bl = blocks[c.n_blocks - 1];
    GemmArgs probe;
    probe.M = rows; probe.N = dff; probe.K = d; probe.ln_dim = d; probe.ln_colsum = bl.c1; probe.act = ACT_RELU; probe.bias = bl.b1;
    probe.out_lo = d_ffn.ptr; probe.A = d_xblo.ptr; probe.W = bl.w1;
    st_alg = use_ln_alg && bl.c1 && d_xblo.ptr && gemm_ln_fusable(probe);
    xblo = d_xblo.as<bf16_t>(); stb = d_stb.as<float2>();
  }
  for (int i = 0; i < c.n_blocks; ++i) {
    const SvBlock& b = blocks[i];
    if (fused && i == 1) {               // layers 1 .. n - 1: one launch, the stream's rows stay in xa
      ProfScope ps(prof, "stream_layers", stream);
      const int nl = c.n_blocks - 1;
      HIP_CHECK(hipMemsetAsync(st_flags.ptr, 0, ((size_t)nl * n * 4 + 4 + (size_t)pdec.size() * n * 8) * 4, stream));
      StreamLayersArgs la;
      la.plan = dp; la.n_streams = n; la.n_layers = nl; la.n_cur = n_cur; la.cap = st_en_cap; la.roll_rows = st_B; la.ktaps = c.fsmn_kernel;
      la.ln_eps = 1e-5f; la.cache_len = st_enlen.as<int32_t>(); la.layers = st_layer_tab.as<StreamLayer>();
      la.x = xa; la.xb = xb; la.ctx = (bf16_t*)ctx; la.hid = (bf16_t*)ffn;
      la.flags = st_flags.as<unsigned>(); la.err = st_flags.as<unsigned>() + (size_t)nl * n * 4;
      la.opt = st_opt; la.fault = inject_fault ? 1 : 0;
      if (st_times_layer >= 0) {
        st_times.reserve((size_t)((n + 7) / 8) * 32 * 16 * 8, stream);
        HIP_CHECK(hipMemsetAsync(st_times.ptr, 0, (size_t)((n + 7) / 8) * 32 * 16 * 8, stream));
        la.times = st_times.as<unsigned long long>(); la.times_layer = st_times_layer;
      }
      launch_stream_layers(la, stream);
      break;
    }
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(x_in, ld_in, rows, b.in_size, b.ln1_g, b.ln1_b, 1e-5f, h, b.kpad, b.kpad, stream); }
    {
      ProfScope ps(prof, "gemm_qkv", stream);
      GemmArgs g;
      g.A = h; g.lda = b.kpad; g.W = b.wqkv; g.ldw = b.kpad; g.M = rows; g.N = 3 * d; g.K = b.kpad; g.bias = b.bqkv; g.out_lo = qkv; g.ld_out_lo = 3 * d;
      gemm(g);
    }
    {
      ProfScope ps(prof, "attention", stream);
      T* ck = st_enk.as<T>() + (size_t)i * en_layer;
      T* cv = st_env.as<T>() + (size_t)i * en_layer;
      StreamAttnArgs sa;
      sa.q = qkv; sa.ld_q = 3 * d; sa.q_col0 = 0; sa.k = qkv; sa.ld_k = 3 * d; sa.k_col0 = d; sa.v = qkv; sa.ld_v = 3 * d; sa.v_col0 = 2 * d;
      sa.n_cur = n_cur; sa.cache_k = ck; sa.cache_v = cv; sa.cache_len = st_enlen.as<int32_t>(); sa.cap = st_en_cap; sa.q_plan = dp; sa.n_heads = H;
      sa.ctx = ctx; sa.ld_ctx = d;
      // the same launch rolls the history -- [-(cap + C) : -C] of (history ++ chunk): append the first B rows, keep the last cap (:421-422) --
      // and computes the FSMN memory term of the chunk rows
      sa.roll_rows = st_B;
      sa.fsmn_w = b.wfsmn; sa.fsmn_b = b.bfsmn; sa.ktaps = c.fsmn_kernel; sa.mem = mem; sa.d = d;
      if (taps_enabled && i == 0) {                        // (bisect taps, tools/probes/stream_determinism.py: the layer's history as the attention launch finds it)
        save_tap("s0_ck", ck, (int64_t)st_max * H * st_en_cap, 128, 128, sizeof(T));
        save_tap("s0_cv", cv, (int64_t)st_max * H * st_en_cap, 128, 128, sizeof(T));
        save_tap("s0_len", st_enlen.ptr, st_max, 1, 1, 4);
      }
      launch_stream_attn<T>(sa, n, stream);
    }
    if (taps_enabled && (i == 0 || i == 3)) {             // (bisect taps of layers 0 and 3: what tools/probes/stream_determinism.py compares between a solo pass and a pass beside a co-tenant)
      const std::string pre = "s" + std::to_string(i) + "_";
      save_tap((pre + "h").c_str(), h, rows, b.kpad, b.kpad, sizeof(T));
      save_tap((pre + "qkv").c_str(), qkv, rows, 3 * d, 3 * d, sizeof(T));
      save_tap((pre + "ctx").c_str(), ctx, rows, d, d, sizeof(T));
      save_tap((pre + "mem").c_str(), mem, rows, d, d, 4);
    }
    {
      ProfScope ps(prof, "gemm_out", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = b.wout; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.add = mem; g.ld_add = d;
      if (i > 0) { g.add2 = x_in; g.ld_add2 = ld_in; }             // residual from the second layer on (:402-403,431-432)
      g.out_f32 = xb; g.ld_out_f32 = d;
      if (st_alg) { g.out_lo = xblo; g.ld_out_lo = d; g.st_out = stb; }
      gemm(g);
    }
    if (!st_alg) { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(xb, d, rows, d, b.ln2_g, b.ln2_b, 1e-5f, h, d, d, stream); }
    {
      ProfScope ps(prof, "gemm_ffn1", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = b.w1; g.ldw = d; g.M = rows; g.N = dff; g.K = d; g.bias = b.b1; g.act = ACT_RELU; g.out_lo = ffn; g.ld_out_lo = dff;
      if (st_alg) { g.A = xblo; g.ln_colsum = b.c1; g.ln_dim = d; g.ln_stats_in = stb; g.ln_slots = d / 32; }
      gemm(g);
    }
    {
      ProfScope ps(prof, "gemm_ffn2", stream);
      GemmArgs g;
      g.A = ffn; g.lda = dff; g.W = b.w2; g.ldw = dff; g.M = rows; g.N = d; g.K = dff; g.bias = b.b2; g.add = xb; g.ld_add = d; g.out_f32 = xa; g.ld_out_f32 = d;
      gemm(g);
    }
    if (taps_enabled && (i == 0 || i == 3)) {
      const std::string pre = "s" + std::to_string(i) + "_";
      save_tap((pre + "xb").c_str(), xb, rows, d, d, 4);
      save_tap((pre + "ffn").c_str(), ffn, rows, dff, dff, sizeof(T));
      save_tap((pre + "xa").c_str(), xa, rows, d, d, 4);
    }
    x_in = xa;
    ld_in = d;
  }
  // ---- after_norm, CIF conv + alphas, unrolled integrate-and-fire with the carried state (:436-462)
  float* enc32 = d_xb.as<float>();
  T* enc_lo = d_enc_lo.as<T>();
  float* dec = d_dec.as<float>();
  UttPlan* tplan = d_tplan.as<UttPlan>();
  {
    ProfScope ps(prof, "layernorm", stream);
    launch_layernorm<float>(xa, d, rows, d, after_g, after_b, 1e-5f, enc32, d, d, stream);
    launch_layernorm<T>(xa, d, rows, d, after_g, after_b, 1e-5f, enc_lo, d, d, stream);
  }
  save_tap("enc_out", enc32, rows, d, d, 4);
  {
    ProfScope ps(prof, "cif", stream);
    launch_shift3<T>(enc_lo, d, dp, d_row_utt, Mpad, d_cifa.as<T>(), stream);
    GemmArgs g;
    g.A = d_cifa.ptr; g.lda = 3 * d; g.W = cif_conv_w; g.ldw = 3 * d; g.M = rows; g.N = d; g.K = 3 * d; g.bias = cif_conv_b; g.act = ACT_RELU;
    g.out_lo = ctx; g.ld_out_lo = d;
    gemm(g);
    launch_alpha<T>(ctx, d, cif_out_w, cif_out_b, rows, d_alpha.as<float>(), stream);
    launch_stream_cif(d_alpha.as<float>(), enc32, d, dp, n, st_B, st_cifh.as<float>(), st_cifa.as<float>(), dec, tplan, d_num.as<int32_t>(), stream);
  }
  save_tap("alphas", d_alpha.ptr, rows, 1, 1, 4);
  save_tap("list_frame", dec, rows, d, d, 4);
  // ---- decoder over the fired frames (:508-553); streams without a fired frame leave their decoder state untouched
  T* q = d_qk.as<T>();
  T* kv = d_skv.as<T>();
  float* ffn32 = d_ffn32.as<float>();
  float* x1 = d_mem.as<float>();
  float* x2 = d_x2.as<float>();
  float* sa32 = d_sa.as<float>();
  const size_t de_layer = (size_t)st_max * H * st_de_cap * 128, fs_layer = (size_t)st_max * (c.fsmn_kernel - 1) * d;
  auto ffn_block = [&](const PfDecLayer& L, float* out) {
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(dec, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream); }
    ProfScope ps(prof, "gemm_dec", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = L.w1; g.ldw = d; g.M = rows; g.N = dd; g.K = d; g.bias = L.b1; g.act = ACT_RELU; g.out_f32 = ffn32; g.ld_out_f32 = dd;
    gemm(g);
    launch_layernorm<T>(ffn32, dd, rows, dd, nullptr, nullptr, 1e-5f, ffn, dd, dd, stream);
    GemmArgs g2;
    g2.A = ffn; g2.lda = dd; g2.W = L.w2; g2.ldw = dd; g2.M = rows; g2.N = d; g2.K = dd; g2.bias = L.b2; g2.out_f32 = out; g2.ld_out_f32 = d;
    gemm(g2);
  };
  int li = 0;
  if (st_dec_fused && step_fused) {      // every decoder block in one launch (stream_dec.hip); streams without a fired frame skip it
    ProfScope ps(prof, "stream_dec", stream);
    StreamDecArgs da;
    da.token_plan = tplan; da.n_streams = n; da.n_layers = (int)pdec.size(); da.n_cur = n_cur; da.cap = st_de_cap; da.ln_eps = 1e-5f;
    da.cache_len = st_delen.as<int32_t>(); da.layers = st_dlayer_tab.as<StreamDecLayer>(); da.enc = (const bf16_t*)enc_lo;
    da.dec = dec; da.x1 = x1; da.x2 = x2; da.hid = ffn32; da.ctx = (bf16_t*)ctx;
    unsigned* fb = st_flags.as<unsigned>() + (size_t)(c.n_blocks - 1) * n * 4;
    da.err = fb; da.flags = fb + 4; da.opt = st_opt >> 8;
    if (st_times_layer <= -2) {
      st_times.reserve((size_t)((n + 7) / 8) * 32 * 16 * 8, stream);
      HIP_CHECK(hipMemsetAsync(st_times.ptr, 0, (size_t)((n + 7) / 8) * 32 * 16 * 8, stream));
      da.times = st_times.as<unsigned long long>(); da.times_layer = -2 - st_times_layer;
    }
    launch_stream_dec(da, stream);
  } else
  for (const PfDecLayer& L : pdec) {
    if (!L.full) { ffn_block(L, dec); continue; }
    ffn_block(L, x1);
    {
      ProfScope ps(prof, "fsmn", stream);
      launch_layernorm<float>(x1, d, rows, d, L.n2_g, L.n2_b, 1e-5f, sa32, d, d, stream);
      launch_stream_dec_fsmn(sa32, dec, L.wfsmn, d, c.fsmn_kernel, tplan, n, st_defsmn.as<float>() + (size_t)li * fs_layer, x2, stream);
    }
    { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(x2, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream); }
    {
      ProfScope ps(prof, "gemm_dec", stream);
      GemmArgs g;
      g.A = h; g.lda = d; g.W = L.wq; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.bias = L.bq; g.out_lo = q; g.ld_out_lo = d;
      gemm(g);
      GemmArgs gk;
      gk.A = enc_lo; gk.lda = d; gk.W = L.wkv; gk.ldw = d; gk.M = rows; gk.N = 2 * d; gk.K = d; gk.bias = L.bkv; gk.out_lo = kv; gk.ld_out_lo = 2 * d;
      gemm(gk);
    }
    {
      ProfScope ps(prof, "attention", stream);
      T* ck = st_dek.as<T>() + (size_t)li * de_layer;
      T* cv = st_dev.as<T>() + (size_t)li * de_layer;
      StreamAttnArgs sa;
      sa.q = q; sa.ld_q = d; sa.q_col0 = 0; sa.k = kv; sa.ld_k = 2 * d; sa.k_col0 = 0; sa.v = kv; sa.ld_v = 2 * d; sa.v_col0 = d; sa.n_cur = n_cur;
      sa.cache_k = ck; sa.cache_v = cv; sa.cache_len = st_delen.as<int32_t>(); sa.cap = st_de_cap; sa.q_plan = tplan; sa.n_heads = H;
      sa.ctx = ctx; sa.ld_ctx = d;
      launch_stream_attn<T>(sa, n, stream);
      launch_stream_cache_roll<T>(ck, cv, st_delen.as<int32_t>(), st_de_cap, kv, 2 * d, 0, kv, 2 * d, d, n_cur, dp, tplan, n, H, stream);
    }
    {
      ProfScope ps(prof, "gemm_dec", stream);
      GemmArgs g;
      g.A = ctx; g.lda = d; g.W = L.wo; g.ldw = d; g.M = rows; g.N = d; g.K = d; g.bias = L.bo; g.add = x2; g.ld_add = d; g.out_f32 = dec; g.ld_out_f32 = d;
      gemm(g);
    }
    ++li;
  }
  { ProfScope ps(prof, "layernorm", stream); launch_layernorm<T>(dec, d, rows, d, nullptr, nullptr, 1e-5f, h, d, d, stream); }
  {
    ProfScope ps(prof, "gemm_out", stream);
    GemmArgs g;
    g.A = h; g.lda = d; g.W = pf_out_w; g.ldw = d; g.M = rows; g.N = vpad; g.K = d; g.bias = pf_out_b; g.out_f32 = d_logits.as<float>(); g.ld_out_f32 = vpad;
    gemm(g);
    launch_argmax_rows(d_logits.as<float>(), vpad, rows, c.vocab, nullptr, d_ids.as<int32_t>(), stream);
    launch_gather_tokens(d_ids.as<int32_t>(), tplan, n, d_tok.as<int32_t>(), max_tokens, stream);
  }
  launch_stream_advance(dp, tplan, n, st_B, st_en_cap, n_cur, st_de_cap, st_enlen.as<int32_t>(), st_delen.as<int32_t>(), stream);
  };
  {
    const bool graphable = use_graph && !taps_enabled && !prof.enabled;
    uint64_t key = 1469598103934665603ull;
    for (const void* q : {(const void*)d_aud, (const void*)d_plan.ptr, (const void*)d_mel.ptr, (const void*)d_x0.ptr, (const void*)d_xa.ptr, (const void*)d_xb.ptr,
                          (const void*)d_h.ptr, (const void*)d_sqkv.ptr, (const void*)d_skv.ptr, (const void*)d_qk.ptr, (const void*)d_ctx.ptr, (const void*)d_mem.ptr,
                          (const void*)d_ffn.ptr, (const void*)d_amax_v.ptr, (const void*)d_amax_i.ptr, (const void*)d_ids.ptr, (const void*)d_tok.ptr, (const void*)d_num.ptr,
                          (const void*)d_logits.ptr, (const void*)d_enc_lo.ptr, (const void*)d_cifa.ptr, (const void*)d_alpha.ptr, (const void*)d_dec.ptr, (const void*)d_x2.ptr,
                          (const void*)d_sa.ptr, (const void*)d_ffn32.ptr, (const void*)d_tplan.ptr, (const void*)stream, (const void*)(uintptr_t)n,
                          (const void*)(uintptr_t)max_tokens, (const void*)st_shadow.ptr, (const void*)d_xblo.ptr, (const void*)d_stb.ptr})
      key = (key ^ (uint64_t)(uintptr_t)q) * 1099511628211ull;
    const int gi = step_fused ? (snapshot ? 2 : 1) : 0;            // one cached graph per path: a session that alternates (a co-tenant comes and goes) does not re-capture
    hipGraphExec_t& st_graph = this->st_graph[gi];
    uint64_t& st_graph_key = this->st_graph_key[gi];
    uint64_t& st_eager_key = this->st_eager_key[gi];
    if (graphable && !inject_fault && st_graph && key == st_graph_key) {
      HIP_CHECK(hipGraphLaunch(st_graph, stream));
    } else if (graphable && !inject_fault && key == st_eager_key) {
      if (st_graph) { (void)hipGraphExecDestroy(st_graph); st_graph = nullptr; }
      hipGraph_t graph = nullptr;
      HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      try {
        enqueue();
      } catch (...) {
        (void)hipStreamEndCapture(stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      HIP_CHECK(hipStreamEndCapture(stream, &graph));
      HIP_CHECK(hipGraphInstantiate(&st_graph, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      st_graph_key = key;
      HIP_CHECK(hipGraphLaunch(st_graph, stream));
    } else {
      enqueue();                                         // first step of a geometry runs eagerly (lazy kernel attributes, workspaces)
      if (graphable && !inject_fault) st_eager_key = key;
    }
  }
  if (taps_enabled) save_tap("logits", d_logits.ptr, rows, c.vocab, vpad, 4);
  HIP_CHECK(hipMemcpyAsync(h_out, d_tok.ptr, (size_t)n * max_tokens * 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)n * max_tokens * 4, d_num.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  const bool fused_ran = step_fused;
  unsigned* h_err = (unsigned*)((unsigned char*)h_out + out_bytes);
  *h_err = 0;
  if (fused_ran)
    HIP_CHECK(hipMemcpyAsync(h_err, st_flags.as<unsigned>() + (size_t)(c.n_blocks - 1) * n * 4, 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  if (prof.enabled) prof.collect();
  // (a cluster that gave up has rolled the histories of the layers in front of it: with a snapshot the state is restored and the step redone on the per-launch path
  //  below; without one -- no other session existed when the step started -- the step fails loudly and the streams have to be reset)
  if (fused_ran && (st_times_layer >= 0 || st_times_layer <= -2)) {               // tuning: mean phase intervals over the workgroups (100 MHz clock -> us)
    const int nwg = (n + 7) / 8 * 32;
    std::vector<unsigned long long> t((size_t)nwg * 16);
    HIP_CHECK(hipMemcpy(t.data(), st_times.ptr, t.size() * 8, hipMemcpyDeviceToHost));
    double sum[14] = {}; int cnt = 0;
    const int last = st_times_layer >= 0 ? 12 : 13;
    for (int w = 0; w < nwg; ++w) {
      if (!t[(size_t)w * 16]) continue;
      ++cnt;
      for (int k = 1; k <= last; ++k) sum[k] += (double)(t[(size_t)w * 16 + k] - t[(size_t)w * 16 + k - 1]) * 0.01;
    }
    fprintf(stderr, "%s layer %d (us, mean of %d workgroups):", st_times_layer >= 0 ? "stream_layers" : "stream_dec", st_times_layer >= 0 ? st_times_layer : -2 - st_times_layer, cnt);
    for (int k = 1; k <= last; ++k) fprintf(stderr, " %.2f", sum[k] / std::max(cnt, 1));
    fprintf(stderr, "\n");
  }
  if (fused_ran && *h_err != 0 && snapshot) {
    // A cluster gave up (its workgroups were not co-resident: somebody else's kernels hold CUs). The histories of the layers in front of the stall are rolled,
    // everything behind the launch ran on garbage: put the active streams' state back as it was in front of the step, redo the step on the per-launch path
    // (no cross-workgroup waits) and stay there for a while.
    ++st_giveups;
    st_cooldown = st_cooldown_steps;
    fprintf(stderr, "asr_mi355x: streaming step: a cluster of the fused launch gave up; state restored, step redone on the per-launch path (give-up %d of this session)\n", st_giveups);
    launch_stream_state_copy(st_segs.as<StreamStateSeg>(), st_n_segs, st_n_items, dp, n, true, stream);
    step_fused = false; snapshot = false;
    enqueue();
    HIP_CHECK(hipMemcpyAsync(h_out, d_tok.ptr, (size_t)n * max_tokens * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync((unsigned char*)h_out + (size_t)n * max_tokens * 4, d_num.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (prof.enabled) prof.collect();
    *h_err = 0;
  }
  ASR_REQUIRE(*h_err == 0, "streaming: a workgroup of the fused encoder launch gave up waiting for its cluster and no snapshot was taken (ASR_STREAM_SNAPSHOT=0, or no other "
              "session existed on this GPU when the step started); the step's results are invalid, reset its streams");
  memcpy(num_out, (unsigned char*)h_out + (size_t)n * max_tokens * 4, (size_t)n * 4);
  const int32_t* ht = (const int32_t*)h_out;
  for (int i = 0; i < n; ++i) memcpy(tok_out + (size_t)i * max_tokens, ht + (size_t)i * max_tokens, (size_t)std::min(num_out[i], max_tokens) * 4);
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
      asr_tenant_attach(s);
      s->precision = precision;
      s->cfg = *cfg;
      s->load_env();
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
    TenantScope tenant(s);
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

extern "C" int asr_paraformer_create(const asr_paraformer_config* cfg, const void* arena, size_t arena_bytes, int arena_mem,
                                     int device_id, int precision, asr_session** out) {
  return asr_guard([&] {
    ASR_REQUIRE(cfg && arena && out, "paraformer_create: null argument");
    ASR_REQUIRE(precision == ASR_PRECISION_BF16 || precision == ASR_PRECISION_F32, "paraformer_create: bad precision %d", precision);
    asr_require_device(device_id);
    SvSession* s = new SvSession();
    try {
      s->kind = 3;
      s->device = device_id;
      asr_tenant_attach(s);
      s->precision = precision;
      s->paraformer = true;
      s->pcfg = *cfg;
      asr_sensevoice_config& c = s->cfg;
      memset(&c, 0, sizeof(c));
      c.sample_rate = cfg->sample_rate; c.n_mels = cfg->n_mels; c.nfft = cfg->nfft; c.win_length = cfg->win_length; c.hop_length = cfg->hop_length;
      c.lfr_m = cfg->lfr_m; c.lfr_n = cfg->lfr_n; c.d_model = cfg->d_model; c.n_heads = cfg->n_heads; c.d_head = cfg->d_head; c.d_ffn = cfg->d_ffn;
      c.n_blocks = cfg->n_blocks; c.n_main = cfg->n_blocks; c.fsmn_kernel = cfg->fsmn_kernel; c.vocab = cfg->vocab; c.blank_id = -1;
      c.n_prompt = 0; c.n_languages = 0; c.max_audio_len = cfg->max_audio_len;
      s->load_env();
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

extern "C" int asr_paraformer_stream_create(const asr_paraformer_config* cfg, const void* arena, size_t arena_bytes, int arena_mem,
                                            int device_id, int precision, int chunk_samples, int look_back_encoder, int look_back_decoder,
                                            int max_streams, asr_session** out) {
  asr_session* base = nullptr;
  const int rc = asr_paraformer_create(cfg, arena, arena_bytes, arena_mem, device_id, precision, &base);
  if (rc != ASR_OK) return rc;
  const int rc2 = asr_guard([&] {
    SvSession* sv = static_cast<SvSession*>(base);
    sv->kind = 4;
    sv->stream_init(chunk_samples, look_back_encoder, look_back_decoder, max_streams);
    *out = base;
  });
  if (rc2 != ASR_OK) delete base;
  return rc2;
}

extern "C" int asr_paraformer_stream_reset(asr_session* s, int stream_id) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 4, "paraformer_stream_reset: not a streaming Paraformer session");
    static_cast<SvSession*>(s)->stream_reset(stream_id);
  });
}

extern "C" int asr_paraformer_stream_step(asr_session* s, const float* audio, int audio_mem, const int32_t* stream_ids, int n_streams,
                                          int32_t* token_ids_out, int max_tokens, int32_t* num_id_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 4, "paraformer_stream_step: not a streaming Paraformer session");
    TenantScope tenant(s);
    SvSession* sv = static_cast<SvSession*>(s);
    if (sv->precision == ASR_PRECISION_BF16) sv->stream_step<bf16_t>(audio, audio_mem, stream_ids, n_streams, token_ids_out, max_tokens, num_id_out);
    else sv->stream_step<float>(audio, audio_mem, stream_ids, n_streams, token_ids_out, max_tokens, num_id_out);
  });
}

extern "C" int asr_paraformer_stream_stats(asr_session* s, int32_t* out8) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 4 && out8, "paraformer_stream_stats: not a streaming Paraformer session");
    SvSession* sv = static_cast<SvSession*>(s);
    out8[0] = sv->st_giveups; out8[1] = sv->st_shared_steps; out8[2] = sv->st_snapshots; out8[3] = sv->st_fused_max; out8[4] = sv->st_cooldown;
    out8[5] = sv->st_fused ? 1 : 0; out8[6] = sv->foreign_diverted; out8[7] = 0;
  });
}

extern "C" int asr_sanm_stats(asr_session* s, int32_t* out8) {
  return asr_guard([&] {
    ASR_REQUIRE(s && (s->kind == 1 || s->kind == 3 || s->kind == 4) && out8, "sanm_stats: not a SenseVoice / Paraformer session");
    SvSession* sv = static_cast<SvSession*>(s);
    out8[0] = sv->block_giveups; out8[1] = sv->block_cooldown; out8[2] = sv->foreign_diverted; out8[3] = sv->use_block ? 1 : 0;
    out8[4] = out8[5] = out8[6] = out8[7] = 0;
  });
}

extern "C" int asr_paraformer_run(asr_session* s, const float* audio, int audio_mem, const int64_t* audio_offsets, int batch,
                                  int32_t* token_ids_out, int max_tokens, int32_t* num_id_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && s->kind == 3, "paraformer_run: not a Paraformer session");
    TenantScope tenant(s);
    SvSession* sv = static_cast<SvSession*>(s);
    if (sv->precision == ASR_PRECISION_BF16)
      sv->run<bf16_t>(audio, audio_mem, audio_offsets, batch, nullptr, token_ids_out, max_tokens, num_id_out);
    else
      sv->run<float>(audio, audio_mem, audio_offsets, batch, nullptr, token_ids_out, max_tokens, num_id_out);
  });
}
