// C ABI: session utilities and operator-level entry points (see include/asr_mi355x.h).
#include <cstring>

#include "../../include/asr_mi355x.h"
#include "engine.h"
#include "gemm.h"
#include "kernels.h"

const std::string& asr_get_error();

extern "C" int asr_abi_version(void) { return ASR_ABI_VERSION; }
extern "C" const char* asr_last_error(void) { return asr_get_error().c_str(); }

extern "C" int asr_device_count(int* count) {
  return asr_guard([&] {
    ASR_REQUIRE(count, "device_count: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
  });
}

extern "C" int asr_mem_alloc(int device_id, size_t bytes, void** out) {
  return asr_guard([&] {
    ASR_REQUIRE(out, "mem_alloc: null argument");
    asr_require_device(device_id);
    HIP_CHECK(hipMalloc(out, std::max<size_t>(bytes, 16)));
  });
}

extern "C" int asr_mem_free(int device_id, void* ptr) {
  return asr_guard([&] {
    if (!ptr) return;
    asr_require_device(device_id);
    HIP_CHECK(hipFree(ptr));
  });
}

extern "C" int asr_mem_copy(int device_id, void* dst, const void* src, size_t bytes, int kind) {
  return asr_guard([&] {
    ASR_REQUIRE(dst && src, "mem_copy: null argument");
    ASR_REQUIRE(kind >= 0 && kind <= 2, "mem_copy: bad kind %d", kind);
    asr_require_device(device_id);
    static const hipMemcpyKind kinds[3] = {hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice};
    HIP_CHECK(hipMemcpy(dst, src, bytes, kinds[kind]));
  });
}

extern "C" int asr_session_destroy(asr_session* s) {
  return asr_guard([&] {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    delete s;
  });
}

extern "C" int asr_session_set_stream(asr_session* s, void* hip_stream) {
  return asr_guard([&] {
    ASR_REQUIRE(s, "set_stream: null session");
    HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    s->stream = reinterpret_cast<hipStream_t>(hip_stream);
    s->own_stream = false;
  });
}

extern "C" int asr_session_device(asr_session* s, int* device_id) {
  return asr_guard([&] {
    ASR_REQUIRE(s && device_id, "session_device: null argument");
    *device_id = s->device;
  });
}

extern "C" int asr_session_profile_enable(asr_session* s, int enable) {
  return asr_guard([&] {
    ASR_REQUIRE(s, "profile_enable: null session");
    s->prof.enabled = enable != 0;
  });
}

extern "C" int asr_session_profile_reset(asr_session* s) {
  return asr_guard([&] {
    ASR_REQUIRE(s, "profile_reset: null session");
    s->prof.reset();
  });
}

extern "C" int asr_session_profile_read(asr_session* s, int cap, char* names, double* total_ms, int64_t* launches, int* n_out) {
  return asr_guard([&] {
    ASR_REQUIRE(s && names && total_ms && launches && n_out, "profile_read: null argument");
    const int n = std::min<int>(cap, (int)s->prof.names.size());
    for (int i = 0; i < n; ++i) {
      memset(names + i * 32, 0, 32);
      strncpy(names + i * 32, s->prof.names[i].c_str(), 31);
      total_ms[i] = s->prof.total_ms[i];
      launches[i] = s->prof.launches[i];
    }
    *n_out = n;
  });
}

extern "C" int asr_session_taps_enable(asr_session* s, int enable) {
  return asr_guard([&] {
    ASR_REQUIRE(s, "taps_enable: null session");
    s->taps_enabled = enable != 0;
  });
}

extern "C" int asr_session_tap_shape(asr_session* s, const char* name, int64_t* rows, int64_t* cols) {
  return asr_guard([&] {
    ASR_REQUIRE(s && name && rows && cols, "tap_shape: null argument");
    auto it = s->taps.find(name);
    if (it == s->taps.end()) throw AsrError{ASR_ERR_NOT_FOUND, std::string("tap '") + name + "' not recorded (enable taps, then run)"};
    *rows = it->second.rows;
    *cols = it->second.cols;
  });
}

extern "C" int asr_session_tap_read(asr_session* s, const char* name, void* host_out, size_t bytes) {
  return asr_guard([&] {
    ASR_REQUIRE(s && name && host_out, "tap_read: null argument");
    auto it = s->taps.find(name);
    if (it == s->taps.end()) throw AsrError{ASR_ERR_NOT_FOUND, std::string("tap '") + name + "' not recorded (enable taps, then run)"};
    const Tap& t = it->second;
    const size_t need = (size_t)t.rows * t.cols * t.elt;
    ASR_REQUIRE(bytes == need, "tap_read: '%s' is %zu bytes, buffer is %zu", name, need, bytes);
    HIP_CHECK(hipSetDevice(s->device));
    // on the session's own stream, not the legacy null stream: a synchronous null-stream copy next to captured graphs is what tools/probes/stream_taps_toggle.py
    // trips over on this runtime (taps read between steps, then the cached step graph replayed: a cluster of the fused launch never saw its siblings)
    HIP_CHECK(hipMemcpyAsync(host_out, t.buf.ptr, need, hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
  });
}

// =========================================================================================== operator hooks
namespace {

struct Tmp {
  std::vector<void*> ptrs;
  ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
    HIP_CHECK(hipMemset(p, 0, std::max<size_t>(bytes, 256)));
    ptrs.push_back(p);
    return p;
  }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// upload a host f32 matrix [rows][cols] into a zero-padded device matrix [rows_pad][ld] of the operand dtype
void* upload_operand(Tmp& t, int precision, const float* src, int rows, int cols, int rows_pad, int ld) {
  if (precision == ASR_PRECISION_F32) {
    float* d = (float*)t.alloc((size_t)rows_pad * ld * 4);
    HIP_CHECK(hipMemcpy2D(d, (size_t)ld * 4, src, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice));
    return d;
  }
  std::vector<bf16_t> tmp((size_t)rows * ld, 0);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) tmp[(size_t)r * ld + c] = f32_to_bf16(src[(size_t)r * cols + c]);
  bf16_t* d = (bf16_t*)t.alloc((size_t)rows_pad * ld * 2);
  HIP_CHECK(hipMemcpy(d, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return d;
}

void download_operand(int precision, const void* dsrc, int rows, int cols, int ld, float* out) {
  if (precision == ASR_PRECISION_F32) {
    HIP_CHECK(hipMemcpy2D(out, (size_t)cols * 4, dsrc, (size_t)ld * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost));
    return;
  }
  std::vector<bf16_t> tmp((size_t)rows * ld);
  HIP_CHECK(hipMemcpy(tmp.data(), dsrc, tmp.size() * 2, hipMemcpyDeviceToHost));
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      union { uint32_t u; float f; } cv;
      cv.u = ((uint32_t)tmp[(size_t)r * ld + c]) << 16;
      out[(size_t)r * cols + c] = cv.f;
    }
}

struct PackedPlan {
  std::vector<UttPlan> plan;
  std::vector<int32_t> qb_utt, qb_q0, row_utt;
  int rows = 0, Mpad = 0;
  PackedPlan(const int32_t* seq_lens, int batch, int q_rows = 64) {
    plan.resize(batch);
    for (int b = 0; b < batch; ++b) {
      ASR_REQUIRE(seq_lens[b] > 0, "op: empty sequence %d", b);
      memset(&plan[b], 0, sizeof(UttPlan));
      plan[b].T = seq_lens[b];
      plan[b].row_off = rows;
      for (int q0 = 0; q0 < seq_lens[b]; q0 += q_rows) { qb_utt.push_back(b); qb_q0.push_back(q0); }
      rows += round_up(seq_lens[b], 16);
    }
    Mpad = round_up(rows, 128);
    row_utt.assign(Mpad, -1);
    for (int b = 0; b < batch; ++b)
      for (int r = 0; r < round_up(seq_lens[b], 16); ++r) row_utt[plan[b].row_off + r] = b;
  }
};

// scatter dense-packed host rows [sum T][cols] into the 16-row-aligned packed layout (and back)
void to_aligned(const PackedPlan& pp, const int32_t* seq_lens, int batch, int cols, const float* src, std::vector<float>& dst) {
  dst.assign((size_t)pp.Mpad * cols, 0.0f);
  size_t r = 0;
  for (int b = 0; b < batch; ++b) {
    memcpy(&dst[(size_t)pp.plan[b].row_off * cols], src + r * cols, (size_t)seq_lens[b] * cols * 4);
    r += seq_lens[b];
  }
}
void from_aligned(const PackedPlan& pp, const int32_t* seq_lens, int batch, int cols, const std::vector<float>& src, float* dst) {
  size_t r = 0;
  for (int b = 0; b < batch; ++b) {
    memcpy(dst + r * cols, &src[(size_t)pp.plan[b].row_off * cols], (size_t)seq_lens[b] * cols * 4);
    r += seq_lens[b];
  }
}

}  // namespace

extern "C" int asr_op_gemm(int precision, const float* a, const float* w, const float* bias, int M, int N, int K, int act,
                           float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(a && w && out, "op_gemm: null argument");
    asr_require_device(0);
    gemm_reload_env();
    Tmp t;
    const int Mp = round_up(M, 128), Np = round_up(N, 128), Kp = round_up(K, 64);
    void* da = upload_operand(t, precision, a, M, K, Mp, Kp);
    void* dw = upload_operand(t, precision, w, N, K, Np, Kp);
    float* db = nullptr;
    if (bias) {
      db = (float*)t.alloc((size_t)Np * 4);
      HIP_CHECK(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    }
    float* dout = (float*)t.alloc((size_t)Mp * Np * 4);
    GemmArgs g;
    g.A = da; g.lda = Kp; g.W = dw; g.ldw = Kp; g.M = M; g.N = Np; g.K = Kp; g.bias = db; g.act = act;
    g.out_f32 = dout; g.ld_out_f32 = Np;
    g.sk_ws = (float*)t.alloc((size_t)8 << 20); g.sk_ws_bytes = (size_t)8 << 20;     // lets the launcher pick the split-K tiled path for small grids
    if (precision == ASR_PRECISION_BF16) launch_gemm_bf16(g, nullptr); else launch_gemm_f32(g, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy2D(out, (size_t)N * 4, dout, (size_t)Np * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost));
  });
}

extern "C" int asr_op_layernorm(int precision, const float* x, int rows, int D, const float* gamma, const float* beta, float eps,
                                float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(x && out, "op_layernorm: null argument");
    asr_require_device(0);
    Tmp t;
    float* dx = (float*)t.alloc((size_t)rows * D * 4);
    HIP_CHECK(hipMemcpy(dx, x, (size_t)rows * D * 4, hipMemcpyHostToDevice));
    float *dg = nullptr, *db = nullptr;
    if (gamma) {
      dg = (float*)t.alloc((size_t)D * 4);
      db = (float*)t.alloc((size_t)D * 4);
      HIP_CHECK(hipMemcpy(dg, gamma, (size_t)D * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(db, beta, (size_t)D * 4, hipMemcpyHostToDevice));
    }
    const int ld = round_up(D, 64);
    if (precision == ASR_PRECISION_F32) {
      float* dout = (float*)t.alloc((size_t)rows * ld * 4);
      launch_layernorm<float>(dx, D, rows, D, dg, db, eps, dout, ld, ld, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      download_operand(precision, dout, rows, D, ld, out);
    } else {
      bf16_t* dout = (bf16_t*)t.alloc((size_t)rows * ld * 2);
      launch_layernorm<bf16_t>(dx, D, rows, D, dg, db, eps, dout, ld, ld, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      download_operand(precision, dout, rows, D, ld, out);
    }
  });
}

extern "C" int asr_op_attention(int precision, const float* q, const float* k, const float* v, const int32_t* seq_lens,
                                int batch, int n_heads, int d_head, float* ctx) {
  return asr_guard([&] {
    ASR_REQUIRE(q && k && v && seq_lens && ctx && batch > 0, "op_attention: null argument");
    ASR_REQUIRE(precision == ASR_PRECISION_F32 || d_head == 128 || d_head == 64, "op_attention: bf16 kernel is built for head_dim 64/128");
    asr_require_device(0);
    Tmp t;
    int max_T = 0, att_qt = 0, att_nw = 4, q_rows = 64;
    for (int b = 0; b < batch; ++b) max_T = std::max(max_T, seq_lens[b]);
    if (precision == ASR_PRECISION_BF16) { attention_geometry(max_T, d_head, &att_qt, &att_nw); q_rows = 16 * att_qt * att_nw; }
    PackedPlan pp(seq_lens, batch, q_rows);
    const int d = n_heads * d_head;
    std::vector<float> qa, ka, va;
    to_aligned(pp, seq_lens, batch, d, q, qa);
    to_aligned(pp, seq_lens, batch, d, k, ka);
    to_aligned(pp, seq_lens, batch, d, v, va);
    std::vector<float> vt((size_t)d * pp.Mpad);
    for (int r = 0; r < pp.Mpad; ++r)
      for (int c = 0; c < d; ++c) vt[(size_t)c * pp.Mpad + r] = va[(size_t)r * d + c];
    void* dq = upload_operand(t, precision, qa.data(), pp.Mpad, d, pp.Mpad, d);
    void* dk = upload_operand(t, precision, ka.data(), pp.Mpad, d, pp.Mpad, d);
    void* dvt = upload_operand(t, precision, vt.data(), d, pp.Mpad, d, pp.Mpad);
    const size_t e = precision == ASR_PRECISION_F32 ? 4 : 2;
    void* dctx = t.alloc((size_t)pp.Mpad * d * e);
    UttPlan* dplan = (UttPlan*)t.alloc(sizeof(UttPlan) * batch);
    int32_t* dqu = (int32_t*)t.alloc(pp.qb_utt.size() * 4);
    int32_t* dq0 = (int32_t*)t.alloc(pp.qb_q0.size() * 4);
    HIP_CHECK(hipMemcpy(dplan, pp.plan.data(), sizeof(UttPlan) * batch, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dqu, pp.qb_utt.data(), pp.qb_utt.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dq0, pp.qb_q0.data(), pp.qb_q0.size() * 4, hipMemcpyHostToDevice));
    AttnArgs aa;
    aa.q = dq; aa.k = dk; aa.ld_qk = d; aa.vt = dvt; aa.ld_vt = pp.Mpad; aa.ctx = dctx; aa.ld_ctx = d;
    aa.plan = dplan; aa.qb_utt = dqu; aa.qb_q0 = dq0; aa.n_qblocks = (int)pp.qb_utt.size(); aa.n_heads = n_heads;
    aa.qt = att_qt; aa.n_waves = att_nw; aa.max_T = max_T;
    if (precision == ASR_PRECISION_F32) launch_attention_f32(aa, d_head, nullptr);
    else if (d_head == 128) launch_attention_bf16_hd128(aa, nullptr);
    else launch_attention_bf16_hd64(aa, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<float> ca((size_t)pp.Mpad * d);
    download_operand(precision, dctx, pp.Mpad, d, d, ca.data());
    from_aligned(pp, seq_lens, batch, d, ca, ctx);
  });
}

extern "C" int asr_op_fsmn(int precision, const float* v, const float* w, const float* b, const int32_t* seq_lens, int batch,
                           int channels, int ktaps, float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(v && w && b && seq_lens && out && batch > 0, "op_fsmn: null argument");
    asr_require_device(0);
    Tmp t;
    PackedPlan pp(seq_lens, batch);
    std::vector<float> va;
    to_aligned(pp, seq_lens, batch, channels, v, va);
    std::vector<float> vt((size_t)channels * pp.Mpad);
    for (int r = 0; r < pp.Mpad; ++r)
      for (int c = 0; c < channels; ++c) vt[(size_t)c * pp.Mpad + r] = va[(size_t)r * channels + c];
    void* dvt = upload_operand(t, precision, vt.data(), channels, pp.Mpad, channels, pp.Mpad);
    float* dw = (float*)t.alloc((size_t)channels * ktaps * 4);
    float* db = (float*)t.alloc((size_t)channels * 4);
    HIP_CHECK(hipMemcpy(dw, w, (size_t)channels * ktaps * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, b, (size_t)channels * 4, hipMemcpyHostToDevice));
    UttPlan* dplan = (UttPlan*)t.alloc(sizeof(UttPlan) * batch);
    int32_t* dru = (int32_t*)t.alloc((size_t)pp.Mpad * 4);
    HIP_CHECK(hipMemcpy(dplan, pp.plan.data(), sizeof(UttPlan) * batch, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dru, pp.row_utt.data(), (size_t)pp.Mpad * 4, hipMemcpyHostToDevice));
    float* dout = (float*)t.alloc((size_t)channels * pp.Mpad * 4);
    if (precision == ASR_PRECISION_F32)
      launch_fsmn<float>((const float*)dvt, pp.Mpad, dw, db, channels, ktaps, dplan, dru, pp.Mpad, dout, channels, nullptr);
    else
      launch_fsmn<bf16_t>((const bf16_t*)dvt, pp.Mpad, dw, db, channels, ktaps, dplan, dru, pp.Mpad, dout, channels, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<float> oa((size_t)pp.Mpad * channels);
    HIP_CHECK(hipMemcpy(oa.data(), dout, oa.size() * 4, hipMemcpyDeviceToHost));
    from_aligned(pp, seq_lens, batch, channels, oa, out);
  });
}

extern "C" int asr_op_ctc_collapse(const int32_t* frame_ids, const int32_t* seq_lens, int batch, int blank_id,
                                   int32_t* token_ids, int max_tokens, int32_t* num_id) {
  return asr_guard([&] {
    ASR_REQUIRE(frame_ids && seq_lens && token_ids && num_id && batch > 0 && max_tokens > 0, "op_ctc_collapse: bad argument");
    asr_require_device(0);
    Tmp t;
    PackedPlan pp(seq_lens, batch);
    std::vector<int32_t> ids(pp.Mpad, 0);
    size_t r = 0;
    for (int b = 0; b < batch; ++b) {
      memcpy(&ids[pp.plan[b].row_off], frame_ids + r, (size_t)seq_lens[b] * 4);
      r += seq_lens[b];
    }
    int32_t* dids = (int32_t*)t.alloc((size_t)pp.Mpad * 4);
    UttPlan* dplan = (UttPlan*)t.alloc(sizeof(UttPlan) * batch);
    int32_t* dtok = (int32_t*)t.alloc((size_t)batch * max_tokens * 4);
    int32_t* dnum = (int32_t*)t.alloc((size_t)batch * 4);
    HIP_CHECK(hipMemcpy(dids, ids.data(), (size_t)pp.Mpad * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dplan, pp.plan.data(), sizeof(UttPlan) * batch, hipMemcpyHostToDevice));
    launch_ctc_collapse(dids, dplan, batch, blank_id, dtok, max_tokens, dnum, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(token_ids, dtok, (size_t)batch * max_tokens * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(num_id, dnum, (size_t)batch * 4, hipMemcpyDeviceToHost));
  });
}

extern "C" int asr_op_gemm_ln(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, int M, int N,
                              int K, float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(x && w && out && M >= 1 && M <= 64 && N % 16 == 0 && K % 256 == 0, "op_gemm_ln: bad argument");
    asr_require_device(0);
    Tmp t;
    float* dx = (float*)t.alloc((size_t)M * K * 4);
    HIP_CHECK(hipMemcpy(dx, x, (size_t)M * K * 4, hipMemcpyHostToDevice));
    void* dw = upload_operand(t, ASR_PRECISION_BF16, w, N, K, N, K);
    float *db = nullptr, *dg = nullptr, *dbe = nullptr;
    if (bias) { db = (float*)t.alloc((size_t)N * 4); HIP_CHECK(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice)); }
    if (gamma) {
      dg = (float*)t.alloc((size_t)K * 4); dbe = (float*)t.alloc((size_t)K * 4);
      HIP_CHECK(hipMemcpy(dg, gamma, (size_t)K * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dbe, beta, (size_t)K * 4, hipMemcpyHostToDevice));
    }
    float* dout = (float*)t.alloc((size_t)M * N * 4);
    GemmArgs g;
    g.W = dw; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = db; g.ln_x = dx; g.ld_ln_x = K; g.ln_gamma = dg; g.ln_beta = dbe;
    g.out_f32 = dout; g.ld_out_f32 = N;
    launch_gemm_bf16(g, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, dout, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  });
}

