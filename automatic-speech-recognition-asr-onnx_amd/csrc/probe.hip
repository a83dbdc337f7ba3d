// Probe / test-hook library (libasr_mi355x_probe.so): timing probes and kernel-selection hooks that are NOT part of the product
// C ABI (include/asr_mi355x.h). Links against libasr_mi355x.so and calls its internal launchers; declared in
// include/asr_mi355x_probe.h, bound by automatic-speech-recognition-asr-onnx_amd/_probe.py, used by tests/ and tools/ only.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/asr_mi355x.h"
#include "../../include/asr_mi355x_probe.h"
#include "engine.h"
#include "gemm.h"
#include "kernels.h"

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
char g_chain_kernel[32] = "";
inline float bf16_bits_to_f32(bf16_t b) { union { uint32_t u; float f; } c; c.u = ((uint32_t)b) << 16; return c.f; }

}  // namespace

extern "C" int asr_probe_gemm_bench(int variant, int M, int N, int K, int epilogue, int iters, float* avg_ms) {
  return asr_guard([&] {
    ASR_REQUIRE(avg_ms && iters > 0, "probe_gemm_bench: bad argument");
    asr_require_device(0);
    Tmp t;
    const int Mp = round_up(M, 128);
    auto rnd = [](size_t n, uint32_t seed) {
      std::vector<bf16_t> v(n);
      uint32_t x = seed;
      for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; v[i] = f32_to_bf16(((int)(x >> 9) % 2001 - 1000) * 1e-3f); }
      return v;
    };
    std::vector<bf16_t> ha = rnd((size_t)Mp * K, 1), hw = rnd((size_t)N * K, 2);
    bf16_t* da = (bf16_t*)t.alloc(ha.size() * 2);
    bf16_t* dw = (bf16_t*)t.alloc(hw.size() * 2);
    HIP_CHECK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    float* bias = (float*)t.alloc((size_t)N * 4);
    float* addm = (float*)t.alloc((size_t)Mp * N * 4);
    float* addt = (float*)t.alloc((size_t)Mp * N * 4);
    float* of32 = (float*)t.alloc((size_t)Mp * N * 4);
    bf16_t* olo = (bf16_t*)t.alloc((size_t)Mp * N * 2);
    bf16_t* ot = (bf16_t*)t.alloc((size_t)Mp * N * 2);
    GemmArgs g;
    g.A = da; g.lda = K; g.W = dw; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias;
    switch (epilogue) {
      case 0: g.out_lo = olo; g.ld_out_lo = N; break;
      case 1: g.out_lo = olo; g.ld_out_lo = N; g.act = ACT_RELU; break;
      case 2: g.add = addm; g.ld_add = N; g.out_f32 = of32; g.ld_out_f32 = N; break;
      case 7: g.out_lo = olo; g.ld_out_lo = N; g.act = ACT_GELU_ERF; break;
      case 3: g.bias = nullptr; g.add = addt; g.ld_add = N; g.add2 = addm; g.ld_add2 = N; g.out_f32 = of32; g.ld_out_f32 = N; break;
      case 4: g.out_t = ot; g.ld_out_t = Mp; break;
      case 5: {   // FFN-1 with the LayerNorm evaluated inside (statistics handed over by the producer)
        float2* st = (float2*)t.alloc((size_t)Mp * (K / 32) * 8);
        HIP_CHECK(hipMemset(st, 0, (size_t)Mp * (K / 32) * 8));
        g.out_lo = olo; g.ld_out_lo = N; g.act = ACT_RELU; g.ln_colsum = bias; g.ln_dim = K; g.ln_stats_in = st; g.ln_slots = K / 32;
        break;
      }
      case 6: {   // out-projection writing the residual stream in f32 + bf16 + row statistics
        float2* st = (float2*)t.alloc((size_t)Mp * (N / 32) * 8);
        g.bias = nullptr; g.add = addt; g.ld_add = N; g.add2 = addm; g.ld_add2 = N; g.out_f32 = of32; g.ld_out_f32 = N;
        g.out_lo = olo; g.ld_out_lo = N; g.st_out = st;
        break;
      }
      default: ASR_THROW(ASR_ERR_INVALID, "probe_gemm_bench: unknown epilogue %d", epilogue);
    }
    g.dbg = variant < 0 ? 0 : variant >> 8;
    uint32_t* clk = nullptr;
    if (getenv("ASR_PP_CLK")) { clk = (uint32_t*)t.alloc(64 * 4); HIP_CHECK(hipMemset(clk, 0, 64 * 4)); g.dbg_clk = clk; }
    gemm_set_variant(variant < 0 ? -1 : (variant & 0xff));      // sticky: later asr_op_gemm calls use it too
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_gemm_bf16(g, nullptr);
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch_gemm_bf16(g, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    if (clk) {
      uint32_t h[40];
      HIP_CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
      const int ksteps = K / 64 - 2;
      for (int w = 0; w < 2; ++w)
        for (int r = 0; r < 4; ++r)
          fprintf(stderr, "pp clock: group %d phase %d: reads %6.1f  vmcnt %6.1f  barrier %6.1f | mfma %6.1f  barrier %6.1f  (cycles per phase)\n", w, r,
                  h[w * 20 + r * 5] / (double)ksteps, h[w * 20 + r * 5 + 1] / (double)ksteps, h[w * 20 + r * 5 + 2] / (double)ksteps,
                  h[w * 20 + r * 5 + 3] / (double)ksteps, h[w * 20 + r * 5 + 4] / (double)ksteps);
    }
    gemm_set_variant(-1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}


// Decode-shaped GEMM in a captured chain with COLD weights: `copies` weight matrices (together larger than the 256 MB Infinity Cache when
// cold_mb says so) are walked round-robin by dependent launches of one hipGraph, the way a decoder layer stack streams its weights from
// HBM once per token. Reports microseconds per launch (graph replay, device time between events).
extern "C" int asr_probe_gemm_chain(int M, int N, int K, int epilogue, int cold_mb, int replays, float* us_per_launch) {
  return asr_guard([&] {
    ASR_REQUIRE(us_per_launch && replays > 0 && M > 0 && N % 16 == 0 && K % 256 == 0, "probe_gemm_chain: bad argument");
    asr_require_device(0);
    gemm_reload_env();
    Tmp t;
    const int Mp = round_up(M, 128);
    const size_t wbytes = (size_t)N * K * 2;
    const int copies = std::max(1, std::min(512, (int)(((size_t)cold_mb << 20) / wbytes) + 1));
    std::vector<bf16_t> hw((size_t)N * K), ha((size_t)Mp * K);
    uint32_t x = 12345;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = f32_to_bf16(((int)(x >> 9) % 2001 - 1000) * 1e-3f); }
    for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = f32_to_bf16(((int)(x >> 9) % 2001 - 1000) * 1e-3f); }
    bf16_t* dw = (bf16_t*)t.alloc(wbytes * copies);
    for (int c = 0; c < copies; ++c) HIP_CHECK(hipMemcpy((char*)dw + wbytes * c, hw.data(), wbytes, hipMemcpyHostToDevice));
    bf16_t* da = (bf16_t*)t.alloc(ha.size() * 2);
    HIP_CHECK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    float* bias = (float*)t.alloc((size_t)N * 4);
    float* addm = (float*)t.alloc((size_t)Mp * N * 4);
    float* of32 = (float*)t.alloc((size_t)Mp * N * 4);
    bf16_t* olo = (bf16_t*)t.alloc((size_t)Mp * N * 2);
    float* lnx = (float*)t.alloc((size_t)Mp * K * 4);
    GemmArgs g;
    g.A = da; g.lda = K; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias;
    g.sk_ws = (float*)t.alloc((size_t)16 << 20); g.sk_ws_bytes = (size_t)16 << 20; g.sk_cnt = (int32_t*)t.alloc(4096 * 4);
    switch (epilogue >= 10 ? 0 : epilogue) {
      case 0: g.out_lo = olo; g.ld_out_lo = N; break;                                                  // bias -> bf16 (q|k|v, cross-q with a separate LayerNorm)
      case 1: g.out_lo = olo; g.ld_out_lo = N; g.act = ACT_GELU_ERF; break;                            // fc1
      case 2: g.add = addm; g.ld_add = N; g.out_f32 = of32; g.ld_out_f32 = N; break;                   // out-proj / fc2: + residual -> f32
      case 3: g.A = nullptr; g.ln_x = lnx; g.ld_ln_x = K; g.out_lo = olo; g.ld_out_lo = N; break;      // LayerNorm prologue -> bf16
      default: ASR_THROW(ASR_ERR_INVALID, "probe_gemm_chain: unknown epilogue %d", epilogue);
    }
    hipStream_t s;
    HIP_CHECK(hipStreamCreate(&s));
    const int chain = std::max(copies, 64);
    DecGemmArgs dg;                                      // epilogue >= 10: the decode GEMM (csrc/decode_gemm.hip)
    const bool use_dg = epilogue >= 10;
    if (use_dg) {
      const int e = epilogue - 10;
      float* cs = (float*)t.alloc((size_t)N * 4);
      dg.A = da; dg.lda = K; dg.ldw = K; dg.M = M; dg.N = N; dg.K = K; dg.bias = bias;
      dg.ws = g.sk_ws; dg.ws_bytes = g.sk_ws_bytes; dg.cnt = g.sk_cnt;
      if (e == 0) { dg.out_lo = olo; dg.ld_out_lo = N; }
      else if (e == 1) { dg.out_lo = olo; dg.ld_out_lo = N; dg.act = ACT_GELU_ERF; dg.colsum = cs; }
      else if (e == 2) { dg.add = addm; dg.ld_add = N; dg.out_f32 = of32; dg.ld_out_f32 = N; dg.out_lo = olo; dg.ld_out_lo = N; }
      else if (e == 3) { dg.out_lo = olo; dg.ld_out_lo = N; dg.colsum = cs; }
      else ASR_THROW(ASR_ERR_INVALID, "probe_gemm_chain: unknown decode epilogue %d", e);
      int nt = 1, sp = 1;
      decode_gemm_plan(dg, &nt, &sp);
      snprintf(g_chain_kernel, sizeof(g_chain_kernel), "decode nt%d ks%d", nt, sp);
    }
    // ASR_PROBE_PREFETCH=1: while node i runs, a side branch touches node i + 1's weights from the workgroups that will stream them (launch_decode_gemm_prefetch)
    const bool prefetch = use_dg && getenv("ASR_PROBE_PREFETCH") && getenv("ASR_PROBE_PREFETCH")[0] == '1';
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> evs;
    if (prefetch) {
      HIP_CHECK(hipStreamCreate(&s2));
      evs.resize(chain + 1);
      for (auto& e : evs) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // ASR_PROBE_CLK=1: every launch of the chain stamps its phase clock (decode_gemm_kernel: DecGemmArgs::dbg_clk); the breakdown of the last replay goes to stderr
    const bool clocked = use_dg && getenv("ASR_PROBE_CLK") && getenv("ASR_PROBE_CLK")[0] == '1';
    unsigned long long* dclk = clocked ? (unsigned long long*)t.alloc((size_t)chain * 10 * 8) : nullptr;
    if (dclk) HIP_CHECK(hipMemset(dclk, 0, (size_t)chain * 10 * 8));
    auto enqueue = [&] {
      for (int i = 0; i < chain; ++i) {
        if (prefetch) {
          DecGemmArgs gn = dg; gn.W = (const bf16_t*)((char*)dw + wbytes * ((i + 1) % copies));
          HIP_CHECK(hipEventRecord(evs[i], s));
          HIP_CHECK(hipStreamWaitEvent(s2, evs[i], 0));
          launch_decode_gemm_prefetch(gn, s2);
        }
        if (use_dg) { DecGemmArgs gi = dg; gi.W = (const bf16_t*)((char*)dw + wbytes * (i % copies)); gi.dbg_clk = dclk ? dclk + (size_t)i * 10 : nullptr; launch_decode_gemm(gi, s); }
        else { GemmArgs gi = g; gi.W = (char*)dw + wbytes * (i % copies); launch_gemm_bf16(gi, s); }
      }
      if (prefetch) { HIP_CHECK(hipEventRecord(evs[chain], s2)); HIP_CHECK(hipStreamWaitEvent(s, evs[chain], 0)); }
    };
    enqueue();                                                 // eager once (lazy attributes)
    HIP_CHECK(hipStreamSynchronize(s));
    if (s2) HIP_CHECK(hipStreamSynchronize(s2));
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    HIP_CHECK(hipStreamEndCapture(s, &graph));
    HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipGraphLaunch(exec, s));
    HIP_CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < replays; ++r) HIP_CHECK(hipGraphLaunch(exec, s));
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *us_per_launch = ms * 1e3f / ((float)replays * chain);
    if (dclk) {
      std::vector<unsigned long long> h((size_t)chain * 10);
      HIP_CHECK(hipMemcpy(h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost));
      double seg[4] = {0, 0, 0, 0}, body = 0, gap = 0, skew = 0, tail = 0;
      int n = 0;
      for (int i = 1; i + 1 < chain; ++i) {            // 100 MHz clock: 0.01 us per tick
        const unsigned long long* a = &h[(size_t)i * 10];
        const unsigned long long* b = &h[(size_t)(i + 1) * 10];
        if (!a[0] || !a[4] || !b[0] || !a[5] || !a[9]) continue;
        for (int q = 0; q < 4; ++q) seg[q] += (double)(a[q + 1] - a[q]) * 0.01;
        body += (double)(a[4] - a[0]) * 0.01;
        skew += (double)((long long)a[5] - (long long)a[0]) * 0.01;                    // the last workgroup starts this much after the first
        tail += (double)((long long)std::max(a[9], a[4]) - (long long)a[4]) * 0.01;    // ... and ends this much after the first one's end
        gap += (double)((long long)std::min(b[0], b[5]) - (long long)std::max(a[4], a[9])) * 0.01;   // last store acknowledged -> first instruction of the next launch
        ++n;
      }
      if (n) fprintf(stderr, "  [clock M=%d N=%d K=%d %s] launch %.2f us = body of workgroup 0 %.2f (setup %.2f, weights + MFMA %.2f, reduce %.2f, hand-over + epilogue + store ack %.2f) + last workgroup start skew %.2f / end tail %.2f; boundary to the next launch %.2f\n",
                     M, N, K, g_chain_kernel, *us_per_launch, body / n, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, skew / n, tail / n, gap / n);
    }
    if (!use_dg) snprintf(g_chain_kernel, sizeof(g_chain_kernel), "%s", gemm_last_kernel());
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    if (s2) (void)hipStreamDestroy(s2);
    for (auto& e : evs) (void)hipEventDestroy(e);
  });
}

// ---- FP8 mode (ASR_PRECISION_FP8W): the row quantiser and the decode GEMM over byte weights, on host arrays
extern "C" int asr_probe_quantize_fp8(const uint16_t* w_bf16, int N, int K, uint8_t* out8, float* scale, uint16_t* dq_bf16) {
  return asr_guard([&] {
    ASR_REQUIRE(w_bf16 && out8 && scale && N > 0 && K > 0 && K % 8 == 0, "probe_quantize_fp8: bad argument");
    asr_require_device(0);
    DeviceBuffer w, q, sc, dq;
    w.reserve((size_t)N * K * 2, nullptr); q.reserve((size_t)N * K, nullptr); sc.reserve((size_t)N * 4, nullptr); dq.reserve((size_t)N * K * 2, nullptr);
    HIP_CHECK(hipMemcpy(w.ptr, w_bf16, (size_t)N * K * 2, hipMemcpyHostToDevice));
    launch_quantize_rows_fp8(w.as<bf16_t>(), K, N, K, q.as<unsigned char>(), sc.as<float>(), dq.as<bf16_t>(), nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out8, q.ptr, (size_t)N * K, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(scale, sc.ptr, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (dq_bf16) HIP_CHECK(hipMemcpy(dq_bf16, dq.ptr, (size_t)N * K * 2, hipMemcpyDeviceToHost));
    w.release(); q.release(); sc.release(); dq.release();
  });
}

// out[M][N] (f32) = decode GEMM of a[M][K] (bf16) with EITHER w_bf16[N][K] OR (w8[N][K], scale[N]); fold != 0: LayerNorm folded in
// (column sums taken from w_bf16, which a byte-weight caller passes as the dequantised copy); bias may be null
extern "C" int asr_probe_decode_gemm(int M, int N, int K, const uint16_t* a, const uint16_t* w_bf16, const uint8_t* w8, const float* scale,
                                     const float* bias, int fold, float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(a && out && (w_bf16 || w8) && !(w8 && !scale) && !(fold && !w_bf16), "probe_decode_gemm: bad argument");
    asr_require_device(0);
    DeviceBuffer da, dw, dw8, dsc, db, dcs, dout, ws, cnt;
    da.reserve((size_t)64 * K * 2, nullptr); dout.reserve((size_t)64 * N * 4, nullptr); ws.reserve((size_t)16 << 20, nullptr); cnt.reserve(4096 * 4, nullptr);
    HIP_CHECK(hipMemset(da.ptr, 0, (size_t)64 * K * 2)); HIP_CHECK(hipMemset(cnt.ptr, 0, 4096 * 4));
    HIP_CHECK(hipMemcpy(da.ptr, a, (size_t)M * K * 2, hipMemcpyHostToDevice));
    DecGemmArgs g;
    g.A = da.as<bf16_t>(); g.lda = K; g.ldw = K; g.M = M; g.N = N; g.K = K; g.out_f32 = dout.as<float>(); g.ld_out_f32 = N;
    g.ws = ws.as<float>(); g.ws_bytes = ws.cap; g.cnt = cnt.as<int32_t>();
    if (w_bf16) { dw.reserve((size_t)N * K * 2, nullptr); HIP_CHECK(hipMemcpy(dw.ptr, w_bf16, (size_t)N * K * 2, hipMemcpyHostToDevice)); g.W = dw.as<bf16_t>(); }
    if (w8) {
      dw8.reserve((size_t)N * K, nullptr); dsc.reserve((size_t)N * 4, nullptr);
      HIP_CHECK(hipMemcpy(dw8.ptr, w8, (size_t)N * K, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dsc.ptr, scale, (size_t)N * 4, hipMemcpyHostToDevice));
      g.W = nullptr; g.W8 = dw8.as<unsigned char>(); g.w_scale = dsc.as<float>();
    }
    if (bias) { db.reserve((size_t)N * 4, nullptr); HIP_CHECK(hipMemcpy(db.ptr, bias, (size_t)N * 4, hipMemcpyHostToDevice)); g.bias = db.as<float>(); }
    if (fold) { dcs.reserve((size_t)N * 4, nullptr); launch_colsum_bf16(dw.as<bf16_t>(), K, N, K, dcs.as<float>(), nullptr); g.colsum = dcs.as<float>(); }
    launch_decode_gemm(g, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, dout.ptr, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    for (DeviceBuffer* b : {&da, &dw, &dw8, &dsc, &db, &dcs, &dout, &ws, &cnt}) b->release();
  });
}

// ---- MXFP4 mode (ASR_PRECISION_MXFP4W): the block quantiser and the decode GEMM over nibble weights, on host arrays
extern "C" int asr_probe_quantize_mxfp4(const uint16_t* w_bf16, int N, int K, uint8_t* out4, uint8_t* scale8, uint16_t* dq_bf16) {
  return asr_guard([&] {
    ASR_REQUIRE(w_bf16 && out4 && scale8 && N > 0 && K > 0 && K % 32 == 0, "probe_quantize_mxfp4: bad argument");
    asr_require_device(0);
    DeviceBuffer w, q, sc, dq;
    w.reserve((size_t)N * K * 2, nullptr); q.reserve((size_t)N * K / 2, nullptr); sc.reserve((size_t)N * K / 32, nullptr); dq.reserve((size_t)N * K * 2, nullptr);
    HIP_CHECK(hipMemcpy(w.ptr, w_bf16, (size_t)N * K * 2, hipMemcpyHostToDevice));
    launch_quantize_rows_mxfp4(w.as<bf16_t>(), K, N, K, q.as<unsigned char>(), sc.as<unsigned char>(), dq.as<bf16_t>(), nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out4, q.ptr, (size_t)N * K / 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(scale8, sc.ptr, (size_t)N * K / 32, hipMemcpyDeviceToHost));
    if (dq_bf16) HIP_CHECK(hipMemcpy(dq_bf16, dq.ptr, (size_t)N * K * 2, hipMemcpyDeviceToHost));
    w.release(); q.release(); sc.release(); dq.release();
  });
}

// out[M][N] (f32) = decode GEMM of a[M][K] (bf16) with the MXFP4 weights (w4 [N][K / 2], scale8 [N][K / 32]); fold != 0: LayerNorm folded in, column sums
// taken from w_dq (the dequantised bf16 copy, as the session does)
extern "C" int asr_probe_decode_gemm_mxfp4(int M, int N, int K, const uint16_t* a, const uint8_t* w4, const uint8_t* scale8, const uint16_t* w_dq,
                                           const float* bias, int fold, float* out) {
  return asr_guard([&] {
    ASR_REQUIRE(a && out && w4 && scale8 && !(fold && !w_dq), "probe_decode_gemm_mxfp4: bad argument");
    asr_require_device(0);
    DeviceBuffer da, dw, dw4, dsc, db, dcs, dout, ws, cnt;
    da.reserve((size_t)64 * K * 2, nullptr); dout.reserve((size_t)64 * N * 4, nullptr); ws.reserve((size_t)16 << 20, nullptr); cnt.reserve(4096 * 4, nullptr);
    HIP_CHECK(hipMemset(da.ptr, 0, (size_t)64 * K * 2)); HIP_CHECK(hipMemset(cnt.ptr, 0, 4096 * 4));
    HIP_CHECK(hipMemcpy(da.ptr, a, (size_t)M * K * 2, hipMemcpyHostToDevice));
    DecGemmArgs g;
    g.A = da.as<bf16_t>(); g.lda = K; g.ldw = K; g.M = M; g.N = N; g.K = K; g.out_f32 = dout.as<float>(); g.ld_out_f32 = N;
    g.ws = ws.as<float>(); g.ws_bytes = ws.cap; g.cnt = cnt.as<int32_t>();
    dw4.reserve((size_t)N * K / 2, nullptr); dsc.reserve((size_t)N * K / 32, nullptr);
    HIP_CHECK(hipMemcpy(dw4.ptr, w4, (size_t)N * K / 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dsc.ptr, scale8, (size_t)N * K / 32, hipMemcpyHostToDevice));
    g.W4 = dw4.as<unsigned char>(); g.w_scale4 = dsc.as<unsigned char>();
    if (bias) { db.reserve((size_t)N * 4, nullptr); HIP_CHECK(hipMemcpy(db.ptr, bias, (size_t)N * 4, hipMemcpyHostToDevice)); g.bias = db.as<float>(); }
    if (fold) {
      dw.reserve((size_t)N * K * 2, nullptr); HIP_CHECK(hipMemcpy(dw.ptr, w_dq, (size_t)N * K * 2, hipMemcpyHostToDevice));
      dcs.reserve((size_t)N * 4, nullptr); launch_colsum_bf16(dw.as<bf16_t>(), K, N, K, dcs.as<float>(), nullptr); g.colsum = dcs.as<float>();
    }
    launch_decode_gemm(g, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, dout.ptr, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    for (DeviceBuffer* b : {&da, &dw, &dw4, &dsc, &db, &dcs, &dout, &ws, &cnt}) b->release();
  });
}

// FP8 matrix-pipe GEMM on host arrays (csrc/gemm_fp8.hip): a8 [M][K], w8 [N][K] e4m3 bytes, w_scale [N], bias [N]; either out8 [M][N] (act, bytes) or
// out_f32 [M][N] (+ add [M][N]). iters > 0: also times that many launches (microseconds per launch in *us).
extern "C" int asr_probe_gemm_fp8(int M, int N, int K, const uint8_t* a8, const uint8_t* w8, const float* w_scale, float a_scale, const float* bias,
                                  const float* add, int act, uint8_t* out8, float* out_f32, int iters, float* us) {
  return asr_guard([&] {
    ASR_REQUIRE(a8 && w8 && w_scale && bias && (out8 || out_f32), "probe_gemm_fp8: bad argument");
    asr_require_device(0);
    DeviceBuffer da, dw, dsc, db, dadd, dout;
    da.reserve((size_t)M * K, nullptr); dw.reserve((size_t)N * K, nullptr); dsc.reserve((size_t)N * 4, nullptr); db.reserve((size_t)N * 4, nullptr);
    HIP_CHECK(hipMemcpy(da.ptr, a8, (size_t)M * K, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dw.ptr, w8, (size_t)N * K, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dsc.ptr, w_scale, (size_t)N * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(db.ptr, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    Fp8GemmArgs g;
    g.A = da.as<unsigned char>(); g.lda = K; g.W = dw.as<unsigned char>(); g.ldw = K; g.M = M; g.N = N; g.K = K; g.w_scale = dsc.as<float>(); g.a_scale = a_scale;
    g.bias = db.as<float>(); g.act = act;
    if (out8) { dout.reserve((size_t)M * N, nullptr); g.out8 = dout.as<unsigned char>(); g.ld_out8 = N; }
    else {
      ASR_REQUIRE(add, "probe_gemm_fp8: the f32 output needs the residual rows");
      dadd.reserve((size_t)M * N * 4, nullptr); HIP_CHECK(hipMemcpy(dadd.ptr, add, (size_t)M * N * 4, hipMemcpyHostToDevice));
      dout.reserve((size_t)M * N * 4, nullptr); g.add = dadd.as<float>(); g.ld_add = N; g.out_f32 = dout.as<float>(); g.ld_out_f32 = N;
    }
    launch_gemm_fp8(g, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (out8) HIP_CHECK(hipMemcpy(out8, dout.ptr, (size_t)M * N, hipMemcpyDeviceToHost));
    else HIP_CHECK(hipMemcpy(out_f32, dout.ptr, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    if (iters > 0 && us) {
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) launch_gemm_fp8(g, nullptr);
      HIP_CHECK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) launch_gemm_fp8(g, nullptr);
      HIP_CHECK(hipEventRecord(e1, nullptr));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      *us = ms * 1e3f / iters;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    for (DeviceBuffer* b : {&da, &dw, &dsc, &db, &dadd, &dout}) b->release();
  });
}

extern "C" const char* asr_probe_last_kernel(void) { return g_chain_kernel; }

extern "C" int asr_probe_gemm_counts(int reset, char* buf, int cap) {
  return asr_guard([&] {
    if (buf && cap > 0) gemm_kernel_counts(buf, cap);
    if (reset) gemm_kernel_counts_reset();
  });
}

// Parity hook for the epilogue families the product's batch-64 path uses and asr_op_gemm cannot express: the LayerNorm folded into
// the product (row statistics handed over in 32-column groups, column sums of the bf16 weights), the fused row arg-max, and the
// producer epilogue (f32 + bf16 stores + row statistics). Reports which kernel family the dispatcher chose.
extern "C" int asr_probe_gemm(asr_probe_gemm_desc* d) {
  return asr_guard([&] {
    ASR_REQUIRE(d && d->a && d->w && d->M > 0 && d->N > 0 && d->K > 0 && d->N % 128 == 0 && d->K % 64 == 0, "probe_gemm: bad argument");
    asr_require_device(0);
    gemm_reload_env();
    Tmp t;
    const int M = d->M, N = d->N, K = d->K, Mp = round_up(M, 128);
    std::vector<bf16_t> ha((size_t)Mp * K, 0), hw((size_t)N * K);
    for (size_t i = 0; i < (size_t)M * K; ++i) ha[i] = f32_to_bf16(d->a[i]);
    for (size_t i = 0; i < (size_t)N * K; ++i) hw[i] = f32_to_bf16(d->w[i]);
    bf16_t* da = (bf16_t*)t.alloc(ha.size() * 2);
    bf16_t* dw = (bf16_t*)t.alloc(hw.size() * 2);
    HIP_CHECK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A = da; g.lda = K; g.W = dw; g.ldw = K; g.M = M; g.N = N; g.K = K; g.act = d->act;
    if (d->bias) {
      float* db = (float*)t.alloc((size_t)N * 4);
      HIP_CHECK(hipMemcpy(db, d->bias, (size_t)N * 4, hipMemcpyHostToDevice));
      g.bias = db;
    }
    g.sk_ws = (float*)t.alloc((size_t)16 << 20); g.sk_ws_bytes = (size_t)16 << 20; g.sk_cnt = (int32_t*)t.alloc(4096 * 4);
    if (d->ln) {            // LayerNorm(no affine, eps) of the bf16-rounded A rows folded into the product
      std::vector<float> cs(N);
      for (int n = 0; n < N; ++n) { double s = 0.0; for (int k = 0; k < K; ++k) s += bf16_bits_to_f32(hw[(size_t)n * K + k]); cs[n] = (float)s; }
      std::vector<float> st((size_t)Mp * (K / 32) * 2, 0.0f);
      for (int m = 0; m < M; ++m)
        for (int q = 0; q < K / 32; ++q) {
          float s1 = 0.0f, s2 = 0.0f;
          for (int k = 0; k < 32; ++k) { const float v = bf16_bits_to_f32(ha[(size_t)m * K + q * 32 + k]); s1 += v; s2 += v * v; }
          st[((size_t)m * (K / 32) + q) * 2] = s1; st[((size_t)m * (K / 32) + q) * 2 + 1] = s2;
        }
      float* dcs = (float*)t.alloc((size_t)N * 4);
      float2* dst = (float2*)t.alloc(st.size() * 4);
      HIP_CHECK(hipMemcpy(dcs, cs.data(), (size_t)N * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice));
      g.ln_colsum = dcs; g.ln_dim = K; g.ln_stats_in = dst; g.ln_slots = K / 32; g.ln_eps = d->ln_eps;
      ASR_REQUIRE(gemm_ln_fusable(g), "probe_gemm: shape cannot take the LayerNorm-folded kernels");
    }
    float* d_add = nullptr;
    if (d->add) {
      d_add = (float*)t.alloc((size_t)Mp * N * 4);
      HIP_CHECK(hipMemcpy(d_add, d->add, (size_t)M * N * 4, hipMemcpyHostToDevice));
      g.add = d_add; g.ld_add = N;
    }
    bf16_t* dlo = nullptr; float* df32 = nullptr; float2* dso = nullptr; float* dav = nullptr; int32_t* dai = nullptr; int32_t* dids = nullptr;
    const int n_slabs = N / 64;
    if (d->argmax) {
      dav = (float*)t.alloc((size_t)Mp * n_slabs * 4); dai = (int32_t*)t.alloc((size_t)Mp * n_slabs * 4); dids = (int32_t*)t.alloc((size_t)Mp * 4);
      g.amax_val = dav; g.amax_idx = dai; g.n_valid = d->n_valid > 0 ? d->n_valid : N;
    } else {
      if (d->out_lo) { dlo = (bf16_t*)t.alloc((size_t)Mp * N * 2); g.out_lo = dlo; g.ld_out_lo = N; }
      if (d->out_f32) { df32 = (float*)t.alloc((size_t)Mp * N * 4); g.out_f32 = df32; g.ld_out_f32 = N; }
      if (d->out_stats) { ASR_REQUIRE(d->out_lo, "probe_gemm: row statistics describe the bf16 output"); dso = (float2*)t.alloc((size_t)Mp * (N / 32) * 8); g.st_out = dso; }
    }
    gemm_set_variant(d->variant);
    try { launch_gemm_bf16(g, nullptr); } catch (...) { gemm_set_variant(-1); throw; }
    gemm_set_variant(-1);
    snprintf(d->kernel, sizeof(d->kernel), "%s", gemm_last_kernel());
    if (d->argmax) launch_argmax_reduce(dav, dai, M, n_slabs, dids, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (d->argmax) {
      ASR_REQUIRE(d->out_ids, "probe_gemm: out_ids missing");
      HIP_CHECK(hipMemcpy(d->out_ids, dids, (size_t)M * 4, hipMemcpyDeviceToHost));
    }
    if (dlo) {
      std::vector<bf16_t> tmp((size_t)M * N);
      HIP_CHECK(hipMemcpy(tmp.data(), dlo, tmp.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < tmp.size(); ++i) d->out_lo[i] = bf16_bits_to_f32(tmp[i]);
    }
    if (df32) HIP_CHECK(hipMemcpy(d->out_f32, df32, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    if (dso) HIP_CHECK(hipMemcpy(d->out_stats, dso, (size_t)M * (N / 32) * 8, hipMemcpyDeviceToHost));
  });
}

// ---- grid-barrier latency probe (tuning hook): a cooperative launch of one workgroup per CU crossing `iters` barriers.
namespace {
__device__ __forceinline__ bool grid_barrier_probe_step(unsigned int* counter, unsigned int target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = false; break; }       // never hang the box: give up after ~1 s
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}
__global__ __launch_bounds__(512) void grid_barrier_probe_kernel(unsigned int* counter, int iters, float* sink, int* failed) {
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    acc += sink[(blockIdx.x * 64 + (it & 63)) & 4095];          // a little cross-workgroup traffic between barriers
    if (threadIdx.x == 0) sink[(blockIdx.x * 64 + ((it + 1) & 63)) & 4095] = acc * 0.5f;
    if (!grid_barrier_probe_step(counter, (unsigned int)(it + 1) * gridDim.x)) { if (threadIdx.x == 0) *failed = 1; return; }
  }
  if (acc == 123.456f) sink[0] = acc;
}
}  // namespace

namespace {
// hierarchical barrier probe: arrivals are counted per XCD (workgroup b runs on XCD b % 8), the last arrival of an XCD counts
// itself on the chip counter, the last XCD publishes the generation. Relaxed agent-scope atomics only -- no fence, i.e. no L2
// write-back / invalidate; payload that must cross XCDs would travel through sc1 accesses.
struct HBar { unsigned int* xcd; unsigned int* chip; unsigned int* flag; };   // xcd[8 * 32], flag[8 * 32] (128-byte spacing)
__device__ __forceinline__ bool hbar_step(const HBar& hb, unsigned int gen, int mode) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    const unsigned int in_xcd = (gridDim.x + 7 - x) >> 3;
    const unsigned int old = __hip_atomic_fetch_add(hb.xcd + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gen * in_xcd) {
      const unsigned int n_xcd = gridDim.x < 8 ? gridDim.x : 8;
      const unsigned int o2 = __hip_atomic_fetch_add(hb.chip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 + 1 == gen * n_xcd) {
        if (mode == 2) { for (int q = 0; q < 8; ++q) __hip_atomic_store(hb.flag + q * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else __hip_atomic_store(hb.flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const unsigned int* f = mode == 2 ? hb.flag + x * 32 : hb.flag;
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}
__global__ __launch_bounds__(512) void hbar_probe_kernel(HBar hb, int iters, int mode, float* sink, int* failed, const unsigned long long* bulk,
                                                         int bulk_words, int bulk_sc1) {
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    // bulk activation read every workgroup does per phase: `bulk_words` 8-byte words of ONE shared buffer, through sc1 (relaxed
    // agent-scope atomic loads: what a persistent kernel must use for data produced by other XCDs) or through plain cached loads
    unsigned long long bsum = 0;
    if (bulk_sc1) { for (int w = threadIdx.x; w < bulk_words; w += 512) bsum += __hip_atomic_load(bulk + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { for (int w = threadIdx.x; w < bulk_words; w += 512) bsum += bulk[w]; }
    if (bsum == 0x123456789abcdefull) acc += 1.0f;
    // payload through sc1 accesses: one value written by this workgroup, one read from the neighbour's slot of the previous round
    const float v = __hip_atomic_load(sink + ((blockIdx.x + 1) % gridDim.x) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (it > 0 && v != (float)it) { if (threadIdx.x == 0) *failed = 2; }      // stale payload => visibility bug
    acc += v;
    if (!hbar_step(hb, 2u * it + 1u, mode)) { if (threadIdx.x == 0) *failed = 1; return; }
    if (threadIdx.x == 0) __hip_atomic_store(sink + blockIdx.x * 32, (float)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!hbar_step(hb, 2u * it + 2u, mode)) { if (threadIdx.x == 0) *failed = 1; return; }
  }
  if (acc == 123.456f) sink[0] = acc;
}
}  // namespace

extern "C" int asr_probe_grid_barrier2(int n_workgroups, int iters, int mode, float* us_per_barrier) {
  return asr_guard([&] {
    // mode = (1 | 2) + 16 * bulk KiB read per workgroup per round + 8 if that read goes through plain cached loads instead of sc1
    int bulk_kib = mode >> 4, bulk_sc1 = (mode & 8) ? 0 : 1;
    mode &= 7;
    ASR_REQUIRE(us_per_barrier && iters > 0 && n_workgroups > 0 && n_workgroups <= 1024 && (mode == 1 || mode == 2) && bulk_kib <= 1024, "probe_grid_barrier2: bad argument");
    asr_require_device(0);
    Tmp t;
    unsigned int* ctr = (unsigned int*)t.alloc(3 * 8 * 32 * 4);
    float* sink = (float*)t.alloc(1024 * 32 * 4);
    int* failed = (int*)t.alloc(256);
    HIP_CHECK(hipMemset(ctr, 0, 3 * 8 * 32 * 4));
    HIP_CHECK(hipMemset(sink, 0, 1024 * 32 * 4));
    HIP_CHECK(hipMemset(failed, 0, 256));
    HBar hb{ctr, ctr + 8 * 32, ctr + 2 * 8 * 32};
    const unsigned long long* bulk = (const unsigned long long*)t.alloc((size_t)std::max(bulk_kib, 1) * 1024);
    HIP_CHECK(hipMemset((void*)bulk, 1, (size_t)std::max(bulk_kib, 1) * 1024));
    int bulk_words = bulk_kib * 128;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    void* args[] = {&hb, &iters, &mode, &sink, &failed, &bulk, &bulk_words, &bulk_sc1};
    HIP_CHECK(hipEventRecord(e0, nullptr));
    HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(hbar_probe_kernel), dim3(n_workgroups), dim3(512), args, 0, nullptr));
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    int hf = 0;
    HIP_CHECK(hipMemcpy(&hf, failed, 4, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ASR_REQUIRE(hf != 1, "probe_grid_barrier2: a workgroup gave up waiting (grid not co-resident?)");
    ASR_REQUIRE(hf != 2, "probe_grid_barrier2: a payload written before the barrier was not visible after it");
    *us_per_barrier = ms * 1e3f / (2.0f * iters);
  });
}

extern "C" int asr_probe_grid_barrier(int n_workgroups, int iters, float* us_per_barrier) {
  return asr_guard([&] {
    ASR_REQUIRE(us_per_barrier && iters > 0 && n_workgroups > 0, "probe_grid_barrier: bad argument");
    asr_require_device(0);
    Tmp t;
    unsigned int* counter = (unsigned int*)t.alloc(256);
    float* sink = (float*)t.alloc(4096 * 4);
    int* failed = (int*)t.alloc(256);
    HIP_CHECK(hipMemset(counter, 0, 256));
    HIP_CHECK(hipMemset(sink, 0, 4096 * 4));
    HIP_CHECK(hipMemset(failed, 0, 256));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    void* args[] = {&counter, &iters, &sink, &failed};
    HIP_CHECK(hipEventRecord(e0, nullptr));
    HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(grid_barrier_probe_kernel), dim3(n_workgroups), dim3(512), args, 0, nullptr));
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    int hf = 0;
    HIP_CHECK(hipMemcpy(&hf, failed, 4, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ASR_REQUIRE(!hf, "probe_grid_barrier: a workgroup gave up waiting (grid not co-resident?)");
    *us_per_barrier = ms * 1e3f / iters;
  });
}
