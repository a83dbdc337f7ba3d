// Qwen3-ASR decode step as ONE persistent kernel (bf16 mode, <= 64 sequences): token embedding, then per decoder layer
//   P1 q|k|v = rstd(x) (x W_qkv^T)            256 column granules of 16, one per workgroup
//   P2 per (sequence, kv head): q / k RMSNorm + RoPE + cache append + GQA attention over the cache (two items per workgroup)
//   P3 x2 = x + ctx W_o^T                     64 granules
//   P4 act = SwiGLU(rstd(x2) (x2 W_gu^T))     384 granules
//   P5 x = x2 + act W_down^T                  64 granules
// with a chip-wide barrier between phases instead of a kernel boundary (DECODER_MAIN.forward, Export_Qwen_ASR.py:1265-1336).
// Motivation: as separate launches every phase costs 8..13 us although its weights stream in 1..2 us (tools/trace_summary.py on the
// decode step); a chip-wide barrier costs 2 us (tools/grid_barrier_probe.py).
//  * Barrier: arrivals counted per XCD (workgroup b runs on XCD b % 8), the last arrival of an XCD counts itself on the chip
//    counter, the last XCD publishes the generation to one flag per XCD. Relaxed agent-scope atomics only: no fence, because a
//    release / acquire fence writes back and invalidates a whole L2 on this part. Generations increase monotonically across
//    launches (the host passes the base), so the counters are never reset.
//  * Coherence: XCD L2s are not coherent with each other, so everything one phase writes for another workgroup to read is written
//    with sc1 stores (relaxed agent-scope atomics on 4- / 8-byte words: written through to the memory side) and every thread drains
//    its stores (s_waitcnt vmcnt(0)) before it arrives at the barrier. Readers: data read once per element (f32 residual stream,
//    q|k|v) comes back through sc1 loads. The GEMM operands that EVERY workgroup reads (bf16 residual copy, attention context, FFN
//    activations) would cost 4 x the weight traffic through sc1 (measured: 13 / 21 / 31 us per phase), so they live in one buffer
//    per layer, written exactly once per launch before anyone reads them: ordinary cached loads are then safe (no L2 can hold a
//    stale line of an address that nobody has read since the launch began) and each XCD fetches the operand once.
//    Weights, rotary table, history counters are read-only here: plain loads. The KV cache rows of earlier positions were written by
//    earlier launches; the new row is consumed from LDS by the workgroup that writes it.
//  * GEMM phases reuse the weight-streaming scheme of gemm_bf16_skinny: 8 waves split K, weight fragments straight from HBM into
//    MFMA operands, sum(x^2) of the RMSNorm from A A^T diagonals, cross-wave reduction through LDS, swapped orientation (4
//    consecutive columns per lane) for the epilogues.
// Measured (MI355X, 0.6B geometry, 64 sequences, ~140 cached positions; ASR_QWEN_MEGA_DBG=1 prints the phase clock): per layer
// q|k|v 10 / attention 14 / o_proj 14 / gate|up 15.5 / down 19 us including the 2 us barrier = 72 us, against 8.5 / 12.4 / 10.6 / 12.8 /
// 12.6 us = 57 us (+ ~4 us of gaps) for the same phases as separate launches inside a hipGraph. A phase is a latency chain --
// operand fetch from the memory side (the producers wrote it a moment ago, so no L2 has it), MFMA + LDS reduction, store
// acknowledgement, barrier -- that a kernel boundary does not make much longer than a barrier does, and the separate launches
// split K across workgroups for the two narrow projections. The kernel therefore stays OPT-IN (ASR_QWEN_MEGA=1); it is kept, and
// covered by the parity tests, as the base for the next step: prefetching the next phase's weight fragments across the barrier
// and splitting K of o_proj / down_proj.
// Spins are bounded: a workgroup that waits ~1 s sets `failed` and leaves, so a scheduling accident cannot hang the device.
#include "kernels.h"

namespace {

constexpr int MG_WAVES = 8;
constexpr int HD = 128;

__device__ __forceinline__ uint64_t ld_sc1_u64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u64(void* p, uint64_t v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u32(void* p, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t pack_f32x2(float a, float b) { return (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32); }
__device__ __forceinline__ float lo_f32(uint64_t v) { return __uint_as_float((uint32_t)v); }
__device__ __forceinline__ float hi_f32(uint64_t v) { return __uint_as_float((uint32_t)(v >> 32)); }

__device__ __forceinline__ float dpp_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

// chip-wide barrier; false when this workgroup gave up waiting
__device__ __forceinline__ bool mega_barrier(const QwMegaArgs& a, unsigned int gen) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's sc1 stores have reached the memory side
  __syncthreads();
  __shared__ int ok_sh;
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    const unsigned int in_xcd = (gridDim.x + 7 - x) >> 3, n_xcd = gridDim.x < 8 ? gridDim.x : 8;
    const unsigned int old = __hip_atomic_fetch_add(a.bar_xcd + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gen * in_xcd) {
      const unsigned int o2 = __hip_atomic_fetch_add(a.bar_chip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 + 1 == gen * n_xcd)
        for (int q = 0; q < 8; ++q) __hip_atomic_store(a.bar_flag + q * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spins = 0, ok = 1;
    while (__hip_atomic_load(a.bar_flag + x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = 0; break; }
    }
    if (!ok) *a.failed = 1;
    ok_sh = ok;
    if (a.dbg_clock && blockIdx.x == 0) a.dbg_clock[gen - a.gen_base] = wall_clock64();
  }
  __syncthreads();
  return ok_sh != 0;
}

// One 16-column granule: sums[t] (wave w finishes row tile w + 8 t) = A[M][K] W[n0 .. n0+16)[K]^T for this lane's (row frow, columns
// n0 + 4 fgrp ..+3); with RMS also rstd of that row. A is a write-once-per-launch buffer: cached loads.
template <int MT, bool RMS>
__device__ __forceinline__ void mega_granule(const bf16_t* __restrict__ W, int ldw, int K, int n0, const bf16_t* A, int lda, float eps,
                                             unsigned char* smem, float4& sum, float& rstd, unsigned long long* dbg = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const int kslice = K / MG_WAVES, k_begin = wave * kslice;
  const bf16_t* wp = W + (size_t)(n0 + frow) * ldw + k_begin + fgrp * 8;
  const bf16_t* ap = A + (size_t)frow * lda + k_begin + fgrp * 8;
  f32x4_t acc[MT], gram[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; gram[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  constexpr int U = 4;
  for (int k = 0; k < kslice; k += 32 * U) {
    bf16x8_t wf[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k + u * 32 < kslice) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + k + u * 32));
    union AF { bf16x8_t v; uint64_t q[2]; } af[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k + u * 32 < kslice) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[u][i].v = *reinterpret_cast<const bf16x8_t*>(ap + (size_t)i * 16 * lda + k + u * 32);
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k + u * 32 < kslice) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u][i].v, acc[i], 0, 0, 0);     // D[n = 4 fgrp + r][m = frow]
          if (RMS) gram[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][i].v, af[u][i].v, gram[i], 0, 0, 0);
        }
      }
  }
  if (dbg && blockIdx.x == 0 && tid == 0) dbg[0] = wall_clock64();                                   // operands arrived, MFMAs issued
  float4* red = reinterpret_cast<float4*>(smem);                                                     // [wave][MT][64]
  float (*ss_red)[MT * 16] = reinterpret_cast<float (*)[MT * 16]>(smem + MG_WAVES * MT * 1024);
  __syncthreads();                                       // the previous granule's readers are done with `red`
#pragma unroll
  for (int i = 0; i < MT; ++i) red[(wave * MT + i) * 64 + lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  if (RMS) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (fgrp == (frow >> 2)) ss_red[wave][i * 16 + frow] = gram[i][frow & 3];
  }
  __syncthreads();
  sum = make_float4(0.f, 0.f, 0.f, 0.f);
  rstd = 1.0f;
  if (wave < MT) {
    sum = red[wave * 64 + lane];
#pragma unroll
    for (int w = 1; w < MG_WAVES; ++w) {
      const float4 q = red[(w * MT + wave) * 64 + lane];
      sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
    }
    if (RMS) {
      float t = ss_red[0][wave * 16 + frow];
#pragma unroll
      for (int w = 1; w < MG_WAVES; ++w) t += ss_red[w][wave * 16 + frow];
      rstd = rsqrtf(t / (float)K + eps);
    }
  }
}

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 r;
  __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float* o) const {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
};

struct AttnLds {
  float qsh[4][HD];
  float knew[HD], vnew[HD];
  float pm[16][4], pl[16][4];
  float pacc[16][4][HD];
};

// P2 for one (sequence, kv head) item on one half (4 waves) of the workgroup; both halves call it in lock-step (workgroup barriers
// inside), `live` = this half has an item. Same algorithm as qw_decode_attn_kernel (qwen.hip).
template <int G>
__device__ __forceinline__ void mega_attention(const QwMegaArgs& a, const QwMegaLayer& L, bf16_t* kc, bf16_t* vc, bf16_t* ctx, int item, bool live,
                                               AttnLds& S) {
  constexpr int KPI = 4, NTASK = (2 + G + 3) / 4;
  const int ht = threadIdx.x & 255, lane = ht & 63, wave = ht >> 6;
  const int b = live ? item / a.n_kv : 0, kvh = live ? item % a.n_kv : 0;
  const int pos = a.hist[b];
  const int heads = a.n_heads + 2 * a.n_kv;
  bf16_t* K = kc + ((size_t)b * a.n_kv + kvh) * a.S_max * HD;
  bf16_t* V = vc + ((size_t)b * a.n_kv + kvh) * a.S_max * HD;
  const float* row = a.qkv + (size_t)b * heads * HD;
  const int lg = lane >> 4, li = lane & 15, gid = wave * 4 + lg;
  float x0[NTASK], x1[NTASK];
#pragma unroll
  for (int t = 0; t < NTASK; ++t) {
    const int task = wave + 4 * t;
    const int hh = task == 0 ? a.n_heads + kvh : task == 1 ? a.n_heads + a.n_kv + kvh : kvh * G + (task - 2);
    x0[t] = x1[t] = 0.0f;
    if (live && task < 2 + G) { x0[t] = ld_sc1_f32(row + hh * HD + lane); x1[t] = ld_sc1_f32(row + hh * HD + lane + 64); }
  }
  Raw8<bf16_t> kb[KPI], vb[KPI];
#pragma unroll
  for (int u = 0; u < KPI; ++u) {
    const int s = u * 16 + gid;
    if (live && s < pos) { kb[u].load(K + (size_t)s * HD + li * 8); vb[u].load(V + (size_t)s * HD + li * 8); }
  }
  const float cs = a.rope[(size_t)pos * HD + lane], sn = a.rope[(size_t)pos * HD + 64 + lane];
#pragma unroll
  for (int t = 0; t < NTASK; ++t) {
    const int task = wave + 4 * t;
    if (!live || task >= 2 + G) continue;
    bf16_t t0, t1;
    if (task == 1) {
      Elem<bf16_t>::store(&t0, x0[t]);
      Elem<bf16_t>::store(&t1, x1[t]);
      V[(size_t)pos * HD + lane] = t0;
      V[(size_t)pos * HD + lane + 64] = t1;
      S.vnew[lane] = bf16_to_f32(t0);
      S.vnew[lane + 64] = bf16_to_f32(t1);
      continue;
    }
    const float r = rsqrtf(wave_sum(x0[t] * x0[t] + x1[t] * x1[t]) / (float)HD + a.eps);
    const float* w = task == 0 ? L.kn : L.qn;
    const float a0 = x0[t] * r * w[lane], a1 = x1[t] * r * w[lane + 64];
    Elem<bf16_t>::store(&t0, a0 * cs - a1 * sn);
    Elem<bf16_t>::store(&t1, a1 * cs + a0 * sn);
    float* dst = task == 0 ? S.knew : S.qsh[task - 2];
    dst[lane] = bf16_to_f32(t0);
    dst[lane + 64] = bf16_to_f32(t1);
    if (task == 0) { K[(size_t)pos * HD + lane] = t0; K[(size_t)pos * HD + lane + 64] = t1; }
  }
  __syncthreads();
  float qr[G][8], m[G], l[G], acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY; l[g] = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { qr[g][e] = S.qsh[g][li * 8 + e]; acc[g][e] = 0.0f; }
  }
  auto block = [&](const float (*k8)[8], const float (*v8)[8], const bool* valid, auto nb_tag) {
    constexpr int NB = decltype(nb_tag)::value;
    float sc[NB][G];
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float t = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(qr[g][e], k8[u][e], t);
        sc[u][g] = t;
      }
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) sc[u][g] = dpp_sum16(sc[u][g]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mn = m[g];
#pragma unroll
      for (int u = 0; u < NB; ++u) if (valid[u]) mn = fmaxf(mn, sc[u][g]);
      const float scale = m[g] == -INFINITY ? 0.0f : __expf(m[g] - mn);
      m[g] = mn;
      l[g] *= scale;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[g][e] *= scale;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float p = valid[u] ? __expf(sc[u][g] - mn) : 0.0f;
        l[g] += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p, v8[u][e], acc[g][e]);
      }
    }
  };
  const int n_old = live ? pos : 0;
  for (int s0 = 0; s0 < n_old; s0 += 16 * KPI) {
    Raw8<bf16_t> kn2[KPI], vn2[KPI];
#pragma unroll
    for (int u = 0; u < KPI; ++u) {
      const int s = s0 + (KPI + u) * 16 + gid;
      if (s < n_old) { kn2[u].load(K + (size_t)s * HD + li * 8); vn2[u].load(V + (size_t)s * HD + li * 8); }
    }
    float k8[KPI][8], v8[KPI][8];
    bool valid[KPI];
#pragma unroll
    for (int u = 0; u < KPI; ++u) {
      valid[u] = s0 + u * 16 + gid < n_old;
      if (valid[u]) { kb[u].unpack(k8[u]); vb[u].unpack(v8[u]); }
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k8[u][e] = 0.0f; v8[u][e] = 0.0f; }
      }
    }
    block(k8, v8, valid, std::integral_constant<int, KPI>{});
#pragma unroll
    for (int u = 0; u < KPI; ++u) { kb[u] = kn2[u]; vb[u] = vn2[u]; }
  }
  {
    float k8[1][8], v8[1][8];
    const bool valid[1] = {gid == 0 && live};
#pragma unroll
    for (int e = 0; e < 8; ++e) { k8[0][e] = S.knew[li * 8 + e]; v8[0][e] = S.vnew[li * 8 + e]; }
    block(k8, v8, valid, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (li == 0) { S.pm[gid][g] = m[g]; S.pl[gid][g] = l[g]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) S.pacc[gid][g][li * 8 + e] = acc[g][e];
  }
  __syncthreads();
  if (live) {
    for (int i = ht; i < G * HD / 2; i += 256) {           // two adjacent elements per thread: one 4-byte sc1 store
      const int g = i / (HD / 2), e = (i - g * (HD / 2)) * 2;
      float mx = S.pm[0][g];
#pragma unroll
      for (int q = 1; q < 16; ++q) mx = fmaxf(mx, S.pm[q][g]);
      float n0 = 0.0f, n1 = 0.0f, den = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float w = __expf(S.pm[q][g] - mx);
        n0 = fmaf(S.pacc[q][g][e], w, n0);
        n1 = fmaf(S.pacc[q][g][e + 1], w, n1);
        den = fmaf(S.pl[q][g], w, den);
      }
      st_sc1_u32(ctx + (size_t)b * a.n_heads * HD + (kvh * G + g) * HD + e, pack_bf16x2(n0 / den, n1 / den));
    }
  }
  __syncthreads();                                         // LDS is reused by the next phase
}

template <int MT, int G>
__global__ __launch_bounds__(512) void qw_decode_mega_kernel(const QwMegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fgrp = lane >> 4;
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int d = a.d, I = a.d_ffn, B = a.B, qkvn = (a.n_heads + 2 * a.n_kv) * HD, ctxn = a.n_heads * HD;
  unsigned int gen = a.gen_base;
  if (a.dbg_clock && wg == 0 && tid == 0) a.dbg_clock[0] = wall_clock64();
  // ---- phase 0: x = embed[ids] (f32 residual stream + its bf16 copy)
  for (int r = wg; r < B; r += nwg) {
    const bf16_t* e = a.embed + (size_t)a.ids[r] * d;
    for (int c = tid * 2; c < d; c += 1024) {
      const uint32_t pk = *reinterpret_cast<const uint32_t*>(e + c);
      st_sc1_u64(a.x + (size_t)r * d + c, pack_f32x2(__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)));
      st_sc1_u32(a.xlo + (size_t)r * d + c, pk);          // layer 0's copy
    }
  }
  if (!mega_barrier(a, ++gen)) return;
  for (int layer = 0; layer < a.n_layers; ++layer) {
    const QwMegaLayer L = a.layers[layer];
    bf16_t* kc = a.kc + (size_t)layer * a.layer_kv;
    bf16_t* vc = a.vc + (size_t)layer * a.layer_kv;
    const bf16_t* xlo = a.xlo + (size_t)layer * a.xlo_stride;
    bf16_t* xlo_next = a.xlo + (size_t)(layer + 1) * a.xlo_stride;       // (n_layers + 1 buffers)
    bf16_t* x2lo = a.x2lo + (size_t)layer * a.xlo_stride;
    bf16_t* ctx = a.ctx + (size_t)layer * a.ctx_stride;
    bf16_t* act = a.act + (size_t)layer * a.act_stride;
    // ---- P1: q|k|v (f32)
    for (int gr = wg; gr < qkvn / 16; gr += nwg) {
      float4 s; float r;
      unsigned long long* dbg = (a.dbg_clock && layer == 1) ? a.dbg_clock + 512 : nullptr;
      if (dbg && wg == 0 && tid == 0) dbg[4] = wall_clock64();
      mega_granule<MT, true>(L.wqkv, d, d, gr * 16, xlo, d, a.eps, smem, s, r, dbg);
      if (dbg && wg == 0 && tid == 0) dbg[1] = wall_clock64();
      const int m = wave * 16 + frow, n = gr * 16 + fgrp * 4;
      if (wave < MT && m < B) {
        st_sc1_u64(a.qkv + (size_t)m * qkvn + n, pack_f32x2(s.x * r, s.y * r));
        st_sc1_u64(a.qkv + (size_t)m * qkvn + n + 2, pack_f32x2(s.z * r, s.w * r));
      }
      if (dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wg == 0 && tid == 0) dbg[2] = wall_clock64();
      }
    }
    if (!mega_barrier(a, ++gen)) return;
    // ---- P2: attention, two (sequence, kv head) items per workgroup per pass
    {
      AttnLds* S = reinterpret_cast<AttnLds*>(smem) + (tid >> 8);
      const int items = B * a.n_kv;
      for (int it0 = 0; it0 < items; it0 += 2 * nwg) {
        const int item = it0 + 2 * wg + (tid >> 8);
        mega_attention<G>(a, L, kc, vc, ctx, item, item < items, *S);
      }
    }
    if (!mega_barrier(a, ++gen)) return;
    // ---- P3: x2 = x + ctx W_o^T
    for (int gr = wg; gr < d / 16; gr += nwg) {
      float4 s; float r;
      mega_granule<MT, false>(L.wo, ctxn, ctxn, gr * 16, ctx, ctxn, 0.0f, smem, s, r);
      const int m = wave * 16 + frow, n = gr * 16 + fgrp * 4;
      if (wave < MT && m < B) {
        const uint64_t r0 = ld_sc1_u64(a.x + (size_t)m * d + n), r1 = ld_sc1_u64(a.x + (size_t)m * d + n + 2);
        const float y0 = s.x + lo_f32(r0), y1 = s.y + hi_f32(r0), y2 = s.z + lo_f32(r1), y3 = s.w + hi_f32(r1);
        st_sc1_u64(a.x2 + (size_t)m * d + n, pack_f32x2(y0, y1));
        st_sc1_u64(a.x2 + (size_t)m * d + n + 2, pack_f32x2(y2, y3));
        st_sc1_u64(x2lo + (size_t)m * d + n, (uint64_t)pack_bf16x2(y0, y1) | ((uint64_t)pack_bf16x2(y2, y3) << 32));
      }
    }
    if (!mega_barrier(a, ++gen)) return;
    // ---- P4: act = silu(gate) * up over interleaved (gate_j, up_j) columns
    for (int gr = wg; gr < 2 * I / 16; gr += nwg) {
      float4 s; float r;
      mega_granule<MT, true>(L.gate_up, d, d, gr * 16, x2lo, d, a.eps, smem, s, r);
      const int m = wave * 16 + frow, n = gr * 16 + fgrp * 4;
      if (wave < MT && m < B) {
        const float g0 = s.x * r, u0 = s.y * r, g1 = s.z * r, u1 = s.w * r;
        st_sc1_u32(act + (size_t)m * I + (n >> 1), pack_bf16x2(g0 / (1.0f + __expf(-g0)) * u0, g1 / (1.0f + __expf(-g1)) * u1));
      }
    }
    if (!mega_barrier(a, ++gen)) return;
    // ---- P5: x = x2 + act W_down^T
    for (int gr = wg; gr < d / 16; gr += nwg) {
      float4 s; float r;
      mega_granule<MT, false>(L.down, I, I, gr * 16, act, I, 0.0f, smem, s, r);
      const int m = wave * 16 + frow, n = gr * 16 + fgrp * 4;
      if (wave < MT && m < B) {
        const uint64_t r0 = ld_sc1_u64(a.x2 + (size_t)m * d + n), r1 = ld_sc1_u64(a.x2 + (size_t)m * d + n + 2);
        const float y0 = s.x + lo_f32(r0), y1 = s.y + hi_f32(r0), y2 = s.z + lo_f32(r1), y3 = s.w + hi_f32(r1);
        st_sc1_u64(a.x + (size_t)m * d + n, pack_f32x2(y0, y1));
        st_sc1_u64(a.x + (size_t)m * d + n + 2, pack_f32x2(y2, y3));
        st_sc1_u64(xlo_next + (size_t)m * d + n, (uint64_t)pack_bf16x2(y0, y1) | ((uint64_t)pack_bf16x2(y2, y3) << 32));
      }
    }
    if (layer + 1 < a.n_layers && !mega_barrier(a, ++gen)) return;
  }
  // every workgroup is past the last read of the history counters (P2 of the last layer): advance them
  if (wg == 0 && tid < B) a.hist_rw[tid] = min(a.hist_rw[tid] + 1, a.S_max - 1);      // (capped: see qw_hist_add_kernel)
}

template <int MT, int G>
void launch_inst(const QwMegaArgs& a, int n_wg, hipStream_t s) {
  const size_t lds = std::max((size_t)MG_WAVES * MT * 1024 + (size_t)MG_WAVES * MT * 64, 2 * sizeof(AttnLds));
  static bool attr = false;
  if (!attr) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qw_decode_mega_kernel<MT, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  QwMegaArgs arg = a;
  void* params[] = {&arg};
  HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(qw_decode_mega_kernel<MT, G>), dim3(n_wg), dim3(512), params, (unsigned int)lds, s));
}

}  // namespace

bool qw_decode_mega_supported(const QwMegaArgs& a) {
  const int G = a.n_kv > 0 ? a.n_heads / a.n_kv : 0;
  return a.B >= 1 && a.B <= 64 && (G == 1 || G == 2 || G == 4) && a.d % 256 == 0 && a.d_ffn % 256 == 0 && (a.n_heads * HD) % 256 == 0;
}

int qw_decode_mega_barriers(const QwMegaArgs& a) { return 5 * a.n_layers; }          // 1 + 5 per layer - 1

void launch_qw_decode_mega(const QwMegaArgs& a, hipStream_t s) {
  ASR_REQUIRE(qw_decode_mega_supported(a), "qwen decode kernel: unsupported geometry");
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipGetDeviceProperties(&p, dev));
    n_cu = p.multiProcessorCount;
  }
  const int n_wg = std::min(256, n_cu);                  // one workgroup per CU: co-resident by construction (cooperative launch checks)
  const int G = a.n_heads / a.n_kv, MT = a.B <= 16 ? 1 : a.B <= 32 ? 2 : 4;
#define QW_MEGA_CASE(MT_, G_) if (MT == MT_ && G == G_) { launch_inst<MT_, G_>(a, n_wg, s); return; }
  QW_MEGA_CASE(1, 1) QW_MEGA_CASE(1, 2) QW_MEGA_CASE(1, 4)
  QW_MEGA_CASE(2, 1) QW_MEGA_CASE(2, 2) QW_MEGA_CASE(2, 4)
  QW_MEGA_CASE(4, 1) QW_MEGA_CASE(4, 2) QW_MEGA_CASE(4, 4)
#undef QW_MEGA_CASE
}
