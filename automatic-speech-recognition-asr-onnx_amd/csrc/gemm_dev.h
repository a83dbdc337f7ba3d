// Device-side pieces shared by the tiled bf16 GEMM kernels (csrc/gemm.hip, csrc/gemm_pp.hip): epilogue flags, activation functions,
// the XCD-aware tile order, the column permutation that gives every lane consecutive output columns, and the register epilogue.
#pragma once
#include "gemm.h"

namespace {

enum { E_ADD = 1, E_ADD2 = 2, E_F32 = 4, E_LO = 8, E_AMAX = 32, E_BIAS = 64, E_LN = 128, E_ST = 256 };

template <int ACT>
__device__ __forceinline__ float apply_act_ct(float v) {
  if constexpr (ACT == ACT_RELU) return fmaxf(v, 0.0f);
  else if constexpr (ACT == ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  else if constexpr (ACT == ACT_GELU_TANH) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.0f + tanhf(u));
  } else return v;
}

// erf-GELU for the bf16 / e4m3 kernels' epilogues. With a = |v|:  gelu(v) = max(v, 0) - a * Phi(-a),  and  Phi(-a) = erfc(a / sqrt 2) / 2 = 2^q(a)  with q a
// degree-6 polynomial fitted to log2 Phi(-a) on [0, 13] (weighted for the absolute error of the product; tools/probes/gelu_fit.py): |error| <= 6e-8
// absolute -- below the f32 rounding of the reference's own erf -- and <= 4e-4 relative in the negative tail, a tenth of a bf16 ulp. Cost per value: one
// v_exp_f32 and, on pairs of values, six v_pk_fma_f32 plus min / max / fma -- about half of the Abramowitz-Stegun 7.1.26 form used before (one v_rcp_f32 more and
// no packed arithmetic), which itself replaced the library erff's two-branch polynomial (8 us of a 256 x 256 tile's epilogue). a is clamped at 12, where
// the term is below 1e-30 and before the polynomial's leading coefficient turns it around. Verification mode keeps erff.
typedef float f32pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf_fast2(float& v0, float& v1) {
  const f32pair_t a = {fminf(fabsf(v0), 12.0f), fminf(fabsf(v1), 12.0f)};
  f32pair_t q = a * 3.30953373e-05f + -0.000769241955f;
  q = q * a + 0.00808078996f;
  q = q * a + -0.0534122179f;
  q = q * a + -0.458770889f;
  q = q * a + -1.15120173f;
  q = q * a + -0.999993058f;
  v0 = fmaf(-a.x, __builtin_amdgcn_exp2f(q.x), fmaxf(v0, 0.0f));
  v1 = fmaf(-a.y, __builtin_amdgcn_exp2f(q.y), fmaxf(v1, 0.0f));
}
__device__ __forceinline__ float gelu_erf_fast(float v) {
  float w = v;
  gelu_erf_fast2(v, w);
  return v;
}

__device__ __forceinline__ float apply_act_rt(float v, int act) {
  switch (act) {
    case ACT_RELU: return apply_act_ct<ACT_RELU>(v);
    case ACT_GELU_ERF: return apply_act_ct<ACT_GELU_ERF>(v);
    case ACT_GELU_TANH: return apply_act_ct<ACT_GELU_TANH>(v);
    default: return v;
  }
}

// XCD-aware bijective remap (8 XCDs, block b runs on XCD b % 8): XCD x gets a contiguous tile range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Column owned by fragment j, fragment-row index fr (0..15) inside a wave's column block. Fragments are paired
// (2p, 2p+1) and the W rows they load are permuted so that the C fragment of the pair gives every lane EIGHT
// consecutive output columns (8 * (fr / 4) + 4 * (j & 1) + fr % 4 inside the 32-column pair): one 16-byte bf16
// store / two adjacent 16-byte f32 stores per lane, and the four lanes of a row cover 64 / 128 contiguous bytes.
__device__ __forceinline__ int frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }
// LDS slot-swizzle key of a W-tile row. A fragment's 16 permuted rows are {8a + b (+4)}: within one row parity (the two
// 128-byte halves of the 256-byte bank row) the 8 rows must get 8 different keys => key = 2 * ((r >> 3) & 3) + ((r >> 1) & 1).
__device__ __forceinline__ int w_swz(int r) { return (((r >> 3) & 3) << 1) | ((r >> 1) & 1); }

template <int EPI, int F> __device__ __forceinline__ bool epi_has(const void* p) {
  if constexpr (EPI < 0) return p != nullptr; else return (EPI & F) != 0;
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  uint2 w;
  w.x = pack_bf16x2(a, b);
  w.y = pack_bf16x2(c, d);
  *reinterpret_cast<uint2*>(p) = w;
}

// ---- swapped orientation: acc[i][j][r] = C[m_wave + 16 i + (lane & 15)][n_wave + frag_col(j, 4 (lane >> 4) + r)]
template <typename OutT> __device__ __forceinline__ void store8(OutT* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 w;
  w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]); w.z = pack_bf16x2(v[4], v[5]); w.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = w;
}

template <typename OutT, int ACT, int EPI, int NJ, int MI = 4>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& g, f32x4_t (&acc)[MI][NJ], int m_wave, int n_wave, int lane,
                                              const float2* ln_stats = nullptr,     // (mean, rstd) of row m_wave + i, in LDS
                                              const float* bias_at = nullptr) {     // non-null: the bias vector is read from here (e.g. a copy staged in LDS) instead of g.bias
  static_assert(NJ % 2 == 0, "fragments are paired");
  const int frow = lane & 15, fgrp = lane >> 4;
  const bool has_bias = epi_has<EPI, E_BIAS>(g.bias), has_add = epi_has<EPI, E_ADD>(g.add), has_add2 = epi_has<EPI, E_ADD2>(g.add2);
  const bool want_f32 = epi_has<EPI, E_F32>(g.out_f32), want_lo = epi_has<EPI, E_LO>(g.out_lo);
  if (epi_has<EPI, E_AMAX>(g.amax_val)) {
    const int n_slabs = g.N / (NJ * 16), slab = n_wave / (NJ * 16);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float best = -INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {            // (pair, half, r) ascending == ascending column inside the lane
        const int n0 = n_wave + frag_col(j, fgrp * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + (has_bias ? (bias_at ? bias_at : g.bias)[n0 + r] : 0.0f);
          if (n0 + r >= g.n_valid) v = -INFINITY;
          if (v > best) { best = v; bidx = n0 + r; }          // first max wins
        }
      }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
      }
      const int m = m_wave + i * 16 + frow;
      if (fgrp == 0 && m < g.M) {
        g.amax_val[(size_t)m * n_slabs + slab] = best;
        g.amax_idx[(size_t)m * n_slabs + slab] = bidx;
      }
    }
  }
  if (!(want_f32 || want_lo)) return;
#pragma unroll
  for (int p = 0; p < NJ / 2; ++p) {
    const int n = n_wave + p * 32 + fgrp * 8;              // this lane's 8 consecutive columns
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
      const float* bp = bias_at ? bias_at : g.bias;
      const float4 lo = *reinterpret_cast<const float4*>(bp + n), hi = *reinterpret_cast<const float4*>(bp + n + 4);
      b8[0] = lo.x; b8[1] = lo.y; b8[2] = lo.z; b8[3] = lo.w; b8[4] = hi.x; b8[5] = hi.y; b8[6] = hi.z; b8[7] = hi.w;
    }
    float c8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI >= 0 && (EPI & E_LN) != 0) {
      const float4 lo = *reinterpret_cast<const float4*>(g.ln_colsum + n), hi = *reinterpret_cast<const float4*>(g.ln_colsum + n + 4);
      c8[0] = lo.x; c8[1] = lo.y; c8[2] = lo.z; c8[3] = lo.w; c8[4] = hi.x; c8[5] = hi.y; c8[6] = hi.z; c8[7] = hi.w;
    }
    float4 t1[MI][2], t2[MI][2];
    if (has_add) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {      // rows up to the 128-row tile edge are readable (padded buffers)
        const float* q = g.add + (size_t)min(m_wave + i * 16 + frow, g.M - 1) * g.ld_add + n;
        t1[i][0] = *reinterpret_cast<const float4*>(q);
        t1[i][1] = *reinterpret_cast<const float4*>(q + 4);
      }
    }
    if (has_add2) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = min(m_wave + i * 16 + frow, g.M - 1);
        const float* q = g.add2 + (size_t)(g.add2_rows ? g.add2_rows[m] : m) * g.ld_add2 + n;
        t2[i][0] = *reinterpret_cast<const float4*>(q);
        t2[i][1] = *reinterpret_cast<const float4*>(q + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m_wave + i * 16 + frow;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r]; v[4 + r] = acc[i][2 * p + 1][r]; }
      if constexpr (EPI >= 0 && (EPI & E_LN) != 0) {
        const float2 mr = ln_stats[i * 16 + frow];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] - mr.x * c8[e]) * mr.y;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += b8[e];
      if (has_add) {
        v[0] += t1[i][0].x; v[1] += t1[i][0].y; v[2] += t1[i][0].z; v[3] += t1[i][0].w;
        v[4] += t1[i][1].x; v[5] += t1[i][1].y; v[6] += t1[i][1].z; v[7] += t1[i][1].w;
      }
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        if constexpr (ACT == ACT_GELU_ERF && sizeof(OutT) == 2) gelu_erf_fast2(v[e], v[e + 1]);
        else if constexpr (ACT >= 0) { v[e] = apply_act_ct<ACT>(v[e]); v[e + 1] = apply_act_ct<ACT>(v[e + 1]); }
        else { v[e] = apply_act_rt(v[e], g.act); v[e + 1] = apply_act_rt(v[e + 1], g.act); }
      }
      if (has_add2) {                                         // post-activation term
        v[0] += t2[i][0].x; v[1] += t2[i][0].y; v[2] += t2[i][0].z; v[3] += t2[i][0].w;
        v[4] += t2[i][1].x; v[5] += t2[i][1].y; v[6] += t2[i][1].z; v[7] += t2[i][1].w;
      }
      if constexpr (EPI >= 0 && (EPI & E_ST) != 0) {           // statistics of the bf16-rounded row segment (32 columns per wave pair)
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const uint32_t pk = pack_bf16x2(v[e], v[e + 1]);
          const float lo = __uint_as_float(pk << 16), hi = __uint_as_float(pk & 0xffff0000u);
          s1 += lo + hi;
          s2 = fmaf(lo, lo, fmaf(hi, hi, s2));
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (fgrp == 0 && m < g.M) g.st_out[(size_t)m * (g.N >> 5) + ((n_wave >> 5) + p)] = make_float2(s1, s2);
      }
      if ((ACT >= 0 ? ACT == ACT_SWIGLU : g.act == ACT_SWIGLU)) {      // interleaved (gate, up) columns -> N / 2 activations
        if (m < g.M && want_lo) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = v[2 * e] / (1.0f + __expf(-v[2 * e])) * v[2 * e + 1];
          store4<OutT>(reinterpret_cast<OutT*>(g.out_lo) + (size_t)m * g.ld_out_lo + (n >> 1), y[0], y[1], y[2], y[3]);
        }
        continue;
      }
      if (m < g.M) {
        if (want_f32) store8<float>(g.out_f32 + (size_t)m * g.ld_out_f32 + n, v);
        if (want_lo) {
          OutT* o = reinterpret_cast<OutT*>(g.out_lo);
          if (g.lo_group > 0) o += (size_t)(n / g.lo_group) * g.ld_out_lo + (size_t)m * g.lo_group + (n % g.lo_group);
          else o += (size_t)m * g.ld_out_lo + n;
          store8<OutT>(o, v);
        }
      }
    }
  }
}

// ---- bias (+ activation) -> operand-dtype store for the 16 x 16 swapped layout with FULL-LINE stores. In epilogue_rows a store instruction covers
// 16 rows x 64 B (the four lanes of a row hold 32 consecutive columns): sixteen half cache lines per wave-instruction (cdna_hip_programming.md T21). Here the two 32-column halves of a row meet in one instruction:
// lanes of fragment rows 0..7 and 8..15 swap one half through a DPP row rotate (row_ror:8 pairs lane r with lane r ^ 8 of its 16-lane row), so
// instruction A writes rows 0..7 and instruction B rows 8..15 of the fragment, each as eight whole 128-byte lines. Worth 2 % on Whisper fc1 (in-run A/B 309 -> 302 us):
// what a tile's epilogue really costs is the write burst itself -- every CU stores its 128 KB tile at the same moment, 33 MB at the ~6 TB/s the memory side takes
// (5 us per tile; the same launch without its stores: 262 us).
// acc[i][j] as in epilogue_rows (NJ = 4: 64 columns per wave); bias_at: the bias vector (global or an LDS copy); handles the lo_group slab layout.
__device__ __forceinline__ uint32_t dpp_ror8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }

template <int ACT, int MI>
__device__ __forceinline__ void epilogue_rows_lo_lines(const GemmArgs& g, f32x4_t (&acc)[MI][4], int m_wave, int n_wave, int lane, const float* bias_at) {
  const int frow = lane & 15, fgrp = lane >> 4;
  const bool low = frow < 8;
  float b8[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float* bp = bias_at + n_wave + p * 32 + fgrp * 8;
    const float4 lo = *reinterpret_cast<const float4*>(bp), hi = *reinterpret_cast<const float4*>(bp + 4);
    b8[p][0] = lo.x; b8[p][1] = lo.y; b8[p][2] = lo.z; b8[p][3] = lo.w; b8[p][4] = hi.x; b8[p][5] = hi.y; b8[p][6] = hi.z; b8[p][7] = hi.w;
  }
  bf16_t* const out = reinterpret_cast<bf16_t*>(g.out_lo);
  const int ncol = n_wave + (low ? 0 : 32) + fgrp * 8;            // both instructions: this lane's 8 columns
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    uint4 w[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r] + b8[p][r]; v[4 + r] = acc[i][2 * p + 1][r] + b8[p][4 + r]; }
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        if constexpr (ACT == ACT_GELU_ERF) gelu_erf_fast2(v[e], v[e + 1]); else { v[e] = apply_act_ct<ACT>(v[e]); v[e + 1] = apply_act_ct<ACT>(v[e + 1]); }
      }
      w[p].x = pack_bf16x2(v[0], v[1]); w[p].y = pack_bf16x2(v[2], v[3]); w[p].z = pack_bf16x2(v[4], v[5]); w[p].w = pack_bf16x2(v[6], v[7]);
    }
    const uint4 send = low ? w[1] : w[0];
    uint4 recv;
    recv.x = dpp_ror8(send.x); recv.y = dpp_ror8(send.y); recv.z = dpp_ror8(send.z); recv.w = dpp_ror8(send.w);
    const uint4 va = low ? w[0] : recv, vb = low ? recv : w[1];
    const int ma = m_wave + i * 16 + (frow & 7), mb = ma + 8;
    if (g.dbg & 8) { asm volatile("" ::"v"(va.x), "v"(va.y), "v"(va.z), "v"(va.w), "v"(vb.x), "v"(vb.y), "v"(vb.z), "v"(vb.w)); continue; }      // timing probe: everything but the stores
    if (g.lo_group > 0) {
      bf16_t* o = out + (size_t)(ncol / g.lo_group) * g.ld_out_lo + (ncol % g.lo_group);
      if (ma < g.M) *reinterpret_cast<uint4*>(o + (size_t)ma * g.lo_group) = va;
      if (mb < g.M) *reinterpret_cast<uint4*>(o + (size_t)mb * g.lo_group) = vb;
    } else {
      if (ma < g.M) *reinterpret_cast<uint4*>(out + (size_t)ma * g.ld_out_lo + ncol) = va;
      if (mb < g.M) *reinterpret_cast<uint4*>(out + (size_t)mb * g.ld_out_lo + ncol) = vb;
    }
  }
}

// ---- un-swapped orientation: acc[i][j][r] = C[m_wave + 16 i + 4 (lane >> 4) + r][n_wave + frag_col(j, lane & 15)]
template <typename OutT, int NJ>
__device__ __forceinline__ void epilogue_transposed(const GemmArgs& g, f32x4_t (&acc)[4][NJ], int m_wave, int n_wave, int lane) {
  const int frow = lane & 15, fgrp = lane >> 4;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n_wave + frag_col(j, frow);
    const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m0 = m_wave + i * 16 + fgrp * 4;
      OutT* o = reinterpret_cast<OutT*>(g.out_t) + (size_t)n * g.ld_out_t + m0;
      if (m0 + 3 < g.M) {
        store4<OutT>(o, acc[i][j][0] + b, acc[i][j][1] + b, acc[i][j][2] + b, acc[i][j][3] + b);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + r < g.M) Elem<OutT>::store(o + r, acc[i][j][r] + b);
      }
    }
  }
}



// ---- 32 x 32 accumulators (v_mfma_f32_32x32x16_bf16, swapped operands, W rows staged in fragment-position order so that
// position p holds column 16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3) of the 32-column block):
// acc[i][b][r] = C[m_wave + 32 i + (lane & 31)][n_wave + 32 b + 16 (lane >> 5) + r], r = 0..15 -- sixteen consecutive columns per lane.
// Epilogue terms: bias, add (pre-activation), activation (incl. SwiGLU pairs), add2 (post-activation, optional row table), f32 / operand-dtype
// stores (plain or lo_group slabs). No arg-max / LayerNorm / statistics forms.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <typename OutT, int ACT, int EPI, int MI, int NB>
__device__ __forceinline__ void epilogue_rows32(const GemmArgs& g, f32x16_t (&acc)[MI][NB], int m_wave, int n_wave, int lane) {
  const int mrow = lane & 31, half = lane >> 5;
  const bool has_bias = epi_has<EPI, E_BIAS>(g.bias), has_add = epi_has<EPI, E_ADD>(g.add), has_add2 = epi_has<EPI, E_ADD2>(g.add2);
  const bool want_f32 = epi_has<EPI, E_F32>(g.out_f32), want_lo = epi_has<EPI, E_LO>(g.out_lo);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n0 = n_wave + b * 32 + half * 16;             // this lane's 16 consecutive columns
    float4 bq[4];
    if (has_bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[e] = *reinterpret_cast<const float4*>(g.bias + n0 + 4 * e);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m_wave + i * 32 + mrow, mc = min(m, g.M - 1);
      float4 t1[4], t2[4];
      if (has_add) {
        const float* q = g.add + (size_t)mc * g.ld_add + n0;
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[e] = *reinterpret_cast<const float4*>(q + 4 * e);
      }
      if (has_add2) {
        const float* q = g.add2 + (size_t)(g.add2_rows ? g.add2_rows[mc] : mc) * g.ld_add2 + n0;
#pragma unroll
        for (int e = 0; e < 4; ++e) t2[e] = *reinterpret_cast<const float4*>(q + 4 * e);
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][b][r];
      if (has_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[4 * e] += bq[e].x; v[4 * e + 1] += bq[e].y; v[4 * e + 2] += bq[e].z; v[4 * e + 3] += bq[e].w; }
      }
      if (has_add) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[4 * e] += t1[e].x; v[4 * e + 1] += t1[e].y; v[4 * e + 2] += t1[e].z; v[4 * e + 3] += t1[e].w; }
      }
      if ((ACT >= 0 ? ACT == ACT_SWIGLU : g.act == ACT_SWIGLU)) {      // interleaved (gate, up) columns -> N / 2 activations
        if (m < g.M && want_lo) {
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = v[2 * e] / (1.0f + __expf(-v[2 * e])) * v[2 * e + 1];
          store8<OutT>(reinterpret_cast<OutT*>(g.out_lo) + (size_t)m * g.ld_out_lo + (n0 >> 1), y);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (ACT == ACT_GELU_ERF && sizeof(OutT) == 2) v[r] = gelu_erf_fast(v[r]);
        else if constexpr (ACT >= 0) v[r] = apply_act_ct<ACT>(v[r]); else v[r] = apply_act_rt(v[r], g.act);
      }
      if (has_add2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[4 * e] += t2[e].x; v[4 * e + 1] += t2[e].y; v[4 * e + 2] += t2[e].z; v[4 * e + 3] += t2[e].w; }
      }
      if (m < g.M) {
        float lo8[8], hi8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { lo8[r] = v[r]; hi8[r] = v[8 + r]; }
        if (want_f32) {
          float* o = g.out_f32 + (size_t)m * g.ld_out_f32 + n0;
          store8<float>(o, lo8); store8<float>(o + 8, hi8);
        }
        if (want_lo) {
          OutT* o = reinterpret_cast<OutT*>(g.out_lo);
          if (g.lo_group > 0) o += (size_t)(n0 / g.lo_group) * g.ld_out_lo + (size_t)m * g.lo_group + (n0 % g.lo_group);   // lo_group % 16 == 0
          else o += (size_t)m * g.ld_out_lo + n0;
          store8<OutT>(o, lo8); store8<OutT>(o + 8, hi8);
        }
      }
    }
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
