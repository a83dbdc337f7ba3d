// Decode-step GEMM: out[M <= 64][N] = A[M][K] W[N][K]^T for the autoregressive decoders (Whisper/Export_Whisper.py:614-667, one row per
// sequence), where every launch streams a few MB of cold weights from HBM once and the activation rows come from L2.
//
// What bounds these launches (tools/probes/decode_gemm_chain.py, cold weights, M = 32): not HBM (3-13 MB at 5 TB/s is 0.7-2.6 us) but
// what ONE CU can pull -- ~40 GB/s of weights + activation rows together -- times how unevenly the workgroups cover the chip:
//   * a workgroup owns 16 NT output columns x all rows and must read every activation row once per column granule, so the rows move
//     (N / 16 NT) times through L2 -> CU paths: 16-column granules move 2 bytes of activations per byte of weights at 32 rows;
//   * N = 1280 gives 80 granules = 80 busy CUs (fc2: 164 KB of weights + 328 KB of rows each = 13 us), N = 5120 gives 320 = a second,
//     quarter-full round (fc1: 11.7 us against 7.1 us for the 240 granules of q|k|v).
// So the launcher shapes the grid to <= 256 EVEN workgroups: NT = 2 (32-column granules: half the activation traffic) where 16-column
// granules would overflow the chip, and K split across workgroups where the granules alone leave CUs idle; the partial tiles are handed
// over with 16-byte write-through stores + one ticket and summed in split order by the last arriver (bit-reproducible).
//
// LayerNorm folded in (FOLD): the pre-LN decoder applies an affine-free LayerNorm before q|k|v, cross-q and fc1 (the affine is folded
// into the weights at arena build, Export_Whisper.py:215-225). A then holds the RAW residual rows in bf16 (second output of the
// producing GEMM) and  LN(x) W^T = rstd (x W^T - mean c),  c[n] = sum_k W[n][k];  sum(x) and sum(x^2) of every row come from two extra
// MFMAs per activation fragment (ones x A^T, A A^T) on the otherwise idle matrix pipe. This removes the per-workgroup LayerNorm prologue
// of the weight-streaming kernel (every workgroup re-normalising all rows: +4.7 us per launch) and the stand-alone LayerNorm launch.
//
// FP8 weights (W8, precision mode ASR_PRECISION_FP8W): W holds OCP e4m3 bytes [N][K] with one power-of-two f32 scale per output
// column; a lane loads 8 bytes per fragment, widens them to bf16 in registers (exact: 4 significant bits) and the epilogue multiplies the
// finished sum by the scale. With power-of-two scales  (sum_k a w8) * s  ==  sum_k a (w8 * s)  bit for bit, so the same kernel over the
// dequantised bf16 copy of the weights is an exact reference for this path (tests/test_whisper_fp8_gpu.py).
#include <type_traits>
#include "gemm.h"

namespace {

constexpr int DW = 8;                           // waves per workgroup; each takes one K sub-slice
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// 8 e4m3 bytes -> one bf16x8 MFMA fragment (v_cvt_scalef32_pk_bf16_fp8: two bytes -> two bf16 per instruction, scale 1)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16x8(u32x2_t r) {
  union { bf16x8_t v; bf16x2_hw_t p[4]; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, false); u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, true);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, false); u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, true);
  return u.v;
}

// 8 e2m1 nibbles (one dword) x the block's e8m0 scale -> one bf16x8 fragment (v_cvt_scalef32_pk_bf16_fp4: a byte = two elements, low nibble first; exact)
__device__ __forceinline__ bf16x8_t fp4x8_to_bf16x8(unsigned r, float scale) {
  union { bf16x8_t v; bf16x2_hw_t p[4]; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(r, scale, 0); u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(r, scale, 1);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(r, scale, 2); u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(r, scale, 3);
  return u.v;
}

// WQ: 0 = bf16 weights, 1 = e4m3 bytes + one scale per output column (W8), 2 = MXFP4: e2m1 nibbles + one e8m0 scale per (column, 32 k) -- the K-step of the MFMA is
// the block of the format, so a fragment needs the one scale byte of (its column, this step), applied by the widening instruction itself
template <int MT, int NT, bool FOLD, int WQ>
__global__ __launch_bounds__(512) void decode_gemm_kernel(const DecGemmArgs g_in) {
  constexpr bool W8 = WQ == 1, W4 = WQ == 2;       // (grids are shaped to <= one workgroup per CU: the activation batches may take the registers of two)
  // row blocks (gridDim.z > 1, no K split): block z multiplies rows 16 MT z .. of the batch -- a narrow output at 64 rows then spreads over twice the CUs, each pulling
  // half the activation rows through its fill path (decode_gemm_plan)
  DecGemmArgs g = g_in;
  if (gridDim.z > 1) {
    const int r0 = (int)blockIdx.z * MT * 16;
    g.A += (size_t)r0 * g.lda;
    if (g.add) g.add += (size_t)r0 * g.ld_add;
    if (g.out_f32) g.out_f32 += (size_t)r0 * g.ld_out_f32;
    if (g.out_lo) g.out_lo += (size_t)r0 * g.ld_out_lo;
    g.M = min(MT * 16, g.M - r0);
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const int n0 = blockIdx.x * 16 * NT;
  const int KS = gridDim.y, ks = blockIdx.y;
  // phase clock (probe only): workgroup (0, 0) -> slots 0..4, the last workgroup -> slots 5..9
  unsigned long long* clk = nullptr;
  if (g.dbg_clk && tid == 0) {
    if (blockIdx.x == 0 && blockIdx.y == 0) clk = g.dbg_clk;
    else if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) clk = g.dbg_clk + 5;
  }
  if (clk) clk[0] = wall_clock64();
  const int kw = (wave + (int)blockIdx.x) & (DW - 1);    // rotate the wave -> slice map per workgroup: the shared activation rows are not hit in lock-step
  // K in steps of 32: this workgroup's share of the steps, then this wave's share of those (shares differ by at most one step, so any split
  // count divides any K)
  const int steps = g.K >> 5;
  const int wg_lo = steps * ks / KS, wg_n = steps * (ks + 1) / KS - wg_lo;
  const int w_lo = wg_lo + wg_n * kw / DW, w_hi = wg_lo + wg_n * (kw + 1) / DW;
  const int k_begin = w_lo * 32, kslice = (w_hi - w_lo) * 32;
  const bf16_t* wp = g.W + (size_t)(n0 + frow) * g.ldw + k_begin + fgrp * 8;
  const unsigned char* wp8 = g.W8 + (size_t)(n0 + frow) * g.ldw + k_begin + fgrp * 8;     // (W8: ldw counts bytes = elements)
  const unsigned char* wp4 = g.W4 + (size_t)(n0 + frow) * (g.ldw >> 1) + ((k_begin + fgrp * 8) >> 1);       // (W4: ldw counts elements, two per byte)
  const unsigned char* sp4 = g.w_scale4 + (size_t)(n0 + frow) * (g.K >> 5) + (k_begin >> 5);
  const bf16_t* ap = g.A + (size_t)frow * g.lda + k_begin + fgrp * 8;
  constexpr int U = 8 / NT;                            // K-steps per trip: U x NT weight fragments in flight per wave (16 B per lane each; byte weights: 8 B)

  f32x4_t acc[MT][NT], sx[MT], sxx[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    sx[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; sxx[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // ---- operand requests (round 5, second form). The phase clock (tools/probes/decode_gemm_clock.py, profiles/r05_decode_gemm_clock.txt) puts 4.3 of a 6.7 us launch at
  // 32 rows (6.6 of 9.4 at 64, 3.2 of 5.3 at one row) between the first weight request and the last MFMA of a wave -- not the ~2.5 us the weights take to arrive, but that
  // plus one L2 round trip per activation fragment: written as "load a fragment, multiply" the loop compiled to load -> s_waitcnt vmcnt(0) -> MFMA, MT x K-steps times per wave.
  // All requests of a trip are therefore issued in front of its MFMAs, weights first, activation fragments in batches of <= 16, and NOTHING older is pending when the first
  // trip's requests go out (the epilogue's residual / bias / column-sum loads come after them: in the first form of this change they came first, their registers were
  // reused by the batch and the compiler parked the batch behind an s_waitcnt for them -- two memory round trips in sequence, slower than the one-at-a-time loop).
  using raw_t = typename std::conditional<W8, u32x2_t, typename std::conditional<W4, unsigned, bf16x8_t>::type>::type;
  unsigned char wsc[W4 ? U : 1][NT];                   // W4: the block scales of the trip's fragments
  constexpr int UA = MT * U > 16 ? (16 / MT < 1 ? 1 : 16 / MT) : U;        // K-steps per activation batch
  raw_t wf[U][NT];
  bf16x8_t af[UA][MT];
  auto issue_w = [&](int k) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k + u * 32 < kslice) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (W8) wf[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(wp8 + (size_t)j * 16 * g.ldw + k + u * 32));
          else if constexpr (W4) {
            wf[u][j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(wp4 + (size_t)j * 16 * (g.ldw >> 1) + ((k + u * 32) >> 1)));
            wsc[u][j] = sp4[(size_t)j * 16 * (g.K >> 5) + ((k + u * 32) >> 5)];
          }
          else wf[u][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)j * 16 * g.ldw + k + u * 32));
        }
      }
  };
  auto issue_a = [&](int k, int u0) {
#pragma unroll
    for (int uu = 0; uu < UA; ++uu)
      if (k + (u0 + uu) * 32 < kslice) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          // rows past M are never stored: their lanes request nothing (at one row the 16-row tile would pull 16 x the bytes through this CU's 64 B / clk path)
          af[uu][i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
          if (i * 16 + frow < g.M) af[uu][i] = *reinterpret_cast<const bf16x8_t*>(ap + (size_t)i * 16 * g.lda + k + (u0 + uu) * 32);
        }
      }
  };
  issue_w(0);
  issue_a(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // the residual term, the bias and the column sums of the fold do not depend on the product: requested here, they arrive under the weight stream
  constexpr int TILES = MT * NT;                       // output tiles of 16 x 16; wave t finishes tile t (TILES <= 8)
  static_assert(TILES <= DW, "one finishing wave per output tile");
  const int ti = wave / NT, tj = wave % NT;            // tile of this wave in the epilogue
  float4 addv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wave < TILES && g.add && ti * 16 + frow < g.M) addv = *reinterpret_cast<const float4*>(g.add + (size_t)(ti * 16 + frow) * g.ld_add + n0 + tj * 16 + fgrp * 4);
  float4 biasv = make_float4(0.f, 0.f, 0.f, 0.f), csumv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wave < TILES) {
    if (g.bias) biasv = *reinterpret_cast<const float4*>(g.bias + n0 + tj * 16 + fgrp * 4);
    if constexpr (FOLD) csumv = *reinterpret_cast<const float4*>(g.colsum + n0 + tj * 16 + fgrp * 4);
  }
  const bf16x8_t ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  if (clk) clk[1] = wall_clock64();
  for (int k = 0; k < kslice; k += 32 * U) {
    if (k > 0) { issue_w(k); issue_a(k, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += UA) {
      if (k + u0 * 32 >= kslice) break;                  // (wave-uniform)
      if (u0 > 0) { issue_a(k, u0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int uu = 0; uu < UA; ++uu) {
        const int u = u0 + uu;
        if (k + u * 32 < kslice) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              bf16x8_t wv;
              if constexpr (W8) wv = fp8x8_to_bf16x8(wf[u][j]);
              else if constexpr (W4) wv = fp4x8_to_bf16x8(wf[u][j], __uint_as_float((unsigned)wsc[u][j] << 23));
              else wv = wf[u][j];
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, af[uu][i], acc[i][j], 0, 0, 0);   // D[n = 4 fgrp + r][m = frow]
            }
            if constexpr (FOLD) {
              sx[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[uu][i], sx[i], 0, 0, 0);       // D[*][m = frow] = sum_k x[m][k]
              sxx[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[uu][i], af[uu][i], sxx[i], 0, 0, 0);       // D[a][b] = x[a] . x[b]: the diagonal is sum(x^2)
            }
          }
        }
      }
    }
  }
  if (clk) { asm volatile("s_nop 0" :: "v"(acc[0][0])); clk[2] = wall_clock64(); }       // (the product of this wave's slice is in its registers)
  // ---- cross-wave reduction through LDS: red[wave][tile][lane] (float4), statistics st[wave][row] (float2)
  float4* red = reinterpret_cast<float4*>(smem);
  float2* st = reinterpret_cast<float2*>(smem + DW * TILES * 1024);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) red[(wave * TILES + i * NT + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  if constexpr (FOLD) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (fgrp == (frow >> 2)) st[wave * MT * 16 + i * 16 + frow] = make_float2(sx[i][0], sxx[i][frow & 3]);
  }
  __syncthreads();
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int tile = wave;
  if (wave < TILES) {
    sum = red[tile * 64 + lane];
#pragma unroll
    for (int w = 1; w < DW; ++w) { const float4 q = red[(w * TILES + tile) * 64 + lane]; sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w; }
  }
  if (clk) { asm volatile("s_nop 0" :: "v"(sum.x)); clk[3] = wall_clock64(); }
  const int rows16 = MT * 16;
  float fold_s1 = 0.0f, fold_s2 = 0.0f;              // FOLD with K split across workgroups: the statistics summed over the splits
  if (KS > 1) {            // hand the partial tiles over (write-through, 16 bytes per lane); the last workgroup of this column granule finishes
    if (wave < TILES) {
      float* dst = g.ws + ((size_t)ks * rows16 + ti * 16 + frow) * g.N + n0 + tj * 16 + fgrp * 4;
      const f32x4_t v = {sum.x, sum.y, sum.z, sum.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
    float2* stat_ws = reinterpret_cast<float2*>(g.ws + (size_t)KS * rows16 * g.N) + ((size_t)blockIdx.x * KS) * rows16;   // [granule][split][row]
    if constexpr (FOLD) {                             // this workgroup's (sum x, sum x^2) over its K share, waves added in a fixed order
      if (wave < MT && fgrp == 0) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < DW; ++w) { const float2 q = st[w * MT * 16 + wave * 16 + frow]; s1 += q.x; s2 += q.y; }
        const u32x2_t v = {__float_as_uint(s1), __float_as_uint(s2)};
        float2* dst = stat_ws + (size_t)ks * rows16 + wave * 16 + frow;
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // (also: every wave is done with `red`, whose first word now carries the verdict)
    int* last = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(g.cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *last = ticket == KS - 1;
      if (ticket == KS - 1) __hip_atomic_store(g.cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*last) return;
    if (wave < TILES) {
      sum = make_float4(0.f, 0.f, 0.f, 0.f);
      // fixed order => bit-reproducible; sc1 loads bypass this CU's L1. Up to eight splits' tiles (and statistics) are requested together and waited for ONCE
      // (the wait names the registers, so no add can move above it): one memory-side round trip per eight splits instead of one per split.
      u32x2_t sq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sq[j] = u32x2_t{0u, 0u};
      float s1 = 0.0f, s2f = 0.0f;
      for (int s0 = 0; s0 < KS; s0 += 8) {
        f32x4_t q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int sj = min(s0 + j, KS - 1);
          const float* src = g.ws + ((size_t)sj * rows16 + ti * 16 + frow) * g.N + n0 + tj * 16 + fgrp * 4;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(q[j]) : "v"(src) : "memory");
          if constexpr (FOLD) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(sq[j]) : "v"(stat_ws + (size_t)sj * rows16 + ti * 16 + frow) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]),
                                            "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]), "+v"(sq[4]), "+v"(sq[5]), "+v"(sq[6]), "+v"(sq[7]) :: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (s0 + j < KS) {
            sum.x += q[j][0]; sum.y += q[j][1]; sum.z += q[j][2]; sum.w += q[j][3];
            if constexpr (FOLD) { s1 += __uint_as_float(sq[j][0]); s2f += __uint_as_float(sq[j][1]); }
          }
      }
      if constexpr (FOLD) { fold_s1 = s1; fold_s2 = s2f; }     // the row's statistics over all splits, in split order, where the epilogue looks for them
    }
  }
  if (wave >= TILES) return;
  const int m = ti * 16 + frow, n = n0 + tj * 16 + fgrp * 4;
  if (m >= g.M) return;
  if constexpr (W8) {      // (an opaque v_mul_f32: the exact power-of-two product must not be contracted into the fused multiply-adds below, or the rounding order would differ from the bf16 kernel's)
    const float4 sc = *reinterpret_cast<const float4*>(g.w_scale + n);
    auto mul = [](float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    sum.x = mul(sum.x, sc.x); sum.y = mul(sum.y, sc.y); sum.z = mul(sum.z, sc.z); sum.w = mul(sum.w, sc.w);
  }
  if constexpr (FOLD) {                                  // rstd (x W^T - mean c): statistics summed over the waves in a fixed order
    float s1 = 0.0f, s2 = 0.0f;
    if (KS > 1) { s1 = fold_s1; s2 = fold_s2; }
    else {
#pragma unroll
      for (int w = 0; w < DW; ++w) { const float2 q = st[w * MT * 16 + m]; s1 += q.x; s2 += q.y; }
    }
    const float inv_k = 1.0f / (float)g.K;
    const float mean = s1 * inv_k;
    const float rstd = rsqrtf(fmaxf(s2 * inv_k - mean * mean, 0.0f) + g.ln_eps);
    const float4 c = csumv;
    sum.x = (sum.x - mean * c.x) * rstd; sum.y = (sum.y - mean * c.y) * rstd; sum.z = (sum.z - mean * c.z) * rstd; sum.w = (sum.w - mean * c.w) * rstd;
  }
  sum.x += biasv.x; sum.y += biasv.y; sum.z += biasv.z; sum.w += biasv.w;
  sum.x += addv.x; sum.y += addv.y; sum.z += addv.z; sum.w += addv.w;
  if (g.act == ACT_GELU_ERF) {
    sum.x = 0.5f * sum.x * (1.0f + erff(sum.x * 0.70710678118654752440f)); sum.y = 0.5f * sum.y * (1.0f + erff(sum.y * 0.70710678118654752440f));
    sum.z = 0.5f * sum.z * (1.0f + erff(sum.z * 0.70710678118654752440f)); sum.w = 0.5f * sum.w * (1.0f + erff(sum.w * 0.70710678118654752440f));
  } else if (g.act == ACT_GELU_TANH) {
    auto gt = [](float v) { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); return 0.5f * v * (1.0f + tanhf(u)); };
    sum.x = gt(sum.x); sum.y = gt(sum.y); sum.z = gt(sum.z); sum.w = gt(sum.w);
  } else if (g.act == ACT_RELU) {
    sum.x = fmaxf(sum.x, 0.f); sum.y = fmaxf(sum.y, 0.f); sum.z = fmaxf(sum.z, 0.f); sum.w = fmaxf(sum.w, 0.f);
  }
  if (g.out_f32) *reinterpret_cast<float4*>(g.out_f32 + (size_t)m * g.ld_out_f32 + n) = sum;
  if (g.out_lo) {
    uint2 w;
    w.x = pack_bf16x2(sum.x, sum.y); w.y = pack_bf16x2(sum.z, sum.w);
    *reinterpret_cast<uint2*>(g.out_lo + (size_t)m * g.ld_out_lo + n) = w;
  }
  if (clk) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); clk[4] = wall_clock64(); }       // (this thread's output stores are acknowledged)
}

// c[n] = sum_k W[n][k] of the bf16 weights, accumulated in double (one wave per row)
__global__ void colsum_kernel(const bf16_t* __restrict__ W, int ldw, int N, int K, float* __restrict__ c) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  double s = 0.0;
  for (int k = lane; k < K; k += 64) s += (double)bf16_to_f32(W[(size_t)n * ldw + k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) c[n] = (float)s;
}

// Weight prefetch for the NEXT decode GEMM of the chain (launched on a side branch of the decode graph while the current GEMM runs): same grid as the GEMM it serves,
// workgroup (x, y) touches one dword of every 128-byte line of exactly the weight bytes that GEMM workgroup (x, y) will stream -- so the lines wait in the L2 of the XCD
// that workgroup lands on (dispatch is round-robin over XCDs by workgroup id for both kernels; if it were not, the prefetch would merely be useless). No output.
__global__ __launch_bounds__(256) void decode_gemm_prefetch_kernel(const unsigned char* __restrict__ W, int row_bytes, int elem, int rows_per_wg, int steps, unsigned* sink) {
  const int KS = gridDim.y, ks = blockIdx.y;
  const int lo = steps * ks / KS, n = steps * (ks + 1) / KS - lo;          // this workgroup's K-steps of 32 elements (the GEMM kernel's split rule)
  const int seg0 = lo * 32 * elem, seg_bytes = n * 32 * elem;
  const int lines = (seg_bytes + 127) >> 7;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < rows_per_wg * lines; i += 256) {
    const int r = i / lines, l = i - r * lines;
    acc += *reinterpret_cast<const unsigned*>(W + (size_t)(blockIdx.x * rows_per_wg + r) * row_bytes + seg0 + (l << 7));
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;                              // keeps the loads alive
}

template <int MT, int NT>
void launch_inst(const DecGemmArgs& g, int splits, hipStream_t s, int row_blocks = 1) {
  const size_t lds = (size_t)DW * MT * NT * 1024 + (size_t)DW * MT * 16 * 8;
  const dim3 grid(g.N / (16 * NT), splits, row_blocks);
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) {
      static PerDeviceOnce once;        // (per instantiation of this lambda: one kernel each)
      if (once.first()) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * DW), lds, s, g);
  };
  if (g.W4) { if (g.colsum) go(decode_gemm_kernel<MT, NT, true, 2>); else go(decode_gemm_kernel<MT, NT, false, 2>); }
  else if (g.W8) { if (g.colsum) go(decode_gemm_kernel<MT, NT, true, 1>); else go(decode_gemm_kernel<MT, NT, false, 1>); }
  else { if (g.colsum) go(decode_gemm_kernel<MT, NT, true, 0>); else go(decode_gemm_kernel<MT, NT, false, 0>); }
  HIP_CHECK(hipGetLastError());
}

}  // namespace

bool decode_gemm_supported(const DecGemmArgs& g) {
  return g.M >= 1 && g.M <= 64 && g.N % 32 == 0 && g.K % 32 == 0 && g.K >= 32 * DW && (g.lda * 2) % 16 == 0 && (g.W4 ? g.ldw % 32 == 0 : (g.ldw * (g.W8 ? 1 : 2)) % 16 == 0) &&
         !(g.colsum && !g.A) && !(g.W8 && !g.w_scale) && !(g.W4 && !g.w_scale4) && !(g.W4 && g.W8);
}

// grid shape: column granules of 16 NT, K split across `splits` workgroups -- at most one even round of the chip's CUs.
// A workgroup pulls (16 NT weight rows + the activation rows) x its K share through one CU's L2 path (the rows come once per granule), and a
// split costs a hand-over (partial tiles written through, a ticket, the last arriver re-reading `splits` tiles): the plan minimises
//   bytes per workgroup / RATE + (splits > 1 ? HAND + splits * PER_SPLIT : 0)
// over NT in {1, 2} and the split counts that keep granules x splits within one round; constants from tools/probes/decode_split_sweep.sh.
void decode_gemm_plan(const DecGemmArgs& g, int* nt, int* splits, int* row_blocks) {
  const int cus = gemm_env_cus();
  const int force_nt = gemm_env_decode_nt(), force_ks = gemm_env_decode_ks();          // tuning hooks, re-read at session creation like every other switch
  const int pm = g.plan_M > g.M ? g.plan_M : g.M;
  const int ws_rows = pm <= 16 ? 16 : pm <= 32 ? 32 : 64;        // rows per split slab as the kernel addresses them: [ks][MT * 16][N]
  const int mt = ws_rows / 16;
  const bool can_split = g.ws && g.cnt;
  double best_cost = 1e30;
  int best_nt = 1, best_ks = 1;
  for (int NT = 1; NT <= 2; ++NT) {
    if (g.N % (16 * NT) != 0 || mt * NT > DW) continue;          // one finishing wave per 16 x 16 output tile
    if (force_nt && NT != force_nt) continue;
    const int granules = g.N / (16 * NT);
    for (int sp = 1; sp <= 16; ++sp) {
      if (force_ks && sp != force_ks) continue;
      if (sp > 1 && (!can_split || (g.K >> 5) < sp * 2)) break;
      if (granules * sp > cus && !(sp == 1 && NT == 2)) continue;                                      // more than a round: only as the last resort (NT = 2, no split)
      if (sp > 1 && (size_t)sp * ws_rows * g.N * 4 + (size_t)granules * sp * ws_rows * 8 > g.ws_bytes) continue;
      const double bytes = (double)(16 * NT + ws_rows) * g.K * 2 / sp;
      const double rounds = (granules * sp + cus - 1) / cus;
      // us: ~45 GB/s per CU; a hand-over costs ~2 us (4 with the statistics of a folded LayerNorm) + 0.25 us per split tile the last arriver
      // re-reads -- measured on the Whisper shapes at 32 / 64 rows: out-proj 6.7 (no split) vs 6.9 (3 splits) at 32 rows, 9.2 vs 8.0 at 64;
      // cross-q with the fold 9.4 (no split) vs 10.4 (32-column granules x 6 splits) at 64 rows
      double cost = rounds * bytes / 45e3 + (sp > 1 ? (g.colsum ? 4.0 : 2.0) + 0.25 * sp : 0.0);
      if (cost < best_cost) { best_cost = cost; best_nt = NT; best_ks = sp; }
    }
  }
  int best_rb = 1;
  if (mt >= 2 && gemm_env_decode_rb() != 1) {                       // 17 .. 64 rows: two blocks of 16 / 32 rows, no K split -- (16 NT + rows / 2) K bytes per workgroup and no hand-over
    const int half = ws_rows / 2;                                   // (measured, Whisper-large-v3 decode: 64 sequences 146.4 -> 139.6 ms per 64 x 8 s batch, profiles/r06_decode_gemm_row_blocks.txt)
    for (int NT = 1; NT <= 2; ++NT) {
      if (g.N % (16 * NT) != 0 || (force_nt && NT != force_nt) || (force_ks && force_ks != 1)) continue;
      const int granules = g.N / (16 * NT);
      if (granules * 2 > cus) continue;
      const double cost = (double)(16 * NT + half) * g.K * 2 / 45e3;
      if (cost < best_cost || gemm_env_decode_rb() == 2) { best_cost = cost; best_nt = NT; best_ks = 1; best_rb = 2; }
    }
  }
  *nt = best_nt; *splits = best_ks;
  if (row_blocks) *row_blocks = best_rb;
}

void launch_decode_gemm(const DecGemmArgs& g, hipStream_t s) {
  ASR_REQUIRE(decode_gemm_supported(g), "decode_gemm: unsupported shape (M = %d, N = %d, K = %d)", g.M, g.N, g.K);
  ASR_REQUIRE(g.A && (g.W || g.W8 || g.W4) && (g.out_f32 || g.out_lo), "decode_gemm: null operand");
  int nt = 1, splits = 1, rb = 1;
  decode_gemm_plan(g, &nt, &splits, &rb);
  ASR_REQUIRE(g.N % (16 * nt) == 0, "decode_gemm: N = %d", g.N);
  if (rb == 2 && g.M > 32) {                                        // two 32-row blocks side by side (splits == 1)
    if (nt == 2) launch_inst<2, 2>(g, 1, s, 2); else launch_inst<2, 1>(g, 1, s, 2);
    return;
  }
  if (rb == 2 && g.M > 16) {                                        // two 16-row blocks
    if (nt == 2) launch_inst<1, 2>(g, 1, s, 2); else launch_inst<1, 1>(g, 1, s, 2);
    return;
  }
  const int mt = g.M <= 16 ? 1 : g.M <= 32 ? 2 : 4;
  if (nt == 2) { if (mt == 1) launch_inst<1, 2>(g, splits, s); else if (mt == 2) launch_inst<2, 2>(g, splits, s); else launch_inst<4, 2>(g, splits, s); }
  else if (mt == 1) launch_inst<1, 1>(g, splits, s);
  else if (mt == 2) launch_inst<2, 1>(g, splits, s);
  else launch_inst<4, 1>(g, splits, s);
}

void launch_colsum_bf16(const bf16_t* W, int ldw, int N, int K, float* c, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, ldw, N, K, c);
  HIP_CHECK(hipGetLastError());
}

void launch_decode_gemm_prefetch(const DecGemmArgs& g, hipStream_t s) {
  int nt = 1, splits = 1;
  decode_gemm_plan(g, &nt, &splits, nullptr);
  const unsigned char* w = g.W8 ? g.W8 : reinterpret_cast<const unsigned char*>(g.W);
  const int elem = g.W8 ? 1 : 2;
  hipLaunchKernelGGL(decode_gemm_prefetch_kernel, dim3(g.N / (16 * nt), splits), dim3(256), 0, s, w, g.ldw * elem, elem, 16 * nt, g.K >> 5, (unsigned*)nullptr);
  HIP_CHECK(hipGetLastError());
}
