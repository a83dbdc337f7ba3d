// Non-GEMM kernels of the ASR hot path for gfx950 (wave64, MFMA, LDS-staged tiles).
#include "kernels.h"
#include "gemm.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int HOP = 160, WIN = 400;          // 10 ms / 25 ms at 16 kHz: every in-scope front-end
constexpr int FB_FRAMES = 64;                // frames per workgroup
constexpr int FB_SPAN = (FB_FRAMES - 1) * HOP + WIN;           // 10480 samples
constexpr int FB_AUDIO_LDS = FB_SPAN + FB_SPAN / HOP + 16;     // skewed: +1 float per hop => row stride 161
constexpr int FB_PLD = 273;                  // power row stride (floats): 272 bins + 1 pad

// ------------------------------------------------------------------------------------ fbank
// One workgroup = 64 frames of one utterance. Stage the 10480-sample span in LDS once (frames
// overlap 2.5x), then   spectrum = frames x foldedDFT^T   on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32),
// power = re^2 + im^2 lane-locally (re / im fragments share the C layout), power tile -> LDS,
// mel = power x melT on the same MFMA, then clamp + ln.  Frame m of the span starts at m*160, so the
// A fragment (row = frame, k = sample) is a strided LDS read; the +1-per-hop skew makes the 16 rows
// of a fragment hit 16 different banks.
__global__ __launch_bounds__(256) void fbank_kernel(const FbankArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* aud = reinterpret_cast<float*>(smem);
  float* pw = aud + FB_AUDIO_LDS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = a.blk_utt[blockIdx.x], f0 = a.blk_f0[blockIdx.x];
  const UttPlan up = a.plan[u];
  const float* src = a.audio + up.audio_off;
  const int s0 = f0 * HOP;
  // every thread requests all of its samples before it stores the first (a load / store loop runs as one memory round trip per iteration: see fbank_split_kernel)
  constexpr int NL = (FB_SPAN + 255) / 256;
  float av[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    // Kaldi reads x[s]; Whisper the padded signal [x[200..1] | x | x[L-2 .. L-41]]  (reflect, right pad shortened by one hop)
    const int i = tid + j * 256, L = up.n_samples, half = WIN / 2, p = s0 + i;
    int idx = p;
    bool ok = i < FB_SPAN && p < L;
    if (a.whisper) {
      idx = p < half ? half - p : p < half + L ? p - half : 2 * L + half - 2 - p;
      ok = i < FB_SPAN && p < L + WIN - HOP;
    }
    av[j] = ok ? src[idx] : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int i = tid + j * 256;
    if (i < FB_SPAN) aud[i + i / HOP] = av[j];
  }
  __syncthreads();

  const int frow = lane & 15, fgrp = lane >> 4;
  const float4* dft = reinterpret_cast<const float4*>(a.dft_packed);
  for (int t = wave; t < a.n_bin_tiles; t += 4) {
    f32x4_t re[4], im[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { re[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; im[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const float4* dre = dft + (size_t)(t * 2 + 0) * a.n_kchunks * 64 + lane;
    const float4* dim = dft + (size_t)(t * 2 + 1) * a.n_kchunks * 64 + lane;
    float4 br = dre[0], bi = dim[0];
    for (int kc = 0; kc < a.n_kchunks; ++kc) {
      float4 brn = br, bin = bi;                 // the next chunk's basis columns are requested before this chunk's MFMAs
      if (kc + 1 < a.n_kchunks) { brn = dre[(kc + 1) * 64]; bin = dim[(kc + 1) * 64]; }
      const float brv[4] = {br.x, br.y, br.z, br.w};
      const float biv[4] = {bi.x, bi.y, bi.z, bi.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kc * 16 + j * 4 + fgrp;
        const int koff = k + k / HOP;
        float af[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = aud[(mt * 16 + frow) * (HOP + 1) + koff];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          re[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], brv[j], re[mt], 0, 0, 0);
          im[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], biv[j], im[mt], 0, 0, 0);
        }
      }
      br = brn; bi = bin;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pw[(mt * 16 + fgrp * 4 + r) * FB_PLD + t * 16 + frow] = re[mt][r] * re[mt][r] + im[mt][r] * im[mt][r];
  }
  __syncthreads();

  // mel: wave w owns frames [16w, 16w+16); all mel tiles
  float wmax = -INFINITY;
  const float4* melp = reinterpret_cast<const float4*>(a.mel_packed);
  for (int nt = 0; nt < a.n_mel_tiles; ++nt) {
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float4 b4 = melp[(size_t)(nt * a.n_bin_tiles) * 64 + lane];
    for (int kc = 0; kc < a.n_bin_tiles; ++kc) {
      float4 b4n = b4;
      if (kc + 1 < a.n_bin_tiles) b4n = melp[(size_t)(nt * a.n_bin_tiles + kc + 1) * 64 + lane];
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = pw[(wave * 16 + frow) * FB_PLD + kc * 16 + j * 4 + fgrp];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc, 0, 0, 0);
      }
      b4 = b4n;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + wave * 16 + fgrp * 4 + r;
      if (f < up.n_frames) {
        const float c = fmaxf(acc[r], a.log_floor);
        const float v = a.whisper ? log10f(c) : logf(c);
        a.mel_out[(size_t)(up.frame_off + f) * a.n_mels + nt * 16 + frow] = v;
        wmax = fmaxf(wmax, v);
      }
    }
  }
  if (a.whisper) {
    __shared__ float red[4];
    wmax = wave_max(wmax);
    if (lane == 0) red[wave] = wmax;
    __syncthreads();
    if (tid == 0) a.blk_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// ------------------------------------------------------------------------------------ fbank, bf16 sessions: split-operand DFT
// The DFT is 92 % of the front-end's flops and the exact-f32 MFMA runs at 1/16 of the bf16 rate. Here both operands are split into bf16
// terms and the product is summed on the bf16 pipe with f32 accumulation:  x = xh + xl  (two terms: exact for 16-bit PCM values, 2^-17
// relative otherwise -- below the quantisation noise of the audio itself),  b = bh + bm + bl  (three terms: the basis to 2^-24);
// x b ~= xh bh + xh bm + xh bl + xl bh + xl bm  -- five 16x16x32 MFMAs (80 cycles) per 32 samples against eight 16x16x4 f32 MFMAs (256).
// The dropped terms (xl bl, xl ... ) are below 2^-24 of |x||b|. Power, mel projection and log stay in exact f32 as before.
constexpr int FB_HP = HOP + 8;                                   // bf16 audio row pitch per hop: +8 elements => 16 rows of a fragment hit 16 different 16-byte slots
constexpr int FB_K32 = (WIN + 31) / 32;                          // 13 chunks of 32 samples (the last one half empty: basis rows >= 400 are zero)
constexpr int FB_A16 = ((FB_SPAN + 32 + HOP - 1) / HOP) * FB_HP; // bf16 elements per split array (span + the tail the last chunk reads)

// Eight waves (round 5; four before): the workgroup's 115 KB of LDS allow one workgroup per CU, so with four waves every SIMD held ONE wave and nothing covered its
// fragment reads, its waits or its accumulator hand-offs (ablations, profiles/r05_fbank_ablations.txt: DFT 42 us and mel 26 us per workgroup against 19 + 5 us of MFMA
// issue). Two waves per SIMD split the bin tiles (DFT) and the mel tiles (mel) of the same 64 frames; every accumulator still sums in the same order.
constexpr int FB_WAVES = 8;
__global__ __launch_bounds__(64 * FB_WAVES) void fbank_split_kernel(const FbankArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* audh = reinterpret_cast<bf16_t*>(smem);
  bf16_t* audl = audh + FB_A16;
  float* pw = reinterpret_cast<float*>(audl + FB_A16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = a.blk_utt[blockIdx.x], f0 = a.blk_f0[blockIdx.x];
  const UttPlan up = a.plan[u];
  const float* src = a.audio + up.audio_off;
  const int s0 = f0 * HOP;
  // the workgroup's audio span: every thread requests ALL of its samples before it uses the first (written as one loop of load / split / store the compiler kept the
  // loads in program order behind a wait each -- 42 dependent memory round trips, ~60 of the workgroup's ~78 us, with nothing else on the CU to cover them)
  constexpr int FB_NL = (FB_SPAN + 32 + 64 * FB_WAVES - 1) / (64 * FB_WAVES);
  float av[FB_NL];
#pragma unroll
  for (int j = 0; j < FB_NL; ++j) {
    const int i = tid + j * 64 * FB_WAVES;
    // one predicated load per sample: Kaldi reads x[s]; Whisper the padded signal [x[200..1] | x | x[L-2 .. L-41]]  (reflect, right pad shortened by one hop)
    const int L = up.n_samples, half = WIN / 2, p = s0 + i;
    int idx = p;
    bool ok = i < FB_SPAN + 32 && p < L;
    if (a.whisper) {
      idx = p < half ? half - p : p < half + L ? p - half : 2 * L + half - 2 - p;
      ok = i < FB_SPAN + 32 && p < L + WIN - HOP;
    }
    av[j] = (ok && !(a.dbg & 4)) ? src[idx] : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < FB_NL; ++j) {
    const int i = tid + j * 64 * FB_WAVES;
    if (i < FB_SPAN + 32) {
      const float v = av[j];
      const uint32_t hb = pack_bf16x2(v, 0.0f) & 0xffffu;
      const float hi = __uint_as_float(hb << 16);
      const int pos = i + (i / HOP) * 8;
      audh[pos] = (bf16_t)hb;
      audl[pos] = (bf16_t)(pack_bf16x2(v - hi, 0.0f) & 0xffffu);
    }
  }
  __syncthreads();

  const int frow = lane & 15, fgrp = lane >> 4;
  const uint4* tab = reinterpret_cast<const uint4*>(a.dft_split);
  // A wave's (bin tile, chunk) steps form ONE stream of basis fragments (6 KB per step: re | im x three bf16 terms) with FB_PF steps in flight, across tile
  // boundaries. With one step in flight (rounds 2-4) the loop ran at one L2 round trip per step -- 13 x 5 trips per workgroup next to 17 us of MFMA work, and the
  // workgroup's 115 KB of LDS leave no second workgroup on the CU to cover them: 286 us per 64 x 8 s batch. The flat step loop is unrolled by FB_PF so that the ring slots
  // are static registers that are never moved while their load is in flight; the scheduling barriers keep the refills where they are written (left alone, the
  // scheduler sinks every refill to just in front of its use -- one step in flight again).
  constexpr int FB_PF = 4;
  union BF { uint4 q; bf16x8_t v; };
  BF ring[FB_PF][6];
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int n_steps = (wv < a.n_bin_tiles && !(a.dbg & 1)) ? (a.n_bin_tiles - wv + FB_WAVES - 1) / FB_WAVES * FB_K32 : 0;
  constexpr size_t FB_IM = (size_t)FB_K32 * 3 * 64;                       // im fragments of a tile follow its re fragments
  auto frag_ptr = [&](int t, int kc) { return tab + (((size_t)(t * 2) * FB_K32 + kc) * 3) * 64 + lane; };
  int pt = wv, pkc = 0, ps = 0;                                           // the step the next refill fetches (it stops at the wave's last step: harmless re-reads)
  auto refill = [&](BF (&slot)[6]) {
    const uint4* src = frag_ptr(pt, pkc);
#pragma unroll
    for (int z = 0; z < 3; ++z) { slot[z].q = src[z * 64]; slot[3 + z].q = src[FB_IM + z * 64]; }
    if (ps + 1 < n_steps) { ++ps; if (++pkc == FB_K32) { pkc = 0; pt += FB_WAVES; } }
  };
  f32x4_t re[4], im[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { re[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; im[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  if (n_steps > 0) {
#pragma unroll
    for (int u = 0; u < FB_PF; ++u) refill(ring[u]);
  }
  int t = wv, kc = 0;
  for (int s0 = 0; s0 < n_steps; s0 += FB_PF) {
#pragma unroll
    for (int u = 0; u < FB_PF; ++u) {
      if (s0 + u < n_steps) {                                               // (wave-uniform)
        const int k = kc * 32 + fgrp * 8;
        const int koff = k + ((k >= HOP) + (k >= 2 * HOP)) * 8;
        bf16x8_t ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int pos = (mt * 16 + frow) * FB_HP + koff;
          ah[mt] = *reinterpret_cast<const bf16x8_t*>(audh + pos); al[mt] = *reinterpret_cast<const bf16x8_t*>(audl + pos);
        }
        // Per accumulator the five terms arrive smallest first (the f32 accumulator then rounds the dominant product last), as before; across accumulators the
        // order is term-major, so that two MFMAs on the SAME accumulator are eight issues apart (mt-major they were two apart: a dependent-accumulator stall per MFMA)
#define ASR_FB_TERM(X, ZR, ZI)                                                                          \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                               \
          re[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[mt], ring[u][ZR].v, re[mt], 0, 0, 0);       \
          im[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[mt], ring[u][ZI].v, im[mt], 0, 0, 0);       \
        }
        ASR_FB_TERM(al, 1, 4)
        ASR_FB_TERM(ah, 2, 5)
        ASR_FB_TERM(al, 0, 3)
        ASR_FB_TERM(ah, 1, 4)
        ASR_FB_TERM(ah, 0, 3)
#undef ASR_FB_TERM
        if (++kc == FB_K32) {                                               // the tile is complete: power spectrum to LDS, next tile
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              pw[(mt * 16 + fgrp * 4 + r) * FB_PLD + t * 16 + frow] = re[mt][r] * re[mt][r] + im[mt][r] * im[mt][r];
            re[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; im[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          }
          kc = 0; t += FB_WAVES;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(a.dbg & 8)) refill(ring[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();

  // mel: wave w owns frames [16w, 16w+16) (exact f32, as in fbank_kernel). The mel tiles advance together: one power fragment read feeds every tile's MFMA, and
  // consecutive MFMAs hit different accumulators (tile by tile, every MFMA waited for its predecessor on the same accumulator -- 40 cycles -- and re-read the fragment);
  // every accumulator still sums its bins in the same order
  float wmax = -INFINITY;
  const float4* melp = reinterpret_cast<const float4*>(a.mel_packed);
  constexpr int FB_MT = 4;                               // mel tiles in flight per wave (80 mels = 5 tiles, 128 mels = 8: split over the two waves of a frame tile)
  const int fw = wave & 3, mg = wave >> 2;               // frames [16 fw, 16 fw + 16), mel tiles mg, mg + 2, ...
  for (int nt0 = mg; nt0 < ((a.dbg & 2) ? 0 : a.n_mel_tiles); nt0 += 2 * FB_MT) {
    f32x4_t acc[FB_MT];
#pragma unroll
    for (int q = 0; q < FB_MT; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mel_frags = [&](float4 (&b)[FB_MT], int kc) {
#pragma unroll
      for (int q = 0; q < FB_MT; ++q) b[q] = nt0 + 2 * q < a.n_mel_tiles ? melp[(size_t)((nt0 + 2 * q) * a.n_bin_tiles + kc) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 b4[FB_MT], b4n[FB_MT];
    mel_frags(b4, 0);
    for (int kc = 0; kc < a.n_bin_tiles; ++kc) {
      mel_frags(b4n, kc + 1 < a.n_bin_tiles ? kc + 1 : kc);         // the next bin tile's weights are requested before this one's MFMAs
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = pw[(fw * 16 + frow) * FB_PLD + kc * 16 + j * 4 + fgrp];
#pragma unroll
        for (int q = 0; q < FB_MT; ++q) {
          const float bv = j == 0 ? b4[q].x : j == 1 ? b4[q].y : j == 2 ? b4[q].z : b4[q].w;
          if (nt0 + 2 * q < a.n_mel_tiles) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < FB_MT; ++q) b4[q] = b4n[q];
    }
#pragma unroll
    for (int q = 0; q < FB_MT; ++q) {
      if (nt0 + 2 * q >= a.n_mel_tiles) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + fw * 16 + fgrp * 4 + r;
        if (f < up.n_frames) {
          const float c = fmaxf(acc[q][r], a.log_floor);
          const float v = a.whisper ? log10f(c) : logf(c);
          a.mel_out[(size_t)(up.frame_off + f) * a.n_mels + (nt0 + 2 * q) * 16 + frow] = v;
          wmax = fmaxf(wmax, v);
        }
      }
    }
  }
  if (a.whisper) {
    __shared__ float red[FB_WAVES];
    wmax = wave_max(wmax);
    if (lane == 0) red[wave] = wmax;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
#pragma unroll
      for (int w = 1; w < FB_WAVES; ++w) m = fmaxf(m, red[w]);
      a.blk_max[blockIdx.x] = m;
    }
  }
}

// one thread per (fragment, lane): gathers the 8 basis values B[k][n] of its slot from the f32 fragment table and writes the three bf16 terms
__global__ void fbank_split_table_kernel(const float* __restrict__ dft_packed, int n_bin_tiles, int n_kchunks16, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = n_bin_tiles * 2 * FB_K32 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, kc = (idx >> 6) % FB_K32, tr = (idx >> 6) / FB_K32;     // tr = bin tile * 2 + (re | im)
  const int frow = lane & 15, fgrp = lane >> 4;
  uint32_t w[3][4] = {};
  for (int e = 0; e < 8; ++e) {
    const int k = kc * 32 + fgrp * 8 + e;
    float v = 0.0f;
    if (k < n_kchunks16 * 16) {
      // packed[((tr) * n_kchunks16 + k / 16) * 64 + ((k % 4) * 16 + frow)].component[(k % 16) / 4]  =  B[k][16 t + frow]
      v = dft_packed[(((size_t)tr * n_kchunks16 + (k >> 4)) * 64 + ((k & 3) * 16 + frow)) * 4 + ((k & 15) >> 2)];
    }
    float r = v;
    for (int z = 0; z < 3; ++z) {
      const uint32_t b = pack_bf16x2(r, 0.0f) & 0xffffu;
      r -= __uint_as_float(b << 16);
      w[z][e >> 1] |= b << ((e & 1) * 16);
    }
  }
  for (int z = 0; z < 3; ++z) out[((size_t)(tr * FB_K32 + kc) * 3 + z) * 64 + lane] = make_uint4(w[z][0], w[z][1], w[z][2], w[z][3]);
}

// ------------------------------------------------------------------------------------ LFR + CMVN
__global__ void lfr_cmvn_kernel(const LfrArgs a) {
  const int m = blockIdx.x;
  float* o = a.out + (size_t)m * a.ld_out;
  const int u = a.row_utt[m];
  bf16_t* olo = a.out_lo ? a.out_lo + (size_t)m * a.ld_out : nullptr;
  if (u < 0) {
    for (int c = threadIdx.x; c < a.ld_out; c += blockDim.x) { o[c] = 0.0f; if (olo) olo[c] = 0; }
    return;
  }
  const UttPlan up = a.plan[u];
  const int t = m - up.row_off;
  const int left = (a.lfr_m - 1) / 2;
  for (int c = threadIdx.x; c < a.ld_out; c += blockDim.x) {
    float v = 0.0f;
    if (c < a.feat && t < up.T) {
      if (a.n_prompt > 0 && t == 0) {
        v = a.language_embed[(size_t)up.lang * a.feat + c];
      } else if (t < a.n_prompt) {
        v = a.system_embed[(size_t)(t - 1) * a.feat + c];
      } else {
        const int j = t - a.n_prompt;
        int f = j * a.lfr_n + c / a.n_mels - left;
        f = min(max(f, 0), up.n_frames - 1);
        const float x = a.mel[(size_t)(up.frame_off + f) * a.n_mels + c % a.n_mels];
        if (a.affine_mode == 1) {
          v = x * a.cmvn_vars[c] + a.speech_pos[(size_t)j * a.feat + c];
        } else {
          v = (x + a.cmvn_means[c]) * a.cmvn_vars[c];
          v = v + a.speech_pos[(size_t)j * a.feat + c];
        }
      }
    }
    o[c] = v;
    if (olo) olo[c] = f32_to_bf16(v);
  }
}

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; the row (D <= 2048) is read ONCE as float4 per lane and held in registers; two-pass
// mean / variance on the registers (same arithmetic order class as torch's), wave-shuffle reductions.
constexpr int LN_MAXV = 8;   // float4 per lane: 8 * 4 * 64 = 2048 columns
template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ld_x, int rows, int D,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, OutT* out, int ld_out, int fill_to, const int32_t* __restrict__ rows_dev) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows || (rows_dev && row >= *rows_dev)) return;
  const float* xr = x + (size_t)row * ld_x;
  float4 v[LN_MAXV];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int i = k * 256 + lane * 4;
    v[k] = (i < D) ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int i = k * 256 + lane * 4;
    if (i < D) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  OutT* o = out + (size_t)row * ld_out;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int i = k * 256 + lane * 4;
    if (i < D) {
      float y0 = (v[k].x - mean) * rstd, y1 = (v[k].y - mean) * rstd, y2 = (v[k].z - mean) * rstd, y3 = (v[k].w - mean) * rstd;
      if (gamma) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + i);
        const float4 b = *reinterpret_cast<const float4*>(beta + i);
        y0 = y0 * g.x + b.x; y1 = y1 * g.y + b.y; y2 = y2 * g.z + b.z; y3 = y3 * g.w + b.w;
      }
      if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<float4*>(o + i) = make_float4(y0, y1, y2, y3);
      } else {
        uint2 w;
        w.x = pack_bf16x2(y0, y1);
        w.y = pack_bf16x2(y2, y3);
        *reinterpret_cast<uint2*>(o + i) = w;
      }
    }
  }
  for (int i = D + lane; i < fill_to; i += 64) Elem<OutT>::store(o + i, 0.0f);
}

// ------------------------------------------------------------------------------------ LayerNorm (affine-free) -> e4m3 bytes
// Operand rows of the FP8 matrix-pipe GEMM (precision mode ASR_PRECISION_FP8MM): y = (x - mean) * rstd * inv_scale, saturated at +-448, OCP e4m3 (RNE).
// One wave per row; D <= 2048, D % 256 == 0.
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, int ld_x, int rows, int D, float eps, float inv_scale,
                                                            unsigned char* __restrict__ out, int ld_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ld_x;
  float4 v[8];
  const int nv = D >> 8;                                       // float4 per lane
  float s1 = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < nv) { v[k] = *reinterpret_cast<const float4*>(xr + k * 256 + lane * 4); s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
  const float mean = wave_sum(s1) / (float)D;
  float s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < nv) { const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean; s2 += (a * a + b * b) + (c * c + d * d); }
  const float rstd = rsqrtf(wave_sum(s2) / (float)D + eps) * inv_scale;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < nv) {
      auto q = [&](float t) { return fminf(fmaxf((t - mean) * rstd, -448.0f), 448.0f); };
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(q(v[k].x), q(v[k].y), w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(q(v[k].z), q(v[k].w), w, true);
      *reinterpret_cast<int*>(out + (size_t)row * ld_out + k * 256 + lane * 4) = w;
    }
}

// ------------------------------------------------------------------------------------ attention (bf16, flash-style)
// One workgroup = a block of NW*QT 16-row query tiles of one (utterance, head); wave w owns tiles w, w+NW, ...
// (QT of them, all live in registers). K and V^T chunks of CHUNK keys are staged ONCE per workgroup by LDS-DMA
// (global_load_lds_dwordx4) with the 16-byte-slot XOR swizzle on the global source side, and every K / V^T
// fragment read from LDS is reused by the wave's QT query tiles. Scores are computed TRANSPOSED, S^T = K Q^T,
// so the C fragment puts one query per lane column (lane & 15) and 4 consecutive keys per lane: the row soft-max
// is lane-local plus two xor-shuffles across the four 16-lane groups, and the bf16-packed probabilities are
// directly the B fragment of O^T = V^T P^T (no LDS round trip for P). O^T's C fragment again has one query per
// lane column, so the online-softmax rescale is lane-local, and each lane ends with 4 consecutive d values of
// its query row (one 8-byte store).
// (a hand-written v_max3_f32 in inline asm would save the canonicalising v_max x, x that fmaxf() costs per MFMA output under IEEE rules, but the hazard recogniser
//  does not see inline asm: it placed the v_max3 directly behind the MFMA that writes its operands and results depended on timing -- round 6, measured as 21 failing
//  determinism tests. Plain fmaxf: the compiler still fuses pairs into v_max3.)
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int HD, int CHUNK, int QT>
__global__ __launch_bounds__(512, (HD * QT <= 128) ? 4 : (HD * QT <= 256 ? 2 : 1)) void attn_bf16_kernel(const AttnArgs a, int n_rows_alloc) {
  constexpr int SLOTS = HD / 8;               // 16-byte slots per K row
  constexpr int KROWB = HD * 2;               // bytes per K row
  constexpr int K_RPI = 64 / SLOTS;           // K rows per LDS-DMA wave-instruction (1 KiB)
  constexpr int VSLOTS = CHUNK / 8;           // 16-byte slots per V^T row
  constexpr int VROWB = CHUNK * 2;
  constexpr int V_RPI = 64 / VSLOTS;
  constexpr int V_NI = HD / V_RPI;
  static_assert(VSLOTS == 16 || VSLOTS == 32, "V^T swizzle assumes 16 or 32 slots per row");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + CHUNK * KROWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = blockDim.x >> 6;
  const int fq = lane & 15, g = lane >> 4;
  const int u = a.qb_utt[blockIdx.x], q_base = a.qb_q0[blockIdx.x], h = blockIdx.y;
  const UttPlan up = a.plan[u];
  const int T = up.T, row0 = up.row_off;                       // keys / values
  const int Tq = a.q_plan ? a.q_plan[u].T : T, rowq0 = a.q_plan ? a.q_plan[u].row_off : row0;

  bf16x8_t qf[QT][HD / 32];
  f32x4_t ot[QT][HD / 16];
  float m_run[QT], l_run[QT];
  bool act[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q0 = q_base + (t * NW + wave) * 16;
    act[t] = q0 < Tq;                         // wave-uniform
    m_run[t] = -INFINITY;
    l_run[t] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) ot[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(a.q) + (size_t)(rowq0 + (act[t] ? q0 : 0) + fq) * a.ld_q + h * HD + g * 8;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qf[t][ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
  }

  const int hk = h / a.kv_group;
  const bf16_t* kbase = reinterpret_cast<const bf16_t*>(a.k) + hk * HD;
  const bf16_t* vbase = reinterpret_cast<const bf16_t*>(a.vt) + (size_t)hk * HD * a.ld_vt;
  // causal: no query of this block sees keys past the block's last row
  const int Tk = a.causal ? min(T, q_base + 16 * QT * NW) : T;

  for (int kv0 = 0; kv0 < Tk; kv0 += CHUNK) {
    const int nkeys = min(CHUNK, (Tk - kv0 + 31) & ~31);       // keys of this chunk that are ever read (32-key sub-tiles)
    __syncthreads();                          // all waves finished reading the previous chunk
    for (int ii = wave; ii * K_RPI < nkeys; ii += NW) {
      const int key = ii * K_RPI + lane / SLOTS;
      const int sslot = (lane % SLOTS) ^ (key & (SLOTS - 1));
      const int grow = min(row0 + kv0 + key, n_rows_alloc - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(kbase + (size_t)grow * a.ld_qk + sslot * 8),
          (__attribute__((address_space(3))) void*)(Ks + ii * 1024), 16, 0, 0);
    }
    for (int ii = wave; ii < V_NI; ii += NW) {
      const int d = ii * V_RPI + lane / VSLOTS;
      const int sslot = (lane % VSLOTS) ^ (d & 15);
      if (sslot * 8 < nkeys) {                // slots beyond the last sub-tile are never read
        const int gcol = min(row0 + kv0 + sslot * 8, a.ld_vt - 8);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(vbase + (size_t)d * a.ld_vt + gcol),
            (__attribute__((address_space(3))) void*)(Vs + ii * 1024), 16, 0, 0);
      }
    }
    __syncthreads();                          // chunk landed (the barrier drains the LDS-DMA queue)
    // One 32-key sub-tile: S^T MFMAs, soft-max update, O^T MFMAs. The soft-max is what bounds this kernel at head_dim 64 (256 MFMA flops per score against
    // ~25 VALU cycles + a quarter-rate v_exp per score: round 5's body ran 86 full-rate VALU ops + 9 exps per tile and sub-tile, 488 issue cycles against 128
    // of MFMA), so: keys are masked only in sub-tiles that need it (the chunk's last one, or any under the causal rule); p = exp2(fma(s, log2 e, -m log2 e))
    // -- one packed FMA per two scores, no subtract / multiply pair; the running output is rescaled only when some lane's maximum moved (wave-uniform
    // test: after the first few sub-tiles it almost never does), with packed multiplies, outside the MFMA chain.
    const int perm16 = (lane ^ 16) << 2, perm32 = (lane ^ 32) << 2;       // ds_bpermute addresses of the two row-maximum exchanges
    auto subtile = [&](const int s, auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      constexpr float LOG2E = 1.4426950408889634f;
      f32x4_t st0[QT], st1[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) { st0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; st1[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
      // score row i of the two S^T tiles of a sub-tile = keys 8 (i / 4) + (i % 4) and + 4: lane group g then holds scores of the 8 CONSECUTIVE keys 8 g .. 8 g + 7
      // (tile 0: the first four, tile 1: the last four), so its packed probabilities are the B fragment for a V^T fragment that is ONE 16-byte LDS read
      const int key0 = s * 32 + ((fq >> 2) << 3) + (fq & 3), key1 = key0 + 4;
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const int c = ks * 4 + g;
        const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t*>(Ks + key0 * KROWB + ((c ^ (key0 & (SLOTS - 1))) << 4));
        const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t*>(Ks + key1 * KROWB + ((c ^ (key1 & (SLOTS - 1))) << 4));
#pragma unroll
        for (int t = 0; t < QT; ++t) {       // (tiles past the utterance's end run too -- straight-line code; nothing of theirs is ever stored)
          st0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[t][ks], st0[t], 0, 0, 0);
          st1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[t][ks], st1[t], 0, 0, 0);
        }
      }
      const int kb = kv0 + s * 32 + g * 8;       // this lane's keys: kb .. kb + 3 (tile 0), kb + 4 .. kb + 7 (tile 1)
      bf16x8_t pfv[QT];
      float alpha[QT];
      bool moved = false;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        f32x4_t s0 = st0[t], s1 = st1[t];
        if constexpr (MASKED) {
          const int qi = q_base + (t * NW + wave) * 16 + fq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kb + r >= T || (a.causal && kb + r > qi)) s0[r] = -INFINITY;
            if (kb + 4 + r >= T || (a.causal && kb + 4 + r > qi)) s1[r] = -INFINITY;
          }
        }
        float mx = vmax3(vmax3(s0[0], s0[1], s0[2]), vmax3(s0[3], s1[0], s1[1]), vmax3(s1[2], s1[3], s1[3]));
        mx = vmax3(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm16, __builtin_bit_cast(int, mx))), mx);
        const float m_old = m_run[t], m_new = vmax3(m_old, mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm32, __builtin_bit_cast(int, mx))));
        m_run[t] = m_new;
        moved = moved || (m_new != m_old);
        alpha[t] = __builtin_amdgcn_exp2f((m_old - m_new) * LOG2E);
        const f32x2_t mneg = {-m_new * LOG2E, -m_new * LOG2E}, l2 = {LOG2E, LOG2E};
        const f32x2_t e0 = __builtin_elementwise_fma(f32x2_t{s0[0], s0[1]}, l2, mneg), e1 = __builtin_elementwise_fma(f32x2_t{s0[2], s0[3]}, l2, mneg);
        const f32x2_t e2 = __builtin_elementwise_fma(f32x2_t{s1[0], s1[1]}, l2, mneg), e3 = __builtin_elementwise_fma(f32x2_t{s1[2], s1[3]}, l2, mneg);
        const f32x2_t p0 = {__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])}, p1 = {__builtin_amdgcn_exp2f(e1[0]), __builtin_amdgcn_exp2f(e1[1])};
        const f32x2_t p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])}, p3 = {__builtin_amdgcn_exp2f(e3[0]), __builtin_amdgcn_exp2f(e3[1])};
        const f32x2_t ps = (p0 + p1) + (p2 + p3);
        l_run[t] = fmaf(l_run[t], alpha[t], ps[0] + ps[1]);
        union { bf16x8_t v; uint32_t w[4]; } pf;
        pf.w[0] = pack_bf16x2(p0[0], p0[1]); pf.w[1] = pack_bf16x2(p1[0], p1[1]); pf.w[2] = pack_bf16x2(p2[0], p2[1]); pf.w[3] = pack_bf16x2(p3[0], p3[1]);
        pfv[t] = pf.v;
      }
      if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
          for (int dt = 0; dt < HD / 16; ++dt) ot[t][dt] *= alpha[t];
      }
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const int d = dt * 16 + fq;
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + d * VROWB + (((s * 4 + g) ^ (d & 15)) << 4));
#pragma unroll
        for (int t = 0; t < QT; ++t) ot[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfv[t], ot[t][dt], 0, 0, 0);
      }
    };
    // sub-tiles whose 32 keys all exist need no mask unless the causal rule cuts into this chunk
    const int n_sub = (nkeys + 31) >> 5;
    const int n_plain = (a.causal && kv0 + nkeys > q_base) ? 0 : min(n_sub, (T - kv0) >> 5);
    for (int s = 0; s < n_plain; ++s) subtile(s, std::false_type{});
    for (int s = n_plain; s < n_sub; ++s) subtile(s, std::true_type{});
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = q_base + (t * NW + wave) * 16 + fq;
    if (!act[t]) continue;
    float l = l_run[t] + __shfl_xor(l_run[t], 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qrow < Tq) {
      bf16_t* op = reinterpret_cast<bf16_t*>(a.ctx) + (size_t)(rowq0 + qrow) * a.ld_ctx + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        uint2 w;
        w.x = pack_bf16x2(ot[t][dt][0] * inv, ot[t][dt][1] * inv);
        w.y = pack_bf16x2(ot[t][dt][2] * inv, ot[t][dt][3] * inv);
        *reinterpret_cast<uint2*>(op + dt * 16) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ attention (f32, verification mode)
// One wave per query row; lanes over keys for the scores, lanes over d for the context. Plain f32 FMA. A query block of the plan (64 rows)
// is spread over gridDim.z workgroups (wave w of workgroup z takes rows w + 4 z, w + 4 z + 4 gridDim.z, ...): a single 8 s window is then
// ~140 workgroups instead of 12 (the row's arithmetic and its order are unchanged).
constexpr int ATT32_MAXT = 2048;
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnArgs a, int HD) {
  __shared__ float sc[4][ATT32_MAXT];
  __shared__ float qs[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = a.qb_utt[blockIdx.x], q0 = a.qb_q0[blockIdx.x], h = blockIdx.y;
  const UttPlan up = a.plan[u];
  const int T = up.T, row0 = up.row_off;
  const int Tq = a.q_plan ? a.q_plan[u].T : T, rowq0 = a.q_plan ? a.q_plan[u].row_off : row0;
  const float* Q = reinterpret_cast<const float*>(a.q);
  const float* K = reinterpret_cast<const float*>(a.k);
  const float* Vt = reinterpret_cast<const float*>(a.vt);
  float* C = reinterpret_cast<float*>(a.ctx);
  for (int qi = wave + 4 * blockIdx.z; qi < 64; qi += 4 * gridDim.z) {
    const int qrow = q0 + qi;
    if (qrow >= Tq) break;                     // wave-uniform
    const float* qp = Q + (size_t)(rowq0 + qrow) * a.ld_q + h * HD;
    for (int d = lane; d < HD; d += 64) qs[wave][d] = qp[d];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int key = lane; key < T; key += 64) {
      const float* kp = K + (size_t)(row0 + key) * a.ld_qk + h * HD;
      float acc = 0.0f;
      for (int d = 0; d < HD; d += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(kp + d);
        acc = fmaf(qs[wave][d], kv.x, acc);
        acc = fmaf(qs[wave][d + 1], kv.y, acc);
        acc = fmaf(qs[wave][d + 2], kv.z, acc);
        acc = fmaf(qs[wave][d + 3], kv.w, acc);
      }
      sc[wave][key] = acc;
      mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int key = lane; key < T; key += 64) {
      const float e = expf(sc[wave][key] - mx);
      sc[wave][key] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    for (int d = lane; d < HD; d += 64) {
      const float* vp = Vt + (size_t)(h * HD + d) * a.ld_vt + row0;
      float acc = 0.0f;
      for (int key = 0; key < T; ++key) acc = fmaf(sc[wave][key], vp[key], acc);
      C[(size_t)(rowq0 + qrow) * a.ld_ctx + h * HD + d] = acc * inv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------ FSMN
// Depth-wise conv over time on the time-contiguous V^T: each thread produces 8 consecutive time steps of one
// channel from three aligned 8-element loads (previous / current / next group), zeroing taps that fall
// outside the utterance [s, e) (symmetric zero padding per utterance, Export_SenseVoice.py:220).
template <typename InT> __device__ __forceinline__ void load8(const InT* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&o)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// One workgroup = 64 time steps x 64 channels. Phase 1 (lanes along time, coalesced on V^T): each thread
// computes 8 consecutive time steps of one channel in registers; phase 2: LDS transpose so the f32 memory is
// written ROW-MAJOR [time][channel] with 16-byte stores (it is the out-projection GEMM's additive operand).
template <typename InT, int KT>
__global__ __launch_bounds__(256) void fsmn_kernel(const InT* __restrict__ vt, int ld, const float* __restrict__ w,
                                                   const float* __restrict__ b, const UttPlan* __restrict__ plan,
                                                   const int32_t* __restrict__ row_utt, int C, float* __restrict__ out,
                                                   int ld_out) {
  constexpr int PAD = (KT - 1) / 2;
  constexpr int TLD = 65;
  __shared__ float tile[64 * TLD];
  const int tid = threadIdx.x;
  const int m_base = blockIdx.x * 64, c_base = blockIdx.y * 64;
  const int mg = tid & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = pass * 32 + (tid >> 3);
    const int c = c_base + cl;
    const int m = m_base + mg * 8;
    const int u = row_utt[m];
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (u >= 0) {
      const int s = plan[u].row_off, e = s + plan[u].T;
      if (m < e) {
        float x[24];
        const InT* v = vt + (size_t)c * ld + m;
        float t8[8];
        if (m >= 8) { load8<InT>(v - 8, t8); } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) t8[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = t8[i];
        load8<InT>(v, t8);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[8 + i] = t8[i];
        if (m + 8 < ld) { load8<InT>(v + 8, t8); } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) t8[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[16 + i] = t8[i];
#pragma unroll
        for (int i = 0; i < 24; ++i) { const int mm = m - 8 + i; if (mm < s || mm >= e) x[i] = 0.f; }
        float wc[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) wc[j] = w[c * KT + j];
        const float bc = b[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float acc = bc;
#pragma unroll
          for (int j = 0; j < KT; ++j) acc = fmaf(wc[j], x[8 + i + j - PAD], acc);
          o[i] = (m + i < e) ? acc : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[(mg * 8 + i) * TLD + cl] = o[i];
  }
  __syncthreads();
  const int q = (tid & 15) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int ml = pass * 16 + (tid >> 4);
    const float* t = tile + ml * TLD + q;
    *reinterpret_cast<float4*>(out + (size_t)(m_base + ml) * ld_out + c_base + q) = make_float4(t[0], t[1], t[2], t[3]);
  }
}

// ------------------------------------------------------------------------------------ CTC collapse
__global__ __launch_bounds__(256) void ctc_collapse_kernel(const int32_t* __restrict__ ids, const UttPlan* __restrict__ plan,
                                                           int blank_id, int32_t* __restrict__ token_ids, int max_tokens,
                                                           int32_t* __restrict__ num_id) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = plan[u].T, base = plan[u].row_off;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    int id = 0;
    bool keep = false;
    if (t < T) {
      id = ids[base + t];
      const int nxt = ids[base + ((t + 1 == T) ? 0 : t + 1)];   // circular: last frame vs first
      keep = (id != nxt) && (id != blank_id);
    }
    const unsigned long long bal = __ballot(keep);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = running + prefix;
    for (int w2 = 0; w2 < wave; ++w2) off += wave_cnt[w2];
    if (keep && off < max_tokens) token_ids[(size_t)u * max_tokens + off] = id;
    __syncthreads();
    if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) num_id[u] = running;
}


// ------------------------------------------------------------------------------------ Whisper log-mel finish / conv padding
__device__ __forceinline__ int gapped_frame(const UttPlan& up, int j) {   // frame index of gapped row j, or -1
  const int f = j - 2 * up.row_off - 1;
  return (f >= 0 && f < up.n_frames) ? f : -1;
}

template <typename T>
__global__ void whisper_mel_finish_kernel(const float* __restrict__ mel, const float* __restrict__ blk_max,
                                          const UttPlan* __restrict__ plan, const int32_t* __restrict__ grow_utt, int n_mels,
                                          T* __restrict__ out) {
  const int j = blockIdx.x;
  T* o = out + (size_t)j * n_mels;
  const int u = grow_utt[j];
  int f = -1;
  UttPlan up;
  if (u >= 0) { up = plan[u]; f = gapped_frame(up, j); }
  if (f < 0) {
    for (int c = threadIdx.x; c < n_mels; c += blockDim.x) Elem<T>::store(o + c, 0.0f);
    return;
  }
  float gmax = -INFINITY;                                   // global max over the clip (Export_Whisper.py:426)
  const int nblk = (up.n_frames + FB_FRAMES - 1) / FB_FRAMES;
  for (int k = 0; k < nblk; ++k) gmax = fmaxf(gmax, blk_max[up.blk0 + k]);
  for (int c = threadIdx.x; c < n_mels; c += blockDim.x) {
    float v = mel[(size_t)(up.frame_off + f) * n_mels + c];
    v = fmaxf(v, gmax - 8.0f);
    Elem<T>::store(o + c, (v + 4.0f) * 0.25f);
  }
}

template <typename T>
__global__ void zero_gap_rows_kernel(T* __restrict__ buf, int ld, int n_cols, const UttPlan* __restrict__ plan,
                                     const int32_t* __restrict__ grow_utt) {
  const int j = blockIdx.x;
  const int u = grow_utt[j];
  if (u >= 0 && gapped_frame(plan[u], j) >= 0) return;
  T* o = buf + (size_t)j * ld;
  for (int c = threadIdx.x; c < n_cols; c += blockDim.x) Elem<T>::store(o + c, 0.0f);
}

// rows of a packed f32 matrix from one utterance plan's row space to another's (Whisper: conv stem rows -> encoder rows); pad rows become zero
__global__ void compact_rows_kernel(const float* __restrict__ src, const UttPlan* __restrict__ from, const UttPlan* __restrict__ to,
                                    const int32_t* __restrict__ row_utt, int d, float* __restrict__ dst) {
  const int m = blockIdx.x, u = row_utt[m];
  float4* o = reinterpret_cast<float4*>(dst + (size_t)m * d);
  const int t = u >= 0 ? m - to[u].row_off : -1;
  if (u < 0 || t >= to[u].T) {
    for (int c = threadIdx.x; c < d / 4; c += blockDim.x) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4* i = reinterpret_cast<const float4*>(src + (size_t)(from[u].row_off + t) * d);
  for (int c = threadIdx.x; c < d / 4; c += blockDim.x) o[c] = i[c];
}

// ------------------------------------------------------------------------------------ decoder embedding
template <typename T>
__global__ void embed_pos_kernel(const int32_t* __restrict__ ids, int n, int hist, const int32_t* __restrict__ hist_dev,
                                 const T* __restrict__ embed, const float* __restrict__ pos, int d, float* __restrict__ x) {
  const int r = blockIdx.x;
  if (hist_dev) hist = *hist_dev;
  const T* e = embed + (size_t)ids[r] * d;
  const float* p = pos + (size_t)(hist + r % n) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) x[(size_t)r * d + c] = Elem<T>::load(e + c) + p[c];
}

// ------------------------------------------------------------------------------------ decoder attention (head_dim 64)
// One workgroup per (sequence, head). Rows of 64 are read cooperatively: 8 lanes per row (8 dims each), 8 rows per
// wave-instruction = 1 KiB contiguous (self cache / cross slab are [row][64]), so the K and V streams are fully
// coalesced -- the cross-attention stream is the HBM-bound part of a decode step.
constexpr int DA_MAXN = 8;
constexpr int DA_MAXKEYS = 1536;

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&o)[8]) const {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float (&o)[8]) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};

// 8 OCP e4m3 bytes of a cached K / V row (FP8 cross-K/V, precision mode ASR_PRECISION_FP8W); the slab's scale is applied by the caller
struct fp8_t { unsigned char v; };
template <> struct Raw8<fp8_t> {
  uint2 v;
  __device__ __forceinline__ void load(const fp8_t* p) { v = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void get(float (&o)[8]) const {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(v.x, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(v.y, true);
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1]; o[6] = d[0]; o[7] = d[1];
  }
};

template <typename T> __device__ __forceinline__ void load8dims(const T* p, float (&o)[8]) { load8<T>(p, o); }

// NQ = compile-time bound on the queries per sequence: 1 for single-token decode steps (lean registers: more loads in flight),
// DA_MAXN for prefills.
// KV8: the cached K / V rows are e4m3 bytes with one power-of-two scale per (sequence, head) slab (cross-attention only: no new rows)
template <typename T, int NQ, bool KV8 = false>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecAttnArgs a) {
  using KT = typename std::conditional<KV8, fp8_t, T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char da_smem[];
  float* sc = reinterpret_cast<float*>(da_smem);          // [n][sc_ld]
  const int sc_ld = a.sc_ld;
  __shared__ float qs[NQ][64];
  __shared__ float red[4][NQ][64];
  __shared__ float stat[2][NQ];
  const int b = blockIdx.x + a.b0, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 7;                     // which 8 dims of the row
  const int n = a.n;
  const int hist = a.hist_dev ? *a.hist_dev : a.hist;
  int n_cached, row0 = 0;
  if (a.plan) { n_cached = a.plan[b].n_lfr; row0 = a.plan[b].row_off; }
  else n_cached = hist;
  const int S = a.plan ? n_cached : hist + n;
  // cache row s of this (sequence, head): contiguous extent, or through the block table of the paged self-KV cache (pages of 16 positions)
  const int32_t* pt = a.page_table ? a.page_table + (size_t)b * a.pages_per_seq : nullptr;
  const KT* Kc = reinterpret_cast<const KT*>(a.k_base) + (pt ? (size_t)h * 1024 : (size_t)b * a.stride_b + (size_t)h * a.stride_h + (size_t)row0 * 64);
  const KT* Vc = reinterpret_cast<const KT*>(a.v_base) + (pt ? (size_t)h * 1024 : (size_t)b * a.stride_b + (size_t)h * a.stride_h + (size_t)row0 * 64);
  auto crow = [&](int s) -> size_t { return pt ? (size_t)pt[s >> 4] * (size_t)a.page_stride + (size_t)(s & 15) * 64 : (size_t)s * 64; };
  const T* Q = reinterpret_cast<const T*>(a.q);
  const T* NEW = KV8 ? nullptr : reinterpret_cast<const T*>(a.kv_new);
  const int s_ld = a.scale_ld ? a.scale_ld : (int)gridDim.x;
  const float k_scale = KV8 ? a.k_scale[(size_t)h * s_ld + b] : 1.0f, v_scale = KV8 ? a.v_scale[(size_t)h * s_ld + b] : 1.0f;

  for (int i = tid; i < n * 64; i += 256) qs[i >> 6][i & 63] = Elem<T>::load(Q + (size_t)(b * n + (i >> 6)) * a.ld_q + a.q_col0 + h * 64 + (i & 63));
  if constexpr (!KV8) if (NEW) {               // append the new rows to the cache (read back only by later steps)
    T* Kw = const_cast<T*>(Kc);
    T* Vw = const_cast<T*>(Vc);
    for (int i = tid; i < n * 64; i += 256) {
      const size_t src = (size_t)(b * n + (i >> 6)) * a.ld_new + h * 64 + (i & 63);
      Kw[crow(hist + (i >> 6)) + (i & 63)] = NEW[src + a.k_col0];
      Vw[crow(hist + (i >> 6)) + (i & 63)] = NEW[src + a.v_col0];
    }
  }
  __syncthreads();
  // ---- scores: the row segments of the NEXT trip are already in flight while this trip is reduced
  constexpr int DU = 8;                          // rows in flight per 8-lane group and trip
  auto k_ptr = [&](int s0) -> const KT* {
    if constexpr (KV8) return Kc + (size_t)s0 * 64 + sub * 8;
    else return s0 < n_cached ? Kc + crow(s0) + sub * 8 : NEW + (size_t)(b * n + (s0 - n_cached)) * a.ld_new + a.k_col0 + h * 64 + sub * 8;
  };
  auto v_ptr = [&](int s0) -> const KT* {
    if constexpr (KV8) return Vc + (size_t)s0 * 64 + sub * 8;
    else return s0 < n_cached ? Vc + crow(s0) + sub * 8 : NEW + (size_t)(b * n + (s0 - n_cached)) * a.ld_new + a.v_col0 + h * 64 + sub * 8;
  };
  {
    Raw8<KT> nxt[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) { const int s0 = (tid >> 3) + u * 32; if (s0 < S) nxt[u].load(k_ptr(s0)); }
    for (int sb = (tid >> 3); sb < S; sb += 32 * DU) {
      Raw8<KT> cur[DU];
#pragma unroll
      for (int u = 0; u < DU; ++u) cur[u] = nxt[u];
#pragma unroll
      for (int u = 0; u < DU; ++u) { const int s1 = sb + 32 * DU + u * 32; if (s1 < S) nxt[u].load(k_ptr(s1)); }
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int s0 = sb + u * 32;
        if (s0 < S) {
          float kvf[8];
          cur[u].get(kvf);
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            if (i >= n) break;
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qs[i][sub * 8 + e], kvf[e], acc);
            // sum over the 8 lanes of a row in three DPP moves (xor 1, xor 2, mirror within 8) instead of LDS-crossbar shuffles
            acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xf, 0xf, true));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xf, 0xf, true));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x141, 0xf, 0xf, true));
            if (sub == 0) sc[i * sc_ld + s0] = acc * k_scale + ((a.causal && s0 > hist + i) ? -128.0f : 0.0f);
          }
        }
      }
    }
  }
  __syncthreads();
  // the V stream does not depend on the scores: its first trip is requested before the soft-max statistics are computed
  Raw8<KT> vnxt[DU];
#pragma unroll
  for (int u = 0; u < DU; ++u) { const int s0 = (tid >> 3) + u * 32; if (s0 < S) vnxt[u].load(v_ptr(s0)); }
  // ---- soft-max statistics per query (wave w handles queries w, w+4)
  for (int i = wave; i < n; i += 4) {
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, sc[i * sc_ld + s]);
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int s = lane; s < S; s += 64) {
      const float e = expf(sc[i * sc_ld + s] - mx);
      sc[i * sc_ld + s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) stat[0][i] = 1.0f / sum;
  }
  __syncthreads();
  // ---- context: each 8-lane group walks keys s = group, group + 32, ...
  float acc[NQ][8];
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.0f;
  {
    for (int sb = (tid >> 3); sb < S; sb += 32 * DU) {
      Raw8<KT> cur[DU];
#pragma unroll
      for (int u = 0; u < DU; ++u) cur[u] = vnxt[u];
#pragma unroll
      for (int u = 0; u < DU; ++u) { const int s1 = sb + 32 * DU + u * 32; if (s1 < S) vnxt[u].load(v_ptr(s1)); }
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int s0 = sb + u * 32;
        if (s0 < S) {
          float vf[8];
          cur[u].get(vf);
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            if (i < n) {
              const float p = sc[i * sc_ld + s0];
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[i][e] = fmaf(p, vf[e], acc[i][e]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    if (i < n) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[i][e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8) red[wave][i][sub * 8 + e] = v;
      }
    }
  }
  __syncthreads();
  T* O = reinterpret_cast<T*>(a.out);
  for (int i = tid; i < n * 64; i += 256) {
    const int qi = i >> 6, dd = i & 63;
    const float v = (red[0][qi][dd] + red[1][qi][dd]) + (red[2][qi][dd] + red[3][qi][dd]);
    Elem<T>::store(O + (size_t)(b * n + qi) * a.ld_out + h * 64 + dd, v * stat[0][qi] * v_scale);
  }
}

// Single-token CROSS-attention (n = 1, ragged extents from the plan, no new rows): the K and the V stream of a (sequence, head) slab in ONE pass with a running soft-max --
// no score buffer, no barrier between the streams, both in flight together (the two-pass kernel above is a chain of two memory round trips + a one-wave soft-max in
// between: 19 us for 65 MB at 32 x 8 s). An 8-lane group walks rows g, g + 32, ... four at a time (the next four K and V rows are requested before the current four are
// reduced); groups keep (max, sum, 8 context dims per lane), merged through shuffles inside a wave and through LDS across the four waves.
template <typename T, bool KV8>
__global__ __launch_bounds__(256) void decode_cross_attn_kernel(const DecAttnArgs a) {
  using KT = typename std::conditional<KV8, fp8_t, T>::type;
  __shared__ float qs[64];
  __shared__ float part[4][64];
  __shared__ float part_ml[4][2];
  const int b = blockIdx.x + a.b0, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 7, grp = tid >> 3;
  const int S = a.plan[b].n_lfr, row0 = a.plan[b].row_off;
  const KT* Kc = reinterpret_cast<const KT*>(a.k_base) + (size_t)b * a.stride_b + (size_t)h * a.stride_h + (size_t)row0 * 64 + sub * 8;
  const KT* Vc = reinterpret_cast<const KT*>(a.v_base) + (size_t)b * a.stride_b + (size_t)h * a.stride_h + (size_t)row0 * 64 + sub * 8;
  const int s_ld = a.scale_ld ? a.scale_ld : (int)gridDim.x;
  const float k_scale = KV8 ? a.k_scale[(size_t)h * s_ld + b] : 1.0f, v_scale = KV8 ? a.v_scale[(size_t)h * s_ld + b] : 1.0f;
  constexpr int DU = 4;
  Raw8<KT> kn[DU], vn[DU];
#pragma unroll
  for (int u = 0; u < DU; ++u) { const int s0 = grp + u * 32; if (s0 < S) { kn[u].load(Kc + (size_t)s0 * 64); vn[u].load(Vc + (size_t)s0 * 64); } }
  if (tid < 64) qs[tid] = Elem<T>::load(reinterpret_cast<const T*>(a.q) + (size_t)b * a.ld_q + a.q_col0 + h * 64 + tid);
  __syncthreads();
  float q8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q8[e] = qs[sub * 8 + e];
  float m = -INFINITY, l = 0.0f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
  for (int sb = grp; sb < S; sb += 32 * DU) {
    Raw8<KT> kc[DU], vc[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) { kc[u] = kn[u]; vc[u] = vn[u]; }
#pragma unroll
    for (int u = 0; u < DU; ++u) { const int s1 = sb + 32 * DU + u * 32; if (s1 < S) { kn[u].load(Kc + (size_t)s1 * 64); vn[u].load(Vc + (size_t)s1 * 64); } }
    float x[DU], mt = -INFINITY;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      float kf[8];
      kc[u].get(kf);
      float d = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(q8[e], kf[e], d);
      d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));
      d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, true));
      d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xf, 0xf, true));
      x[u] = (sb + u * 32 < S) ? d * k_scale : -INFINITY;
      mt = fmaxf(mt, x[u]);
    }
    const float m_new = fmaxf(m, mt);                         // (finite: row sb itself is valid)
    const float alpha = __expf(m - m_new);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const float p = __expf(x[u] - m_new);                   // 0 for the rows past S
      float vf[8];
      vc[u].get(vf);
      l += p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, (sb + u * 32 < S) ? vf[e] : 0.0f, acc[e]);
    }
    m = m_new;
  }
  // ---- the eight groups of a wave, then the four waves
  float mw = m;
  mw = fmaxf(mw, __shfl_xor(mw, 8, 64)); mw = fmaxf(mw, __shfl_xor(mw, 16, 64)); mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  const float sc = (m == -INFINITY) ? 0.0f : __expf(m - mw);   // (a group without rows: S < 32)
  l *= sc;
  l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e] * sc;
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (lane < 8) part[wave][sub * 8 + e] = v;
  }
  if (lane == 0) { part_ml[wave][0] = mw; part_ml[wave][1] = l; }
  __syncthreads();
  if (tid < 64) {
    const float M = fmaxf(fmaxf(part_ml[0][0], part_ml[1][0]), fmaxf(part_ml[2][0], part_ml[3][0]));
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (part_ml[w][0] == -INFINITY) ? 0.0f : __expf(part_ml[w][0] - M);
      num = fmaf(part[w][tid], f, num);
      den = fmaf(part_ml[w][1], f, den);
    }
    Elem<T>::store(reinterpret_cast<T*>(a.out) + (size_t)b * a.ld_out + h * 64 + tid, num / den * v_scale);
  }
}

// ------------------------------------------------------------------------------------ row arg-max
// one workgroup of 1024 threads per row, 16-byte loads, two independent loads in flight per thread (a 150 k-column row is ~19 trips)
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ logits, int ld, int n_valid,
                                                           const float* __restrict__ extra, int32_t* __restrict__ ids) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (size_t)r * ld;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  auto take = [&](const float4 v, int c) {                    // ascending columns: strict > keeps the first maximum
    if (c + 0 < n_valid && v.x > best) { best = v.x; bidx = c; }
    if (c + 1 < n_valid && v.y > best) { best = v.y; bidx = c + 1; }
    if (c + 2 < n_valid && v.z > best) { best = v.z; bidx = c + 2; }
    if (c + 3 < n_valid && v.w > best) { best = v.w; bidx = c + 3; }
  };
  auto load = [&](int c) {
    float4 v = *reinterpret_cast<const float4*>(x + c);       // rows are padded to a multiple of 128 columns
    if (extra) { const float4 e = *reinterpret_cast<const float4*>(extra + c); v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
    return v;
  };
  int c = tid * 4;
  for (; c + 4096 < n_valid; c += 8192) {
    const float4 a = load(c), b = load(c + 4096);
    take(a, c);
    take(b, c + 4096);
  }
  if (c < n_valid) take(load(c), c);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
    ids[r] = bidx;
  }
}


__global__ __launch_bounds__(1024) void no_speech_prob_kernel(const float* __restrict__ logits, int ld, int n_valid, const float* __restrict__ penalty,
                                                              int no_speech_id, float* __restrict__ prob) {
  __shared__ float red[16];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (size_t)r * ld;
  float mx = -INFINITY;
  for (int c = tid; c < n_valid; c += 1024) mx = fmaxf(mx, x[c] - penalty[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float sum = 0.0f;
  for (int c = tid; c < n_valid; c += 1024) sum += expf(x[c] - penalty[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int w = 0; w < 16; ++w) t += red[w];
    prob[r] = expf(x[no_speech_id] - penalty[no_speech_id] - mx) / t;
  }
}

__global__ __launch_bounds__(64) void apply_penalty_kernel(float* __restrict__ logits, int ld, const int32_t* __restrict__ save_ids,
                                                           int ld_save, const int32_t* __restrict__ n_saved, int range, float value, int partial) {
  const int n = *n_saved;
  if (n < range) {
    if (!partial || n == 0) return;            // Whisper's host: multiplier 1.0 until the window is full; Qwen3-ASR: save_id[:, -range:] of what exists
    range = n;
  }
  float* x = logits + (size_t)blockIdx.x * ld;
  const int32_t* sv = save_ids + (size_t)blockIdx.x * ld_save + (n - range);
  const int i = threadIdx.x;                   // range <= 64: every original value is gathered before the first write
  const int id = i < range ? sv[i] : -1;
  const float v = id >= 0 ? x[id] : 0.0f;
  __syncthreads();
  if (id >= 0) x[id] = v * value;
}

__device__ __forceinline__ float uniform_from_counter(uint64_t seed, uint32_t step, uint32_t row, uint32_t j) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)step * 0x100000001B3ull + ((uint64_t)row << 8) + j + 1);   // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void sample_topk_topp_kernel(const SampleArgs a) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  __shared__ float top_v[64];
  __shared__ int top_i[64];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* x = a.logits + (size_t)r * a.ld;
  const int n_prev = min(*a.n_saved, a.ld_save);
  // 1. repetition penalty over the whole history (<= 1024 ids: four per thread), every original gathered before the first write
  const int32_t* sv = a.save_ids + (size_t)r * a.ld_save;
  int idh[4];
  float vh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    idh[j] = tid + 256 * j < n_prev ? sv[tid + 256 * j] : -1;
    vh[j] = idh[j] >= 0 ? x[idh[j]] : 0.0f;
  }
  __syncthreads();
  const float rp = a.repetition_penalty, irp = 1.0f / a.repetition_penalty;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (idh[j] >= 0) x[idh[j]] = vh[j] < 0.0f ? vh[j] * rp : vh[j] * irp;
  __syncthreads();
  // 2. top-k by k passes of (value desc, index asc) selection below the previous pick
  const float inv_t = 1.0f / a.temperature;
  float last_v = INFINITY;
  int last_i = -1;
  for (int k = 0; k < a.top_k; ++k) {
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = tid; c < a.n_valid; c += 256) {
      const float v = (x[c] + (a.extra ? a.extra[c] : 0.0f)) * inv_t;
      const bool below = v < last_v || (v == last_v && c > last_i);
      if (below && (v > best || (v == best && c < bidx))) { best = v; bidx = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bidx, o, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
    __syncthreads();
    best = bv[0]; bidx = bi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
    if (tid == 0) { top_v[k] = best; top_i[k] = bidx; }
    last_v = best; last_i = bidx;
    __syncthreads();
  }
  // 3. soft-max over the k sorted scores, exclusive-cumsum top-p cut, Gumbel-max
  if (tid == 0) {
    const int K = a.top_k;
    float sum = 0.0f;
    for (int k = 0; k < K; ++k) sum += expf(top_v[k] - top_v[0]);
    float cum = 0.0f, best = -INFINITY;
    int win = 0;
    const uint32_t step = (uint32_t)*a.n_saved;
    for (int k = 0; k < K; ++k) {
      const float p = expf(top_v[k] - top_v[0]) / sum;
      const bool keep = cum <= a.top_p;              // (cumsum - p) <= top_p, cumsum taken before adding p
      cum += p;
      float u = a.noise ? a.noise[(size_t)r * K + k] : uniform_from_counter(a.seed, step, (uint32_t)r, (uint32_t)k);
      u = fminf(fmaxf(u, 1.0e-7f), 1.0f - 1.0e-7f);
      const float sc = keep ? top_v[k] - logf(-logf(u)) : -INFINITY;
      if (sc > best) { best = sc; win = k; }
    }
    a.next[r] = top_i[win];
  }
}

__global__ void append_ids_kernel(const int32_t* __restrict__ next, int rows, int32_t* __restrict__ save_ids, int ld_save,
                                  const int32_t* __restrict__ n_saved) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = *n_saved;
  if (r < rows && n < ld_save) save_ids[(size_t)r * ld_save + n] = next[r];
}

// ------------------------------------------------------------------------------------ Paraformer predictor / decoder helpers
template <typename T>
__global__ void shift3_kernel(const T* __restrict__ x, int d, const UttPlan* __restrict__ plan, const int32_t* __restrict__ row_utt,
                              T* __restrict__ out) {
  const int m = blockIdx.x;
  const int u = row_utt[m];
  int s = 0, e = 0;
  if (u >= 0) { s = plan[u].row_off; e = s + plan[u].T; }
  for (int c = threadIdx.x; c < 3 * d; c += blockDim.x) {
    const int tap = c / d, cc = c - tap * d;
    const int mm = m + tap - 1;
    const bool ok = (u >= 0) && (m < e) && (mm >= s) && (mm < e);
    out[(size_t)m * 3 * d + c] = ok ? x[(size_t)mm * d + cc] : T(0);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void alpha_kernel(const T* __restrict__ h, int d, const float* __restrict__ w, const float* __restrict__ b,
                                                    int rows, float* __restrict__ alpha) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= rows) return;
  float acc = 0.f;
  for (int c = lane; c < d; c += 64) acc = fmaf(Elem<T>::load(h + (size_t)m * d + c), w[c], acc);
  acc = wave_sum(acc);
  if (lane == 0) alpha[m] = 1.0f / (1.0f + expf(-(acc + b[0])));
}

__global__ void cif_scan_kernel(const float* __restrict__ alpha, const float* __restrict__ enc, int d, const UttPlan* __restrict__ plan,
                                float tail, float* __restrict__ acoustic, UttPlan* __restrict__ tplan, int32_t* __restrict__ num_id) {
  const int u = blockIdx.x;
  const UttPlan up = plan[u];
  const int T = up.T, row0 = up.row_off;
  const int rows16 = (T + 15) & ~15;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    double psum = 0.0;                        // FunASR accumulates alpha in float64 and rounds once (:505-507)
    float hsum = 0.f, prev_floor = 0.f, prev_completed = 0.f;
    int k = 0;
    for (int t = 0; t <= T; ++t) {
      const float a = (t < T) ? alpha[row0 + t] : tail;
      const float hv = (t < T) ? enc[(size_t)(row0 + t) * d + c] : 0.f;
      psum += (double)a;
      const float p32 = (float)psum, fl = floorf(p32);
      hsum = __fadd_rn(hsum, __fmul_rn(a, hv));              // f32 prefix sum of alpha * hidden (torch.cumsum order)
      if (fl > prev_floor) {
        const float completed = __fsub_rn(hsum, __fmul_rn(p32 - fl, hv));
        if (k < rows16) acoustic[(size_t)(row0 + k) * d + c] = completed - prev_completed;
        prev_completed = completed;
        ++k;
      }
      prev_floor = fl;
    }
    for (int r = k; r < rows16; ++r) acoustic[(size_t)(row0 + r) * d + c] = 0.f;   // incl. the dummy row of a zero-fire clip
    if (c == 0) {
      UttPlan tp = up;
      tp.n_lfr = k;
      tp.T = k > 0 ? k : 1;                   // Conv1d cannot take an empty token axis: one zero row, dropped at the end (:525-530)
      tplan[u] = tp;
      num_id[u] = k;
    }
  }
}

__global__ __launch_bounds__(256) void fsmn_rows_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w,
                                                        int d, int ktaps, const UttPlan* __restrict__ tplan,
                                                        const int32_t* __restrict__ row_utt, float* __restrict__ out,
                                                        const int32_t* __restrict__ rows_dev) {
  const int m = blockIdx.x;
  if (rows_dev && m >= *rows_dev) return;
  const int u = row_utt[m];
  int s = 0, e = 0;
  if (u >= 0) { s = tplan[u].row_off; e = s + tplan[u].T; }
  const int pad = (ktaps - 1) / 2;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    if (m >= s && m < e) {
      acc = res[(size_t)m * d + c];
      for (int j = 0; j < ktaps; ++j) {
        const int mm = m + j - pad;
        if (mm >= s && mm < e) acc = fmaf(w[c * ktaps + j], x[(size_t)mm * d + c], acc);
      }
    }
    out[(size_t)m * d + c] = acc;
  }
}

__global__ __launch_bounds__(256) void token_compact_kernel(const UttPlan* __restrict__ own, int n_utts, int n_rows_max,
                                                            UttPlan* __restrict__ compact, int32_t* __restrict__ row_utt,
                                                            int32_t* __restrict__ total_rows) {
  __shared__ int total_s;
  if (threadIdx.x == 0) {                       // a few hundred utterances at most: a serial prefix sum is fine
    int off = 0;
    for (int u = 0; u < n_utts; ++u) {
      UttPlan tp = own[u];
      tp.row_off = off;
      compact[u] = tp;
      off += (tp.T + 15) & ~15;
    }
    total_s = off;
    *total_rows = off;
  }
  __syncthreads();
  const int total = total_s;
  for (int r = threadIdx.x; r < n_rows_max; r += 256) row_utt[r] = -1;
  __syncthreads();
  for (int u = 0; u < n_utts; ++u) {
    const int r0 = compact[u].row_off, n16 = (compact[u].T + 15) & ~15;
    for (int r = threadIdx.x; r < n16; r += 256)
      if (r0 + r < min(total, n_rows_max)) row_utt[r0 + r] = u;
  }
}

__global__ __launch_bounds__(256) void compact_rows_kernel(const float* __restrict__ src, const UttPlan* __restrict__ own,
                                                           const UttPlan* __restrict__ compact, int d, float* __restrict__ dst) {
  const int u = blockIdx.x, s0 = own[u].row_off, d0 = compact[u].row_off, n16 = (compact[u].T + 15) & ~15;
  for (int e = threadIdx.x; e < n16 * d; e += 256) dst[(size_t)d0 * d + e] = src[(size_t)s0 * d + e];
}

__global__ void gather_tokens_kernel(const int32_t* __restrict__ ids, const UttPlan* __restrict__ tplan, int32_t* __restrict__ token_ids,
                                     int max_tokens) {
  const int u = blockIdx.x;
  const int n = min(tplan[u].n_lfr, max_tokens);
  for (int i = threadIdx.x; i < n; i += blockDim.x) token_ids[(size_t)u * max_tokens + i] = ids[tplan[u].row_off + i];
}

}  // namespace

// ==================================================================================== launchers
void launch_fbank(const FbankArgs& a, int n_blocks, hipStream_t s) {
  ASR_REQUIRE(a.win == WIN && a.hop == HOP, "fbank: only win=400 hop=160 is built (got %d/%d)", a.win, a.hop);
  ASR_REQUIRE(a.n_bin_tiles * 16 <= FB_PLD - 1, "fbank: too many frequency bins");
  ASR_REQUIRE(a.n_kchunks * 16 == WIN, "fbank: k-chunks must cover the window");
  const size_t lds = (size_t)(FB_AUDIO_LDS + FB_FRAMES * FB_PLD) * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fbank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (a.dft_split) {
    FbankArgs b = a;
    if (const char* e = getenv("ASR_FBANK_DBG")) b.dbg = atoi(e);
    const size_t lds2 = (size_t)FB_A16 * 2 * 2 + (size_t)FB_FRAMES * FB_PLD * sizeof(float);
    static PerDeviceOnce attr2_once;
    if (attr2_once.first()) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fbank_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    }
    hipLaunchKernelGGL(fbank_split_kernel, dim3(n_blocks), dim3(64 * FB_WAVES), lds2, s, b);
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(fbank_kernel, dim3(n_blocks), dim3(256), lds, s, a);
  HIP_CHECK(hipGetLastError());
}

size_t fbank_split_table_bytes(int n_bin_tiles, int win) {
  ASR_REQUIRE(win == WIN, "fbank: only win=400 is built (got %d)", win);
  return (size_t)n_bin_tiles * 2 * FB_K32 * 3 * 64 * 16;
}
void launch_fbank_split_table(const float* dft_packed, int n_bin_tiles, int n_kchunks16, void* out, hipStream_t s) {
  const int total = n_bin_tiles * 2 * FB_K32 * 64;
  hipLaunchKernelGGL(fbank_split_table_kernel, dim3((total + 255) / 256), dim3(256), 0, s, dft_packed, n_bin_tiles, n_kchunks16, reinterpret_cast<uint4*>(out));
  HIP_CHECK(hipGetLastError());
}

void launch_lfr_cmvn(const LfrArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(lfr_cmvn_kernel, dim3(a.n_rows), dim3(256), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

template <typename OutT>
void launch_layernorm(const float* x, int ld_x, int rows, int D, const float* gamma, const float* beta, float eps,
                      OutT* out, int ld_out, int fill_to, hipStream_t s, const int32_t* rows_dev) {
  ASR_REQUIRE(D % 4 == 0 && D <= LN_MAXV * 256 && ld_x % 4 == 0 && ld_out % 4 == 0, "layernorm: D=%d ld=%d unsupported", D, ld_x);
  hipLaunchKernelGGL(layernorm_kernel<OutT>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ld_x, rows, D, gamma, beta, eps, out,
                     ld_out, fill_to, rows_dev);
  HIP_CHECK(hipGetLastError());
}
template void launch_layernorm<float>(const float*, int, int, int, const float*, const float*, float, float*, int, int, hipStream_t, const int32_t*);
template void launch_layernorm<bf16_t>(const float*, int, int, int, const float*, const float*, float, bf16_t*, int, int, hipStream_t, const int32_t*);

void launch_layernorm_fp8(const float* x, int ld_x, int rows, int D, float eps, float inv_scale, unsigned char* out, int ld_out, hipStream_t s) {
  ASR_REQUIRE(D % 256 == 0 && D <= 2048 && ld_x % 4 == 0 && ld_out % 4 == 0, "layernorm_fp8: D=%d unsupported", D);
  hipLaunchKernelGGL(layernorm_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ld_x, rows, D, eps, inv_scale, out, ld_out);
  HIP_CHECK(hipGetLastError());
}

template <int HD, int CHUNK, int QT>
static void launch_attn_inst(const AttnArgs& a, hipStream_t s) {
  constexpr int lds = CHUNK * HD * 2 * 2;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_kernel<HD, CHUNK, QT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  AttnArgs b = a;
  if (b.ld_q == 0) b.ld_q = b.ld_qk;
  hipLaunchKernelGGL((attn_bf16_kernel<HD, CHUNK, QT>), dim3(a.n_qblocks, a.n_heads), dim3(64 * a.n_waves), lds, s, b, a.ld_vt);
  HIP_CHECK(hipGetLastError());
}

void attention_geometry(int max_T, int head_dim, int* qt, int* nw) {
  // smallest number of query tiles per wave that lets one workgroup (<= 8 waves) cover the longest utterance -- capped where the
  // accumulator set still lets two workgroups share a CU: with 64-wide heads 3 or 4 tiles per wave cost 190-256 VGPRs (spills at 4), one
  // workgroup per CU and every chunk's DMA round trip exposed. Whisper-large-v3 encoder attention per step, 4 -> 2 tiles per wave:
  // 42.5 -> 23.2 ms (B = 32 x 30 s), 4.75 -> 2.80 ms (B = 32 x 8 s); a second chunk buffer instead bought nothing (still one workgroup).
  const int qt_max = 2;                       // (128-wide heads: 3 tiles per wave spill 15 registers)
  const int n_tiles = (max_T + 15) / 16;
  int q = 1;
  while (q < qt_max && (n_tiles + q - 1) / q > 8) ++q;
  if (const char* e = getenv("ASR_ATTN_QT")) q = std::max(1, std::min(qt_max, atoi(e)));
  int w = std::min(8, (n_tiles + q - 1) / q);
  if (const char* e = getenv("ASR_ATTN_NW")) w = std::max(1, std::min(8, atoi(e)));
  *qt = q;
  *nw = w;
}

void launch_attention_bf16_hd128(const AttnArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.qt >= 1 && a.qt <= 2 && a.n_waves >= 1 && a.n_waves <= 8, "attention: bad geometry qt=%d nw=%d", a.qt, a.n_waves);
  const bool big = a.max_T <= 256;             // whole utterance in one 256-key chunk (128 KiB LDS)
  if (a.qt == 1) { big ? launch_attn_inst<128, 256, 1>(a, s) : launch_attn_inst<128, 128, 1>(a, s); }
  else { big ? launch_attn_inst<128, 256, 2>(a, s) : launch_attn_inst<128, 128, 2>(a, s); }
}
void launch_attention_bf16_hd64(const AttnArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.qt >= 1 && a.qt <= 2 && a.n_waves >= 1 && a.n_waves <= 8, "attention: bad geometry qt=%d nw=%d", a.qt, a.n_waves);
  if (a.qt == 1) launch_attn_inst<64, 256, 1>(a, s); else launch_attn_inst<64, 256, 2>(a, s);
}

void launch_attention_f32(const AttnArgs& a, int head_dim, hipStream_t s) {
  ASR_REQUIRE(head_dim <= 128 && head_dim % 4 == 0, "attention_f32: head_dim %d unsupported", head_dim);
  AttnArgs b = a;
  if (b.ld_q == 0) b.ld_q = b.ld_qk;
  const int wgs = a.n_qblocks * a.n_heads;
  const int z = wgs >= 1024 ? 1 : wgs >= 256 ? 4 : 16;           // few query blocks (single utterances): one row per wave
  hipLaunchKernelGGL(attn_f32_kernel, dim3(a.n_qblocks, a.n_heads, z), dim3(256), 0, s, b, head_dim);
  HIP_CHECK(hipGetLastError());
}

template <typename InT>
void launch_fsmn(const InT* vt, int ld, const float* w, const float* b, int C, int ktaps, const UttPlan* plan,
                 const int32_t* row_utt, int n_rows_pad, float* out, int ld_out, hipStream_t s) {
  ASR_REQUIRE(ktaps == 11, "fsmn: built for 11 taps (got %d)", ktaps);
  ASR_REQUIRE(ld % 8 == 0 && n_rows_pad % 64 == 0 && n_rows_pad <= ld && C % 64 == 0 && ld_out % 4 == 0, "fsmn: bad geometry");
  hipLaunchKernelGGL((fsmn_kernel<InT, 11>), dim3(n_rows_pad / 64, C / 64), dim3(256), 0, s, vt, ld, w, b, plan, row_utt, C, out, ld_out);
  HIP_CHECK(hipGetLastError());
}
template void launch_fsmn<float>(const float*, int, const float*, const float*, int, int, const UttPlan*, const int32_t*, int, float*, int, hipStream_t);
template void launch_fsmn<bf16_t>(const bf16_t*, int, const float*, const float*, int, int, const UttPlan*, const int32_t*, int, float*, int, hipStream_t);

void launch_ctc_collapse(const int32_t* frame_ids, const UttPlan* plan, int n_utts, int blank_id, int32_t* token_ids,
                         int max_tokens, int32_t* num_id, hipStream_t s) {
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(n_utts), dim3(256), 0, s, frame_ids, plan, blank_id, token_ids, max_tokens, num_id);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_whisper_mel_finish(const float* mel, const float* blk_max, const UttPlan* plan, const int32_t* grow_utt,
                               int n_gapped_rows, int n_mels, T* out, hipStream_t s) {
  hipLaunchKernelGGL(whisper_mel_finish_kernel<T>, dim3(n_gapped_rows), dim3(128), 0, s, mel, blk_max, plan, grow_utt, n_mels, out);
  HIP_CHECK(hipGetLastError());
}
template void launch_whisper_mel_finish<float>(const float*, const float*, const UttPlan*, const int32_t*, int, int, float*, hipStream_t);
template void launch_whisper_mel_finish<bf16_t>(const float*, const float*, const UttPlan*, const int32_t*, int, int, bf16_t*, hipStream_t);

template <typename T>
void launch_zero_gap_rows(T* buf, int ld, int n_cols, const UttPlan* plan, const int32_t* grow_utt, int n_gapped_rows, hipStream_t s) {
  hipLaunchKernelGGL(zero_gap_rows_kernel<T>, dim3(n_gapped_rows), dim3(256), 0, s, buf, ld, n_cols, plan, grow_utt);
  HIP_CHECK(hipGetLastError());
}
template void launch_zero_gap_rows<float>(float*, int, int, const UttPlan*, const int32_t*, int, hipStream_t);
template void launch_zero_gap_rows<bf16_t>(bf16_t*, int, int, const UttPlan*, const int32_t*, int, hipStream_t);

void launch_compact_rows(const float* src, const UttPlan* from, const UttPlan* to, const int32_t* row_utt, int rows, int d, float* dst, hipStream_t s) {
  ASR_REQUIRE(d % 4 == 0, "compact_rows: row length must be a multiple of 4");
  hipLaunchKernelGGL(compact_rows_kernel, dim3(rows), dim3(256), 0, s, src, from, to, row_utt, d, dst);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_embed_pos(const int32_t* ids, int rows, int n, int hist, const int32_t* hist_dev, const T* embed, const float* pos, int d,
                      float* x, hipStream_t s) {
  hipLaunchKernelGGL(embed_pos_kernel<T>, dim3(rows), dim3(256), 0, s, ids, n, hist, hist_dev, embed, pos, d, x);
  HIP_CHECK(hipGetLastError());
}
template void launch_embed_pos<float>(const int32_t*, int, int, int, const int32_t*, const float*, const float*, int, float*, hipStream_t);
template void launch_embed_pos<bf16_t>(const int32_t*, int, int, int, const int32_t*, const bf16_t*, const float*, int, float*, hipStream_t);

namespace { __global__ void add_scalar_kernel(int32_t* p, int v) { *p += v; } }
void launch_add_scalar(int32_t* p, int v, hipStream_t s) {
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, s, p, v);
  HIP_CHECK(hipGetLastError());
}

namespace {
// ---- single-token self-attention, one WAVE per (sequence, head): the launch above spends four workgroup barriers and three dependent
// global-load rounds on <= 40 cached positions (15 us per launch at 64 sequences: 22 % of a Whisper decode step). Here every load a wave needs
// (its query, the new K / V row, up to 64 cached K and V rows) is requested at once, 8 lanes per 128-byte row (16 B each, coalesced), the dot
// products are reduced inside the 8-lane groups by DPP, the soft-max runs online over blocks of 64 keys with wave shuffles, and nothing goes through LDS.
__global__ __launch_bounds__(256) void decode_self_attn_wave_kernel(const DecAttnArgs a) {
  const int b = blockIdx.x + a.b0, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y * 4 + wave;
  if (h >= a.n_heads) return;
  const int hist = a.hist_dev ? *a.hist_dev : a.hist;          // keys 0 .. hist - 1 are cached, key `hist` is the new row
  const int sub = lane & 7, grp = lane >> 3;
  // cache row s: contiguous extent, or through the block table (a wave instruction covers 8 consecutive rows = one 16-position page: the lookup is wave-uniform)
  const int32_t* pt = a.page_table ? a.page_table + (size_t)b * a.pages_per_seq : nullptr;
  bf16_t* Kc = reinterpret_cast<bf16_t*>(a.k_base) + (pt ? (size_t)h * 1024 : (size_t)b * a.stride_b + (size_t)h * a.stride_h);
  bf16_t* Vc = reinterpret_cast<bf16_t*>(a.v_base) + (pt ? (size_t)h * 1024 : (size_t)b * a.stride_b + (size_t)h * a.stride_h);
  auto crow = [&](int s) -> size_t { return pt ? (size_t)pt[s >> 4] * (size_t)a.page_stride + (size_t)(s & 15) * 64 : (size_t)s * 64; };
  const bf16_t* NEW = reinterpret_cast<const bf16_t*>(a.kv_new) + (size_t)b * a.ld_new + h * 64 + sub * 8;
  Raw8<bf16_t> q8, nk, nv;
  q8.load(reinterpret_cast<const bf16_t*>(a.q) + (size_t)b * a.ld_q + a.q_col0 + h * 64 + sub * 8);
  nk.load(NEW + a.k_col0);
  nv.load(NEW + a.v_col0);
  auto dot8 = [&](const float (&x)[8], const float (&y)[8]) {
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(x[e], y[e], acc);
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xf, 0xf, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xf, 0xf, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x141, 0xf, 0xf, true));
    return acc;                                              // the row's dot product, in all 8 lanes of the group
  };
  float m = -INFINITY, l = 0.0f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
  Raw8<bf16_t> kr[8], vr[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    kr[u].v = make_uint4(0, 0, 0, 0); vr[u].v = make_uint4(0, 0, 0, 0);      // slots past the history stay finite: 0 x garbage could be NaN
    const int s0 = u * 8 + grp;
    if (s0 < hist) { const size_t ro = crow(s0); kr[u].load(Kc + ro + sub * 8); vr[u].load(Vc + ro + sub * 8); }
  }
  if (grp == 0) {                                            // append the new row (read back only by later steps)
    const size_t ro = crow(hist);
    *reinterpret_cast<uint4*>(Kc + ro + sub * 8) = nk.v;
    *reinterpret_cast<uint4*>(Vc + ro + sub * 8) = nv.v;
  }
  float qf[8];
  q8.get(qf);
  for (int blk = 0; blk < hist; blk += 64) {
    float sc[8], vf[8][8];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float kf[8];
      kr[u].get(kf);
      vr[u].get(vf[u]);
      const float d = dot8(qf, kf);
      sc[u] = blk + u * 8 + grp < hist ? d : -INFINITY;
      mx = fmaxf(mx, sc[u]);
    }
    if (blk + 64 < hist) {                                   // the next block's rows (more than 64 positions of history)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s1 = blk + 64 + u * 8 + grp;
        kr[u].v = make_uint4(0, 0, 0, 0); vr[u].v = make_uint4(0, 0, 0, 0);
        if (s1 < hist) { const size_t ro = crow(s1); kr[u].load(Kc + ro + sub * 8); vr[u].load(Vc + ro + sub * 8); }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx), alpha = __expf(m - m_new);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float p = __expf(sc[u] - m_new);                 // exp(-inf) = 0 for the slots past the history
      l += p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vf[u][e], acc[e]);
    }
    m = m_new;
  }
  // partial sums of the 8 key groups -> totals (every lane of a column group ends with the same values)
  l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 8, 64); acc[e] += __shfl_xor(acc[e], 16, 64); acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  float nkf[8], nvf[8];
  nk.get(nkf);
  nv.get(nvf);
  const float s_new = dot8(qf, nkf);
  const float m_fin = fmaxf(m, s_new), alpha = __expf(m - m_fin), p_new = __expf(s_new - m_fin);
  const float inv = 1.0f / (l * alpha + p_new);
  if (grp == 0) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (acc[e] * alpha + p_new * nvf[e]) * inv;
    uint4 w;
    w.x = pack_bf16x2(o[0], o[1]); w.y = pack_bf16x2(o[2], o[3]); w.z = pack_bf16x2(o[4], o[5]); w.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)b * a.ld_out + h * 64 + sub * 8) = w;
  }
}

}  // namespace

template <typename T>
void launch_decode_attention(const DecAttnArgs& a, int batch, hipStream_t s) {
  ASR_REQUIRE(a.n >= 1 && a.n <= DA_MAXN, "decode attention: %d new positions per call (max %d)", a.n, DA_MAXN);
  ASR_REQUIRE(a.plan || a.hist_dev || a.hist + a.n <= DA_MAXKEYS, "decode attention: %d keys exceed %d", a.hist + a.n, DA_MAXKEYS);
  if constexpr (std::is_same<T, bf16_t>::value) {
    const bool wave_on = gemm_env_decode_attn_wave();
    if (wave_on && a.n == 1 && !a.plan && a.kv_new && !a.k_scale && !a.v_scale && (a.ld_q % 8) == 0 && (a.ld_new % 8) == 0 && (a.q_col0 % 8) == 0 &&
        (a.k_col0 % 8) == 0 && (a.v_col0 % 8) == 0 && (a.ld_out % 8) == 0) {                 // single-token self-attention: one wave per (sequence, head)
      hipLaunchKernelGGL(decode_self_attn_wave_kernel, dim3(batch, (a.n_heads + 3) / 4), dim3(256), 0, s, a);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  // single-token cross-attention: one pass with a running soft-max -- except bf16 slabs of long extents (30 s windows: 1500 keys), where the two-pass kernel's single
  // 8-row stream measured 1.7 % faster (195.5 vs 199 ms per 32 x 30 s batch; 8 s windows: 33.2 -> 30.0 ms at 64 sequences; FP8 slabs 123.4 -> 115.5 at 30 s)
  const bool long_bf16 = !(a.k_scale || a.v_scale) && a.max_keys > 1000;
  if (gemm_env_decode_attn_online() && !long_bf16 && a.n == 1 && a.plan && !a.page_table && !a.kv_new && !a.causal && (a.ld_q % 8) == 0) {
    if (a.k_scale || a.v_scale) {
      if constexpr (std::is_same<T, bf16_t>::value) {
        ASR_REQUIRE(a.k_scale && a.v_scale, "decode attention: the FP8 cache path needs both scale arrays");
        hipLaunchKernelGGL((decode_cross_attn_kernel<T, true>), dim3(batch, a.n_heads), dim3(256), 0, s, a);
      } else ASR_REQUIRE(false, "decode attention: FP8 K / V need a bf16 session");
    } else hipLaunchKernelGGL((decode_cross_attn_kernel<T, false>), dim3(batch, a.n_heads), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
    return;
  }
  DecAttnArgs b = a;
  b.sc_ld = (std::min(a.max_keys > 0 ? a.max_keys : DA_MAXKEYS, DA_MAXKEYS) + 63) & ~63;
  const size_t lds = (size_t)a.n * b.sc_ld * 4;
  if (a.k_scale || a.v_scale) {
    if constexpr (std::is_same<T, bf16_t>::value) {
      ASR_REQUIRE(a.k_scale && a.v_scale && !a.kv_new && a.plan, "decode attention: the FP8 cache path is the cross-attention of bf16 sessions");
      if (a.n == 1) hipLaunchKernelGGL((decode_attn_kernel<T, 1, true>), dim3(batch, a.n_heads), dim3(256), lds, s, b);
      else hipLaunchKernelGGL((decode_attn_kernel<T, DA_MAXN, true>), dim3(batch, a.n_heads), dim3(256), lds, s, b);
    } else ASR_REQUIRE(false, "decode attention: FP8 K / V need a bf16 session");
  }
  else if (a.n == 1) hipLaunchKernelGGL((decode_attn_kernel<T, 1>), dim3(batch, a.n_heads), dim3(256), lds, s, b);
  else hipLaunchKernelGGL((decode_attn_kernel<T, DA_MAXN>), dim3(batch, a.n_heads), dim3(256), lds, s, b);
  HIP_CHECK(hipGetLastError());
}

// ---- FP8 (OCP e4m3) quantisers of precision mode ASR_PRECISION_FP8W. Scales are powers of two (amax lands in (224, 448]): dequantised
//      values are then exact in bf16, so a bf16 run over the dequantised copies reproduces the FP8 kernels bit for bit (the tests use that).
namespace {
__device__ __forceinline__ float pow2_scale(float amax) {          // smallest power of two s with amax / s <= 448
  if (!(amax > 0.0f)) return 1.0f;
  int e;
  const float m = frexpf(amax * (1.0f / 448.0f), &e);                // amax / 448 = m 2^e, m in [0.5, 1)
  return ldexpf(1.0f, m == 0.5f ? e - 1 : e);
}
__device__ __forceinline__ float block_amax(float v, float* sh) {    // 256 threads
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// 8 consecutive bf16 -> 8 e4m3 bytes (and, optionally, their exact bf16 dequantisation)
__device__ __forceinline__ uint2 quant8(const uint4 raw, float inv_s, float s, uint4* dq) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float f[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(w[e] << 16) * inv_s; f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) * inv_s; }
  uint2 q;
  unsigned p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], p, false); p = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p, true); q.x = p;
  p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], p, false); p = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], p, true); q.y = p;
  if (dq) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(q.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(q.x, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(q.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(q.y, true);
    dq->x = pack_bf16x2(a[0] * s, a[1] * s); dq->y = pack_bf16x2(b[0] * s, b[1] * s); dq->z = pack_bf16x2(c[0] * s, c[1] * s); dq->w = pack_bf16x2(d[0] * s, d[1] * s);
  }
  return q;
}
__device__ __forceinline__ float amax8(const uint4 raw) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float m = 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(__uint_as_float(w[e] << 16)), fabsf(__uint_as_float(w[e] & 0xffff0000u))));
  return m;
}
// one workgroup per weight row: W [N][K] bf16 (pitch ld) -> W8 [N][K] bytes + scale[N] (+ Wdq [N][K] bf16)
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ W, int ld, int K, unsigned char* __restrict__ W8, float* __restrict__ scale,
                                                                bf16_t* __restrict__ Wdq) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  const bf16_t* row = W + (size_t)n * ld;
  float m = 0.0f;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) m = fmaxf(m, amax8(*reinterpret_cast<const uint4*>(row + k)));
  const float s = pow2_scale(block_amax(m, sh)), inv_s = 1.0f / s;
  if (threadIdx.x == 0) scale[n] = s;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    uint4 dq;
    const uint2 q = quant8(*reinterpret_cast<const uint4*>(row + k), inv_s, s, Wdq ? &dq : nullptr);
    *reinterpret_cast<uint2*>(W8 + (size_t)n * K + k) = q;
    if (Wdq) *reinterpret_cast<uint4*>(Wdq + (size_t)n * K + k) = dq;
  }
}
// ---- MXFP4 (OCP Microscaling v1.0: e2m1 elements, one e8m0 scale per 32 consecutive k): W [N][K] bf16 -> W4 [N][K / 2] bytes (element 2 i in the low nibble of
// byte i), S [N][K / 32] e8m0 bytes (+ Wdq [N][K] bf16, the exact dequantisation: 2 significant bits times a power of two). Shared exponent of a block =
// floor(log2(amax)) - 2 (e2m1's largest exponent), so amax / scale lies in [4, 8) and everything above 6 saturates at 6 (the specification's clamp); elements are
// rounded to nearest, ties to the even code. A block of zeros takes scale 1. The reference's counterpart: its MatMulNBits Q4 graphs (Optimize_ONNX_Common.py:55-60,
// README.md:70: Qwen3-ASR is published as q4f32).
__device__ __forceinline__ unsigned e2m1_code(float v) {        // v >= 0, already divided by the block scale
  return (unsigned)(v > 0.25f) + (unsigned)(v >= 0.75f) + (unsigned)(v > 1.25f) + (unsigned)(v >= 1.75f) + (unsigned)(v > 2.5f) + (unsigned)(v >= 3.5f) + (unsigned)(v > 5.0f);
}
__global__ __launch_bounds__(256) void quantize_rows_mxfp4_kernel(const bf16_t* __restrict__ W, int ld, int K, unsigned char* __restrict__ W4, unsigned char* __restrict__ S,
                                                                  bf16_t* __restrict__ Wdq) {
  const int n = blockIdx.x;
  const bf16_t* row = W + (size_t)n * ld;
  for (int blk = threadIdx.x; blk * 32 < K; blk += 256) {
    uint4 raw[4];
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { raw[i] = *reinterpret_cast<const uint4*>(row + blk * 32 + i * 8); m = fmaxf(m, amax8(raw[i])); }
    int eb = (int)((__float_as_uint(m) >> 23) & 0xffu) - 2;            // biased exponent of the scale
    if (m == 0.0f) eb = 127;
    eb = max(eb, 1);                                                   // (a block whose amax is below 2^-124 keeps the smallest normal scale)
    const float inv_s = __uint_as_float((unsigned)(254 - eb) << 23);   // 2^-(eb - 127)
    const float sc = __uint_as_float((unsigned)eb << 23);
    S[(size_t)n * (K >> 5) + blk] = (unsigned char)eb;
    uint4 packed;
    unsigned pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned wd[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
      unsigned nib = 0, dqw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(wd[e] << 16), hi = __uint_as_float(wd[e] & 0xffff0000u);
        const unsigned cl = e2m1_code(fabsf(lo) * inv_s) | ((wd[e] >> 12) & 8u), ch = e2m1_code(fabsf(hi) * inv_s) | ((wd[e] >> 28) & 8u);
        nib |= (cl | (ch << 4)) << (8 * e);
        // dequantised value = level * scale, exact in f32 and in bf16 (the widening instruction of the GEMM computes the same product)
        const float lv[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
        const float dl = lv[cl & 7u] * sc, dh = lv[ch & 7u] * sc;
        dqw[e] = ((__float_as_uint(dl) >> 16) | ((cl & 8u) << 12)) | ((__float_as_uint(dh) & 0xffff0000u) | ((ch & 8u) << 28));
      }
      pw[i] = nib;
      if (Wdq) *reinterpret_cast<uint4*>(Wdq + (size_t)n * K + blk * 32 + i * 8) = make_uint4(dqw[0], dqw[1], dqw[2], dqw[3]);
    }
    packed = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    *reinterpret_cast<uint4*>(W4 + (size_t)n * (K >> 1) + blk * 16) = packed;
  }
}
// one workgroup per (sequence, slab): the T rows x 64 dims of sequence b inside slab (kv, layer, head) -> bytes + scale[slab][b];
// dq_in_place: additionally overwrite the bf16 rows with their dequantisation (the bf16 reference of the FP8 path)
__global__ __launch_bounds__(256) void quantize_crosskv_fp8_kernel(bf16_t* __restrict__ slabs, size_t slab_elems, const UttPlan* __restrict__ plan,
                                                                   unsigned char* __restrict__ out8, float* __restrict__ scale, int dq_in_place) {
  __shared__ float sh[4];
  const int b = blockIdx.x, slab = blockIdx.y, B = gridDim.x;
  const UttPlan p = plan[b];
  bf16_t* src = slabs + (size_t)slab * slab_elems + (size_t)p.row_off * 64;
  unsigned char* dst = out8 + (size_t)slab * slab_elems + (size_t)p.row_off * 64;
  const int n8 = p.n_lfr * 8;                                        // 16-byte groups of the region
  float m = 0.0f;
  for (int i = threadIdx.x; i < n8; i += 256) m = fmaxf(m, amax8(reinterpret_cast<const uint4*>(src)[i]));
  const float s = pow2_scale(block_amax(m, sh)), inv_s = 1.0f / s;
  if (threadIdx.x == 0) scale[(size_t)slab * B + b] = s;
  for (int i = threadIdx.x; i < n8; i += 256) {
    uint4 dq;
    const uint2 q = quant8(reinterpret_cast<const uint4*>(src)[i], inv_s, s, dq_in_place ? &dq : nullptr);
    reinterpret_cast<uint2*>(dst)[i] = q;
    if (dq_in_place) reinterpret_cast<uint4*>(src)[i] = dq;
  }
}
}  // namespace

void launch_quantize_rows_fp8(const bf16_t* W, int ld, int N, int K, unsigned char* W8, float* scale, bf16_t* Wdq, hipStream_t s) {
  ASR_REQUIRE(K % 8 == 0 && ld % 8 == 0, "quantize_rows_fp8: K and the row pitch must be multiples of 8");
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(N), dim3(256), 0, s, W, ld, K, W8, scale, Wdq);
  HIP_CHECK(hipGetLastError());
}
void launch_quantize_rows_mxfp4(const bf16_t* W, int ld, int N, int K, unsigned char* W4, unsigned char* S, bf16_t* Wdq, hipStream_t s) {
  ASR_REQUIRE(K % 32 == 0 && ld % 8 == 0, "quantize_rows_mxfp4: K must be a multiple of the 32-element block (and the row pitch of 8)");
  hipLaunchKernelGGL(quantize_rows_mxfp4_kernel, dim3(N), dim3(256), 0, s, W, ld, K, W4, S, Wdq);
  HIP_CHECK(hipGetLastError());
}
void launch_quantize_crosskv_fp8(bf16_t* slabs, size_t slab_elems, int n_slabs, const UttPlan* plan, int batch, unsigned char* out8, float* scale,
                                 int dq_in_place, hipStream_t s) {
  hipLaunchKernelGGL(quantize_crosskv_fp8_kernel, dim3(batch, n_slabs), dim3(256), 0, s, slabs, slab_elems, plan, out8, scale, dq_in_place);
  HIP_CHECK(hipGetLastError());
}
template void launch_decode_attention<float>(const DecAttnArgs&, int, hipStream_t);
template void launch_decode_attention<bf16_t>(const DecAttnArgs&, int, hipStream_t);

void launch_argmax_rows(const float* logits, int ld, int rows, int n_valid, const float* extra, int32_t* ids, hipStream_t s) {
  ASR_REQUIRE(ld % 4 == 0 && n_valid <= ld, "argmax_rows: rows must be padded to a multiple of 4 columns");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(1024), 0, s, logits, ld, n_valid, extra, ids);
  HIP_CHECK(hipGetLastError());
}

void launch_no_speech_prob(const float* logits, int ld, int rows, int n_valid, const float* penalty, int no_speech_id, float* prob, hipStream_t s) {
  ASR_REQUIRE(no_speech_id >= 0 && no_speech_id < n_valid, "no_speech_prob: id %d outside the vocabulary", no_speech_id);
  hipLaunchKernelGGL(no_speech_prob_kernel, dim3(rows), dim3(1024), 0, s, logits, ld, n_valid, penalty, no_speech_id, prob);
  HIP_CHECK(hipGetLastError());
}

void launch_apply_penalty(float* logits, int ld, int rows, const int32_t* save_ids, int ld_save, const int32_t* n_saved,
                          int range, float value, hipStream_t s, int partial) {
  ASR_REQUIRE(range >= 1 && range <= 64 && range <= ld_save, "apply_penalty: range %d (1..64)", range);
  hipLaunchKernelGGL(apply_penalty_kernel, dim3(rows), dim3(64), 0, s, logits, ld, save_ids, ld_save, n_saved, range, value, partial);
  HIP_CHECK(hipGetLastError());
}

void launch_sample_topk_topp(const SampleArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.top_k >= 1 && a.top_k <= 64 && a.top_k <= a.n_valid, "sampling: top_k %d (1..64)", a.top_k);
  ASR_REQUIRE(a.temperature > 0.0f && a.repetition_penalty > 0.0f && a.ld_save <= 1024, "sampling: bad temperature / penalty / history capacity");
  hipLaunchKernelGGL(sample_topk_topp_kernel, dim3(a.rows), dim3(256), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

void launch_append_ids(const int32_t* next, int rows, int32_t* save_ids, int ld_save, const int32_t* n_saved, hipStream_t s) {
  hipLaunchKernelGGL(append_ids_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, next, rows, save_ids, ld_save, n_saved);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_shift3(const T* x, int d, const UttPlan* plan, const int32_t* row_utt, int n_rows, T* out, hipStream_t s) {
  hipLaunchKernelGGL(shift3_kernel<T>, dim3(n_rows), dim3(256), 0, s, x, d, plan, row_utt, out);
  HIP_CHECK(hipGetLastError());
}
template void launch_shift3<float>(const float*, int, const UttPlan*, const int32_t*, int, float*, hipStream_t);
template void launch_shift3<bf16_t>(const bf16_t*, int, const UttPlan*, const int32_t*, int, bf16_t*, hipStream_t);

template <typename T>
void launch_alpha(const T* h, int d, const float* w, const float* b, int rows, float* alpha, hipStream_t s) {
  hipLaunchKernelGGL(alpha_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, s, h, d, w, b, rows, alpha);
  HIP_CHECK(hipGetLastError());
}
template void launch_alpha<float>(const float*, int, const float*, const float*, int, float*, hipStream_t);
template void launch_alpha<bf16_t>(const bf16_t*, int, const float*, const float*, int, float*, hipStream_t);

void launch_cif_scan(const float* alpha, const float* enc_out, int d, const UttPlan* plan, int n_utts, float tail_threshold,
                     float* acoustic, UttPlan* token_plan, int32_t* num_id, hipStream_t s) {
  hipLaunchKernelGGL(cif_scan_kernel, dim3(n_utts), dim3(std::min(1024, (d + 63) / 64 * 64)), 0, s, alpha, enc_out, d, plan, tail_threshold,
                     acoustic, token_plan, num_id);
  HIP_CHECK(hipGetLastError());
}

void launch_fsmn_rows(const float* x, const float* res, const float* w, int d, int ktaps, const UttPlan* token_plan,
                      const int32_t* row_utt, int n_rows, float* out, hipStream_t s, const int32_t* rows_dev) {
  hipLaunchKernelGGL(fsmn_rows_kernel, dim3(n_rows), dim3(256), 0, s, x, res, w, d, ktaps, token_plan, row_utt, out, rows_dev);
  HIP_CHECK(hipGetLastError());
}

void launch_token_compact(const UttPlan* own_plan, int n_utts, int n_rows_max, UttPlan* compact_plan, int32_t* row_utt, int32_t* total_rows,
                          hipStream_t s) {
  hipLaunchKernelGGL(token_compact_kernel, dim3(1), dim3(256), 0, s, own_plan, n_utts, n_rows_max, compact_plan, row_utt, total_rows);
  HIP_CHECK(hipGetLastError());
}

void launch_compact_rows(const float* src, const UttPlan* own_plan, const UttPlan* compact_plan, int n_utts, int d, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(compact_rows_kernel, dim3(n_utts), dim3(256), 0, s, src, own_plan, compact_plan, d, dst);
  HIP_CHECK(hipGetLastError());
}

void launch_gather_tokens(const int32_t* ids, const UttPlan* token_plan, int n_utts, int32_t* token_ids, int max_tokens, hipStream_t s) {
  hipLaunchKernelGGL(gather_tokens_kernel, dim3(n_utts), dim3(256), 0, s, ids, token_plan, token_ids, max_tokens);
  HIP_CHECK(hipGetLastError());
}


// ==================================================================================== streaming Paraformer
namespace {

__global__ __launch_bounds__(256) void stream_lfr_kernel(const StreamLfrArgs a) {
  const int m = blockIdx.x, i = m >> 4, t = m & 15;
  float* o = a.out + (size_t)m * a.ld;
  const UttPlan up = a.plan[i];
  const int sid = up.lang, left = (a.lfr_m - 1) / 2;
  for (int c = threadIdx.x; c < a.ld; c += 256) {
    float v = 0.0f;
    if (c < a.feat && t < a.n_prev + a.n_new) {
      if (t < a.n_prev) {
        v = a.prev[((size_t)sid * a.n_prev + t) * a.ld + c];
      } else {
        const int j = t - a.n_prev;
        int f = j * a.lfr_n + c / a.n_mels - left;
        f = min(max(f, 0), a.n_frames - 1);
        const int p = min(a.start[sid] + j, a.pos_rows - 1);
        v = a.mel[(size_t)(up.frame_off + f) * a.n_mels + c % a.n_mels] * a.cmvn_vars[c] + a.pos_bias[(size_t)p * a.feat + c];
      }
    }
    o[c] = v;
  }
}

__global__ __launch_bounds__(256) void stream_carry_kernel(const float* __restrict__ x, int ld, const UttPlan* __restrict__ plan, int n_prev,
                                                           int n_new, float* __restrict__ prev, int32_t* __restrict__ start) {
  const int i = blockIdx.x, sid = plan[i].lang;
  for (int e = threadIdx.x; e < n_prev * ld; e += 256) {
    const int t = e / ld, c = e - t * ld;
    prev[((size_t)sid * n_prev + t) * ld + c] = x[(size_t)(plan[i].row_off + n_new + t) * ld + c];
  }
  if (threadIdx.x == 0) start[sid] += n_new;
}

constexpr int SA_MAXK = 64, SA_HD = 128;

// One workgroup per (stream, head), 256 threads, every query row of the chunk at once: K / V (history + current rows) and up to 16
// query rows are staged in LDS, the nq x nk scores are spread over the threads (one 128-long dot product each), a wave per
// query row does the soft-max, and the nq x 128 outputs are spread over the threads again.
constexpr int SA_MAXQ = 16;
constexpr int SA_F32_DYN = 144 * 1024;          // dynamic LDS of the f32 instance: its K / V images (67.6 KB) + the rest of the CU, so that nothing shares the CU's LDS with it

template <typename T>
__global__ __launch_bounds__(256) void stream_attn_kernel(const StreamAttnArgs a) {
  // The K / V images hold the session's element type (round 5): as f32 the kernel's static LDS was 78 KB -- more than 64 KB, yet small enough to share a CU with other
  // workgroups -- and next to ANOTHER session's kernels its results moved by a percent on some query rows, silently and only then (tools/probes/stream_determinism.py,
  // profiles/r05_stream_determinism.txt: identical q|k|v, history and lengths in, different context out; gone when the workgroup owns the CU's LDS, not gone with the LDS
  // cleared, an acquire at kernel start or one workgroup of this kernel per CU). bf16 sessions now use 46 KB; the f32 instance is launched with the CU's LDS to itself.
  // LDS reads come FOUR values at a time (round 5: at 256 streams this launch was 31 % of the per-launch chunk step -- 37.6 us for 1 024 workgroups, five scalar LDS
  // reads per four FMAs in the score phase, nine per eight in the context phase): a key row's four dims are one 8- / 16-byte read, a query row's and a score row's four values
  // one 16-byte broadcast read. Every FMA chain still runs in the original order, so the results are unchanged bit for bit. Row pitches: K / V 4 elements over 128 (rows stay
  // 8- / 16-byte aligned; 64 keys at one column pair hit every bank once per half / quarter wave), Q and P rows 16-byte aligned.
  // (round 6, ADVICE r05: the f32 instance's K / V images live in DYNAMIC LDS -- launch_stream_attn asks for SA_F32_DYN bytes, which with the static Q / P tiles is the
  //  whole CU -- so its static footprint is 13 KB like any other kernel's and growth of SA_MAXK / the pitches fails at compile time, not at launch)
  constexpr int SA_KP = SA_HD + 4, SA_QP = SA_HD + 4, SA_PP = SA_MAXK + 4;
  constexpr bool DYN = sizeof(T) == 4;
  static_assert(!DYN || 2 * SA_MAXK * SA_KP * (int)sizeof(T) <= SA_F32_DYN, "stream_attn<float>: K / V images exceed the dynamic LDS the launcher asks for");
  static_assert(SA_F32_DYN + SA_MAXQ * (SA_QP + SA_PP) * 4 + 2 * SA_KP * 4 <= 160 * 1024, "stream_attn<float>: static + dynamic LDS exceed a CU");
  __shared__ __attribute__((aligned(16))) T KsS[DYN ? 1 : SA_MAXK][SA_KP];
  __shared__ __attribute__((aligned(16))) T VsS[DYN ? 1 : SA_MAXK][SA_KP];
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_dyn[];
  T (*Ks)[SA_KP] = DYN ? reinterpret_cast<T (*)[SA_KP]>(sa_dyn) : KsS;
  T (*Vs)[SA_KP] = DYN ? reinterpret_cast<T (*)[SA_KP]>(sa_dyn + (size_t)SA_MAXK * SA_KP * sizeof(T)) : VsS;
  __shared__ __attribute__((aligned(16))) float Qs[SA_MAXQ][SA_QP];
  __shared__ __attribute__((aligned(16))) float Ps[SA_MAXQ][SA_PP];
  auto lds4 = [](const T* p) -> float4 {                    // four consecutive elements of a K / V row
    if constexpr (sizeof(T) == 2) {
      const uint2 w = *reinterpret_cast<const uint2*>(p);
      return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
    } else {
      return *reinterpret_cast<const float4*>(p);
    }
  };
  const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const UttPlan qp = a.q_plan[i];
  const int nq_all = qp.T, sid = qp.lang, row0 = qp.row_off;
  if (nq_all <= 0) return;
  const int len = a.cache_len[sid], nk = len + a.n_cur;
  const T* ck = reinterpret_cast<const T*>(a.cache_k) + ((size_t)sid * a.n_heads + h) * a.cap * SA_HD;
  const T* cv = reinterpret_cast<const T*>(a.cache_v) + ((size_t)sid * a.n_heads + h) * a.cap * SA_HD;
  const T* kk = reinterpret_cast<const T*>(a.k);
  const T* vv = reinterpret_cast<const T*>(a.v);
  const T* qq = reinterpret_cast<const T*>(a.q);
  T* out = reinterpret_cast<T*>(a.ctx);
  {                                                      // stage K and V: 16 threads per row (8 elements = 16 bytes each), all loads of a thread in flight
    constexpr int NIT = SA_MAXK * 16 / 256;              // 4 row segments per thread cover 64 rows
    Raw8<T> rk[NIT], rv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256, p = idx >> 4, c0 = (idx & 15) * 8;
      if (p < len) {
        rk[it].load(ck + (size_t)p * SA_HD + c0);
        rv[it].load(cv + (size_t)p * SA_HD + c0);
      } else if (p < nk) {
        const size_t r = (size_t)(row0 + p - len);
        rk[it].load(kk + r * a.ld_k + a.k_col0 + h * SA_HD + c0);
        rv[it].load(vv + r * a.ld_v + a.v_col0 + h * SA_HD + c0);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256, p = idx >> 4, c0 = (idx & 15) * 8;
      if (p < nk) {
        if constexpr (sizeof(T) == 2) {                   // the 16 bytes as they came, in two 8-byte stores (rows are 8-byte aligned)
          *reinterpret_cast<uint2*>(&Ks[p][c0]) = make_uint2(rk[it].v.x, rk[it].v.y); *reinterpret_cast<uint2*>(&Ks[p][c0 + 4]) = make_uint2(rk[it].v.z, rk[it].v.w);
          *reinterpret_cast<uint2*>(&Vs[p][c0]) = make_uint2(rv[it].v.x, rv[it].v.y); *reinterpret_cast<uint2*>(&Vs[p][c0 + 4]) = make_uint2(rv[it].v.z, rv[it].v.w);
        } else {
          *reinterpret_cast<float4*>(&Ks[p][c0]) = rk[it].a; *reinterpret_cast<float4*>(&Ks[p][c0 + 4]) = rk[it].b;
          *reinterpret_cast<float4*>(&Vs[p][c0]) = rv[it].a; *reinterpret_cast<float4*>(&Vs[p][c0 + 4]) = rv[it].b;
        }
      }
    }
  }
  for (int q0 = 0; q0 < nq_all; q0 += SA_MAXQ) {
    const int nq = min(SA_MAXQ, nq_all - q0);
    __syncthreads();                                     // K / V staged; the previous block's readers are done with Qs / Ps
    for (int e = tid; e < nq * SA_HD; e += 256) {
      const int q = e >> 7, c = e & 127;
      Qs[q][c] = Elem<T>::load(qq + (size_t)(row0 + q0 + q) * a.ld_q + a.q_col0 + h * SA_HD + c);
    }
    __syncthreads();
    {   // scores: thread = (key, query group); the key row is read once and serves four queries (independent FMA chains, each in the
        // original c order -- bit-identical to one chain per (query, key))
      const int k = tid & 63, qg = tid >> 6;
      if (k < nk) {
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int c = 0; c < SA_HD; c += 4) {
          const float4 kv = lds4(&Ks[k][c]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 q4 = *reinterpret_cast<const float4*>(&Qs[qg + 4 * j][c]);
            s4[j] = fmaf(q4.x, kv.x, s4[j]); s4[j] = fmaf(q4.y, kv.y, s4[j]); s4[j] = fmaf(q4.z, kv.z, s4[j]); s4[j] = fmaf(q4.w, kv.w, s4[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (qg + 4 * j < nq) Ps[qg + 4 * j][k] = s4[j];
      }
    }
    __syncthreads();
    for (int q = wave; q < nq; q += 4) {                 // nk <= 64: one lane per key
      const float sc = lane < nk ? Ps[q][lane] : -INFINITY;
      const float mx = wave_max(sc);
      const float ex = lane < nk ? expf(sc - mx) : 0.0f;
      const float sum = wave_sum(ex);
      if (lane < nk) Ps[q][lane] = ex / sum;
    }
    __syncthreads();
    {   // context: thread = (channel, query parity); one V read serves eight queries
      const int c = tid & 127, qh = tid >> 7;
      float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      int p = 0;
      for (; p + 4 <= nk; p += 4) {                        // four keys per trip: four V reads, one 16-byte (broadcast) read of each score row
        const float v0 = Elem<T>::load(&Vs[p][c]), v1 = Elem<T>::load(&Vs[p + 1][c]), v2 = Elem<T>::load(&Vs[p + 2][c]), v3 = Elem<T>::load(&Vs[p + 3][c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 p4 = *reinterpret_cast<const float4*>(&Ps[qh + 2 * j][p]);
          acc[j] = fmaf(p4.x, v0, acc[j]); acc[j] = fmaf(p4.y, v1, acc[j]); acc[j] = fmaf(p4.z, v2, acc[j]); acc[j] = fmaf(p4.w, v3, acc[j]);
        }
      }
      for (; p < nk; ++p) {
        const float v = Elem<T>::load(&Vs[p][c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(Ps[qh + 2 * j][p], v, acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = qh + 2 * j;
        if (q < nq) Elem<T>::store(out + (size_t)(row0 + q0 + q) * a.ld_ctx + h * SA_HD + c, acc[j]);
      }
    }
  }
  if (a.mem) {                                           // FSMN memory term of the 16-row slot (rows >= n_cur are zero), channels of this head
    const int pad = (a.ktaps - 1) / 2;
    const int c = tid & 127, hc = h * SA_HD + c;        // (256 % 128 == 0: a thread keeps its channel, its taps are loaded once)
    float wc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wc[j] = j < a.ktaps ? a.fsmn_w[hc * a.ktaps + j] : 0.0f;
    const float bc = a.fsmn_b[hc];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = (tid >> 7) + 2 * i;
      float acc = 0.0f;
      if (t < a.n_cur) {
        acc = bc;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int tt = t + j - pad;
          if (j < a.ktaps && tt >= 0 && tt < a.n_cur) acc = fmaf(wc[j], Elem<T>::load(&Vs[len + tt][c]), acc);
        }
      }
      a.mem[(size_t)(row0 + t) * a.d + hc] = acc;
    }
  }
  if (a.roll_rows > 0) {                                 // history <- last cap of (history ++ current[:roll_rows]); every old row is in LDS already
    const int total = len + a.roll_rows, new_len = min(total, a.cap), drop = total - new_len;
    T* wk = const_cast<T*>(ck);
    T* wv = const_cast<T*>(cv);
    if (drop > 0 || a.roll_rows > 0) {
      for (int e = tid; e < new_len * SA_HD; e += 256) {
        const int p = e >> 7, c = e & 127, src = p + drop;
        if (src >= len || drop > 0) {                    // rows that move or are new (values round-trip exactly: they came from T)
          wk[(size_t)p * SA_HD + c] = Ks[src][c];
          wv[(size_t)p * SA_HD + c] = Vs[src][c];
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(128) void stream_cache_roll_kernel(T* cache_k, T* cache_v, const int32_t* __restrict__ cache_len, int cap,
                                                                const T* __restrict__ k, int ld_k, int k_col0, const T* __restrict__ v,
                                                                int ld_v, int v_col0, int n_app, const UttPlan* __restrict__ plan,
                                                                const UttPlan* __restrict__ cond_plan, int n_heads) {
  const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  if (cond_plan && cond_plan[i].T <= 0) return;
  const int sid = plan[i].lang, row0 = plan[i].row_off;
  const int len = cache_len[sid], total = len + n_app, new_len = min(total, cap), drop = total - new_len;
  T* ck = cache_k + ((size_t)sid * n_heads + h) * cap * SA_HD;
  T* cv = cache_v + ((size_t)sid * n_heads + h) * cap * SA_HD;
  for (int p = 0; p < new_len; ++p) {                // ascending: a thread only ever re-reads positions it has not overwritten yet
    const int src = p + drop;
    if (src < len) {
      if (drop) { ck[(size_t)p * SA_HD + tid] = ck[(size_t)src * SA_HD + tid]; cv[(size_t)p * SA_HD + tid] = cv[(size_t)src * SA_HD + tid]; }
    } else {
      const size_t r = (size_t)(row0 + src - len);
      ck[(size_t)p * SA_HD + tid] = k[r * ld_k + k_col0 + h * SA_HD + tid];
      cv[(size_t)p * SA_HD + tid] = v[r * ld_v + v_col0 + h * SA_HD + tid];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stream_fsmn_kernel(const T* __restrict__ v, int ld_v, int v_col0, const float* __restrict__ w,
                                                          const float* __restrict__ b, int d, int ktaps, int n_cur, float* __restrict__ mem) {
  const int m = blockIdx.x, t = m & 15, base = m - t, pad = (ktaps - 1) / 2;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.0f;
    if (t < n_cur) {
      acc = b[c];
      for (int j = 0; j < ktaps; ++j) {
        const int tt = t + j - pad;
        if (tt >= 0 && tt < n_cur) acc = fmaf(w[c * ktaps + j], Elem<T>::load(v + (size_t)(base + tt) * ld_v + v_col0 + c), acc);
      }
    }
    mem[(size_t)m * d + c] = acc;
  }
}

__global__ __launch_bounds__(256) void stream_cif_kernel(const float* __restrict__ alpha, const float* __restrict__ enc, int d,
                                                         const UttPlan* __restrict__ plan, int n_int, float* __restrict__ cif_hidden,
                                                         float* __restrict__ cif_alphas, float* __restrict__ frames_out,
                                                         UttPlan* __restrict__ token_plan, int32_t* __restrict__ num) {
  const int i = blockIdx.x, sid = plan[i].lang, row0 = plan[i].row_off;
  const float ca0 = cif_alphas[sid];
  int n_fired = 0;
  float ca_final = 0.0f;
  for (int c = threadIdx.x; c < d; c += 256) {
    float ca = ca0;
    const float ch = cif_hidden[(size_t)sid * d + c];
    int k = 0;
    float cond_a = ca < 1.0f ? 1.0f : 0.0f, cond_b = 1.0f - cond_a;
    float frames = ca * ch * cond_a + ch * cond_b;
    float listed = frames;                                 // last entry of the reference's list_frame (fired or not)
    if (cond_b != 0.0f) frames_out[(size_t)(row0 + k++) * d + c] = frames;
    ca -= cond_b;
    frames = frames * cond_a + ca * ch * cond_b;
    for (int t = 0; t < n_int; ++t) {
      const float al = alpha[row0 + t], hid = enc[(size_t)(row0 + t) * d + c];
      const float thr = 1.0f - ca;
      cond_a = al < thr ? 1.0f : 0.0f;
      cond_b = 1.0f - cond_a;
      frames = (frames + al * hid) * cond_a + (frames + thr * hid) * cond_b;
      listed = frames;
      if (cond_b != 0.0f) frames_out[(size_t)(row0 + k++) * d + c] = frames;
      ca = ca + al;
      ca -= cond_b;
      frames = frames * cond_a + ca * hid * cond_b;
    }
    cif_hidden[(size_t)sid * d + c] = listed / ca;        // list_frame[:, -1] / cif_alphas, exactly as the reference carries it (:460)
    n_fired = k;
    ca_final = ca;
  }
  __syncthreads();                                         // every thread has read cif_alphas[sid] (before the loop) by now
  if (threadIdx.x == 0) {
    cif_alphas[sid] = ca_final;
    UttPlan tp = plan[i];
    tp.T = n_fired; tp.n_lfr = n_fired;
    token_plan[i] = tp;
    num[i] = n_fired;
  }
}

__global__ __launch_bounds__(256) void stream_dec_fsmn_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                              const float* __restrict__ w, int d, int ktaps,
                                                              const UttPlan* __restrict__ token_plan, float* __restrict__ hist,
                                                              float* __restrict__ out) {
  const int i = blockIdx.x;
  const UttPlan tp = token_plan[i];
  const int n = tp.T, sid = tp.lang, row0 = tp.row_off, nh = ktaps - 1;
  if (n <= 0) return;
  for (int c = threadIdx.x; c < d; c += 256) {
    float cat[32];                                         // nh (10) history columns + up to 16 tokens
    float* hc = hist + ((size_t)sid * nh) * d + c;
    for (int j = 0; j < nh; ++j) cat[j] = hc[(size_t)j * d];
    for (int t = 0; t < n; ++t) cat[nh + t] = x[(size_t)(row0 + t) * d + c];
    for (int t = 0; t < n; ++t) {
      float acc = res[(size_t)(row0 + t) * d + c];
      for (int j = 0; j < ktaps; ++j) acc = fmaf(w[c * ktaps + j], cat[t + j], acc);
      out[(size_t)(row0 + t) * d + c] = acc;
    }
    for (int j = 0; j < nh; ++j) hc[(size_t)j * d] = cat[n + j];
  }
}

__global__ void stream_advance_kernel(const UttPlan* __restrict__ plan, const UttPlan* __restrict__ token_plan, int n_active, int en_add,
                                      int en_cap, int de_add, int de_cap, int32_t* __restrict__ en_len, int32_t* __restrict__ de_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_active) return;
  const int sid = plan[i].lang;
  en_len[sid] = min(en_len[sid] + en_add, en_cap);
  if (token_plan[i].T > 0) de_len[sid] = min(de_len[sid] + de_add, de_cap);
}

}  // namespace

void launch_stream_lfr(const StreamLfrArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_prev + a.n_new <= 16 && a.n_rows % 16 == 0, "stream_lfr: %d + %d rows do not fit a 16-row slot", a.n_prev, a.n_new);
  hipLaunchKernelGGL(stream_lfr_kernel, dim3(a.n_rows), dim3(256), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

void launch_stream_carry(const float* x, int ld, const UttPlan* plan, int n_active, int n_prev, int n_new, float* prev, int32_t* start,
                         hipStream_t s) {
  hipLaunchKernelGGL(stream_carry_kernel, dim3(n_active), dim3(256), 0, s, x, ld, plan, n_prev, n_new, prev, start);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_stream_attn(const StreamAttnArgs& a, int n_active, hipStream_t s) {
  ASR_REQUIRE(a.cap + a.n_cur <= SA_MAXK, "stream_attn: %d + %d keys exceed %d", a.cap, a.n_cur, SA_MAXK);
  // f32 sessions: the K / V images are dynamic LDS, and the request covers the rest of the CU: nothing else shares the CU's LDS with this workgroup (see the kernel)
  size_t pad = 0;
  if (sizeof(T) == 4) {
    pad = SA_F32_DYN;
    static PerDeviceOnce once;
    if (once.first()) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_attn_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
  }
  hipLaunchKernelGGL(stream_attn_kernel<T>, dim3(n_active, a.n_heads), dim3(256), pad, s, a);
  HIP_CHECK(hipGetLastError());
}
template void launch_stream_attn<float>(const StreamAttnArgs&, int, hipStream_t);
template void launch_stream_attn<bf16_t>(const StreamAttnArgs&, int, hipStream_t);

template <typename T>
void launch_stream_cache_roll(void* cache_k, void* cache_v, const int32_t* cache_len, int cap, const void* k, int ld_k, int k_col0,
                              const void* v, int ld_v, int v_col0, int n_app, const UttPlan* plan, const UttPlan* cond_plan, int n_active,
                              int n_heads, hipStream_t s) {
  hipLaunchKernelGGL(stream_cache_roll_kernel<T>, dim3(n_active, n_heads), dim3(128), 0, s, (T*)cache_k, (T*)cache_v, cache_len, cap,
                     (const T*)k, ld_k, k_col0, (const T*)v, ld_v, v_col0, n_app, plan, cond_plan, n_heads);
  HIP_CHECK(hipGetLastError());
}
template void launch_stream_cache_roll<float>(void*, void*, const int32_t*, int, const void*, int, int, const void*, int, int, int,
                                              const UttPlan*, const UttPlan*, int, int, hipStream_t);
template void launch_stream_cache_roll<bf16_t>(void*, void*, const int32_t*, int, const void*, int, int, const void*, int, int, int,
                                               const UttPlan*, const UttPlan*, int, int, hipStream_t);

template <typename T>
void launch_stream_fsmn(const T* v, int ld_v, int v_col0, const float* w, const float* b, int d, int ktaps, int n_cur, int n_rows, float* mem,
                        hipStream_t s) {
  hipLaunchKernelGGL(stream_fsmn_kernel<T>, dim3(n_rows), dim3(256), 0, s, v, ld_v, v_col0, w, b, d, ktaps, n_cur, mem);
  HIP_CHECK(hipGetLastError());
}
template void launch_stream_fsmn<float>(const float*, int, int, const float*, const float*, int, int, int, int, float*, hipStream_t);
template void launch_stream_fsmn<bf16_t>(const bf16_t*, int, int, const float*, const float*, int, int, int, int, float*, hipStream_t);

void launch_stream_cif(const float* alpha, const float* enc, int d, const UttPlan* plan, int n_active, int n_int, float* cif_hidden,
                       float* cif_alphas, float* frames_out, UttPlan* token_plan, int32_t* num, hipStream_t s) {
  ASR_REQUIRE(n_int + 1 <= 16, "stream_cif: %d frames per chunk exceed the 16-row slot", n_int + 1);
  hipLaunchKernelGGL(stream_cif_kernel, dim3(n_active), dim3(256), 0, s, alpha, enc, d, plan, n_int, cif_hidden, cif_alphas, frames_out,
                     token_plan, num);
  HIP_CHECK(hipGetLastError());
}

void launch_stream_dec_fsmn(const float* x, const float* res, const float* w, int d, int ktaps, const UttPlan* token_plan, int n_active,
                            float* hist, float* out, hipStream_t s) {
  ASR_REQUIRE(ktaps - 1 + 16 <= 32, "stream_dec_fsmn: %d taps", ktaps);
  hipLaunchKernelGGL(stream_dec_fsmn_kernel, dim3(n_active), dim3(256), 0, s, x, res, w, d, ktaps, token_plan, hist, out);
  HIP_CHECK(hipGetLastError());
}

void launch_stream_advance(const UttPlan* plan, const UttPlan* token_plan, int n_active, int en_add, int en_cap, int de_add, int de_cap,
                           int32_t* en_len, int32_t* de_len, hipStream_t s) {
  hipLaunchKernelGGL(stream_advance_kernel, dim3((n_active + 63) / 64), dim3(64), 0, s, plan, token_plan, n_active, en_add, en_cap, de_add,
                     de_cap, en_len, de_len);
  HIP_CHECK(hipGetLastError());
}


// ---- snapshot / restore of the active streams' recurrent state (kernels.h: StreamStateSeg). blockIdx.x = active stream, blockIdx.y = (segment, outer index)
__global__ __launch_bounds__(256) void stream_state_copy_kernel(const StreamStateSeg* segs, int n_segs, const UttPlan* plan, int restore) {
  const int item = blockIdx.y;
  int sg = 0;
  while (sg + 1 < n_segs && segs[sg + 1].first_item <= item) ++sg;
  const StreamStateSeg g = segs[sg];
  const int sid = plan[blockIdx.x].lang;
  const size_t off = (size_t)(item - g.first_item) * g.outer_stride + (size_t)sid * g.per_stream;
  const unsigned char* src = (restore ? g.shadow : g.live) + off;
  unsigned char* dst = (restore ? g.live : g.shadow) + off;
  if ((g.per_stream & 15) == 0) {
    for (size_t i = threadIdx.x; i < g.per_stream / 16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  } else {
    for (size_t i = threadIdx.x; i < g.per_stream / 4; i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
  }
}
void launch_stream_state_copy(const StreamStateSeg* segs, int n_segs, int n_items, const UttPlan* plan, int n_active, bool restore, hipStream_t s) {
  hipLaunchKernelGGL(stream_state_copy_kernel, dim3(n_active, n_items), dim3(256), 0, s, segs, n_segs, plan, restore ? 1 : 0);
  HIP_CHECK(hipGetLastError());
}
