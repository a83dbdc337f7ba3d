// Fused SANM attention half for windows of <= 144 encoder rows (8 s chunks: T = 137 -> 144 padded rows):
// ONE workgroup per (utterance, head) computes that head's q | k | v projection on MFMA, keeps the three
// [144 x 128] results in LDS, runs the soft-max attention and the FSMN memory on them and writes only the
// context (bf16) and the FSMN term (f32). The q/k/v activations, the transposed V copy and two launches of the
// unfused path never touch memory (Export_SenseVoice.py:227-245: linear_q_k_v -> softmax(QK^T)V and fsmn(V)).
//
//  phase 1  C[144][384] = h[144][K] Wh[384][K]^T, K-step 64, two LDS stages filled by LDS-DMA with the 16-byte-slot
//           XOR swizzle on the global side. 12 waves = 3 column groups (q, k, v) x 4 waves of 32 columns; every wave
//           covers all 9 row fragments (A fragments are shared by its 2 column fragments: 11 LDS reads per 18 MFMA).
//           q / k waves use the swapped operand order (4 consecutive columns per lane -> row-major LDS image), v waves
//           the plain order (4 consecutive rows per lane -> V^T image, time contiguous).
//  phase 2  all threads: FSMN (thread = channel x 24-step segment, 11 taps from the V^T image, row-major coalesced f32
//           stores); waves 0..8: attention for query tile w -- S^T = K Q^T so the soft-max is lane local and P is directly
//           the B fragment of O^T = V^T P^T; with <= 160 keys all scores stay in registers: one soft-max pass, no rescale.
// The projection and the FSMN reproduce the separate kernels bit for bit; the attention differs from the chunked
// online-softmax kernel only in summation order.
#include <type_traits>
#include "kernels.h"

namespace {

constexpr int FR = 144;                     // rows per workgroup tile
constexpr int FMI = FR / 16;                // row fragments
constexpr int FHD = 128;                    // head dim
constexpr int FNW = 12;                     // waves
constexpr int FA_BYTES = FR * 128;          // A stage: 144 rows x 64 bf16
constexpr int FW_BYTES = 3 * FHD * 128;     // W stage: 384 rows x 64 bf16
constexpr int FSTAGE = FA_BYTES + FW_BYTES;
constexpr int FA_INSTR = FR / 8;            // LDS-DMA wave-instructions (8 rows x 128 B) per A stage
constexpr int FW_INSTR = 3 * FHD / 8;
constexpr int FKEYS = 160;                  // key rows addressable in phase 2 (32-key sub-tiles)
constexpr int FQS = 0;                      // q  [144][256 B]
constexpr int FKS = FQS + FR * 256;         // k  [160][256 B]
constexpr int FVS = FKS + FKEYS * 256;      // v^T [128][512 B] (32 slots of 8 keys, 20 used)
constexpr int FLDS2 = FVS + FHD * 512;
constexpr int FST_P = (2 * FSTAGE > FLDS2) ? 2 * FSTAGE : FLDS2;   // LayerNorm statistics: partial sums [4][144] float2
constexpr int FST_F = FST_P + 4 * FR * 8;                          // (mean, rstd) [144] float2
constexpr int FLDS = FST_F + FR * 8;
constexpr int FTAPS = 11;

__device__ __forceinline__ void wait_all_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(768) void sanm_qkv_attn_kernel(const SanmFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, sub = wave & 3;              // grp: 0 q, 1 k, 2 v
  const int frow = lane & 15, fgrp = lane >> 4;
  const int u = blockIdx.x / a.n_heads, h = blockIdx.x % a.n_heads;
  const UttPlan up = a.plan[u];
  const int T = up.T, row0 = up.row_off;
  const int n_act = (T + 15) >> 4;                        // active row fragments (workgroup-uniform)

  // ---------------------------------------------------------------- phase 1: projection
  const int srow = lane >> 3, sslot = (lane & 7) ^ srow;  // every staged row index is == srow (mod 8)
  const bf16_t* hb = reinterpret_cast<const bf16_t*>(a.h);
  const bf16_t* wb = reinterpret_cast<const bf16_t*>(a.wqkv);
  auto src_of = [&](int ii) -> const bf16_t* {            // ii: wave-instruction index inside a stage
    if (ii < FA_INSTR) {
      const int r = min(row0 + ii * 8 + srow, a.n_rows_alloc - 1);
      return hb + (size_t)r * a.ld_h + sslot * 8;
    }
    const int wr = (ii - FA_INSTR) * 8 + srow;            // 0..383: q | k | v rows of this head
    return wb + (size_t)((wr >> 7) * a.d + h * FHD + (wr & 127)) * a.ldw + sslot * 8;
  };
  const bf16_t* src[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) src[t] = src_of(min(wave + t * FNW, FA_INSTR + FW_INSTR - 1));
  const int n_instr = (wave + 5 * FNW < FA_INSTR + FW_INSTR) ? 6 : 5;
  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * FSTAGE;
#pragma unroll
    for (int t = 0; t < 6; ++t)
      if (t < n_instr)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                         (__attribute__((address_space(3))) void*)(base + (wave + t * FNW) * 1024), 16, 0, 0);
  };

  f32x4_t acc[FMI][2];
#pragma unroll
  for (int i = 0; i < FMI; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const int nk = a.K / 64;
  const int a_lane = frow * 128, w_lane = FA_BYTES + (grp * FHD + sub * 32 + frow) * 128, key7 = frow & 7;
  const bool ln = a.ln_colsum != nullptr, ln_here = ln && !a.ln_stats_in;     // ln_here: statistics accumulated from the LDS tiles
  const int st_row = tid % FR, st_sg = tid / FR;          // statistics: threads < 576 own (row, two 16-byte K slots)
  float st_s = 0.0f, st_ss = 0.0f;
  if (ln && !ln_here && tid < FR) {         // producer-side statistics: finalise (mean, rstd) while the first stage flies
    const float2* sp = a.ln_stats_in + (size_t)min(row0 + tid, a.n_rows_alloc - 1) * a.ln_slots;
    const float2 ss = sum_row_partials(sp, a.ln_slots);
    const float inv_d = 1.0f / (float)a.ln_dim;
    const float mean = ss.x * inv_d;
    const float var = fmaxf(ss.y * inv_d - mean * mean, 0.0f);
    reinterpret_cast<float2*>(smem + FST_F)[tid] = make_float2(mean, rsqrtf(var + a.ln_eps));
  }
  auto k_loop = [&](auto full_tag, auto swap_tag) {
    constexpr bool FULL = decltype(full_tag)::value;      // FULL: all 9 row fragments, straight-line MFMA body
    constexpr bool SWAP = decltype(swap_tag)::value;      // q / k waves: mfma(W, A); v waves: mfma(A, W)
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      wait_all_dma();
      __builtin_amdgcn_s_barrier();          // stage kt landed for every wave; every wave finished reading the other slot
      if (kt + 1 < nk && !(a.dbg & 1)) stage((kt + 1) & 1, (kt + 1) * 64);
      if (a.dbg & 2) continue;
      const unsigned char* St = smem + (kt & 1) * FSTAGE;
      if (ln_here && wave < FMI) {                         // row sums of the raw operand tile (VALU, beside the MFMA waves)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint4 raw = *reinterpret_cast<const uint4*>(St + st_row * 128 + (((2 * st_sg + q) ^ (st_row & 7)) << 4));
          const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(wds[e] << 16), hi = __uint_as_float(wds[e] & 0xffff0000u);
            st_s += lo + hi;
            st_ss = fmaf(lo, lo, fmaf(hi, hi, st_ss));
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fgrp) ^ key7) << 4;
        const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(St + w_lane + sw);
        const bf16x8_t w1 = *reinterpret_cast<const bf16x8_t*>(St + w_lane + 16 * 128 + sw);
        bf16x8_t af[FMI];
#pragma unroll
        for (int i = 0; i < FMI; ++i)
          if (FULL || i < n_act) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_lane + i * 2048 + sw);
#pragma unroll
        for (int i = 0; i < FMI; ++i) {
          if (FULL || i < n_act) {
            if constexpr (SWAP) {
              acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af[i], acc[i][0], 0, 0, 0);
              acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af[i], acc[i][1], 0, 0, 0);
            } else {
              acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], w0, acc[i][0], 0, 0, 0);
              acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], w1, acc[i][1], 0, 0, 0);
            }
          }
        }
      }
    }
  };
  if (n_act == FMI) {
    if (grp < 2) k_loop(std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{});
  } else {
    if (grp < 2) k_loop(std::false_type{}, std::true_type{}); else k_loop(std::false_type{}, std::false_type{});
  }
  float2* st_part = reinterpret_cast<float2*>(smem + FST_P);
  float2* st_fin = reinterpret_cast<float2*>(smem + FST_F);
  if (ln_here && wave < FMI) st_part[st_sg * FR + st_row] = make_float2(st_s, st_ss);
  __syncthreads();                           // ring is dead: phase-2 images may overwrite it
  if (ln_here) {
    if (tid < FR) {
      const float2 p0 = st_part[tid], p1 = st_part[FR + tid], p2 = st_part[2 * FR + tid], p3 = st_part[3 * FR + tid];
      const float inv_d = 1.0f / (float)a.ln_dim;
      const float mean = ((p0.x + p1.x) + (p2.x + p3.x)) * inv_d;
      const float var = fmaxf(((p0.y + p1.y) + (p2.y + p3.y)) * inv_d - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + a.ln_eps));
    }
    __syncthreads();
  }

  if (grp < 2) {            // acc[i][j][r] = C[16 i + frow][32 sub + 16 j + 4 fgrp + r] -> row-major, 8-byte writes
    unsigned char* dst = smem + (grp == 0 ? FQS : FKS);
    const float* bias = a.bqkv + grp * a.d + h * FHD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = sub * 32 + j * 16 + fgrp * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
      float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ln) c4 = *reinterpret_cast<const float4*>(a.ln_colsum + grp * a.d + h * FHD + col);
#pragma unroll
      for (int i = 0; i < FMI; ++i) {
        const int row = i * 16 + frow;
        float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
        if (ln) {
          const float2 mr = st_fin[row];
          v0 = (v0 - mr.x * c4.x) * mr.y; v1 = (v1 - mr.x * c4.y) * mr.y; v2 = (v2 - mr.x * c4.z) * mr.y; v3 = (v3 - mr.x * c4.w) * mr.y;
        }
        uint2 w;
        w.x = pack_bf16x2(v0 + b4.x, v1 + b4.y);
        w.y = pack_bf16x2(v2 + b4.z, v3 + b4.w);
        *reinterpret_cast<uint2*>(dst + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + ((col >> 2) & 1) * 8) = w;
      }
    }
  } else {                  // acc[i][j][r] = C[16 i + 4 fgrp + r][32 sub + 16 j + frow] -> V^T[d][t], 8-byte writes
    unsigned char* dst = smem + FVS;
    const float* bias = a.bqkv + 2 * a.d + h * FHD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int dcol = sub * 32 + j * 16 + frow;
      const float b = bias[dcol];
      const float cs = ln ? a.ln_colsum[2 * a.d + h * FHD + dcol] : 0.0f;
#pragma unroll
      for (int i = 0; i < FMI; ++i) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (ln) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float2 mr = st_fin[i * 16 + fgrp * 4 + r]; v[r] = (v[r] - mr.x * cs) * mr.y; }
        }
        uint2 w;
        w.x = pack_bf16x2(v[0] + b, v[1] + b);
        w.y = pack_bf16x2(v[2] + b, v[3] + b);
        *reinterpret_cast<uint2*>(dst + dcol * 512 + (((2 * i + (fgrp >> 1)) ^ (dcol & 15)) << 4) + (fgrp & 1) * 8) = w;
      }
    }
    // keys 144..159 of the last 32-key sub-tile: finite (zero) values so that 0 * v stays 0
    if (sub == 0 && lane < 32) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int dcol = rr * 16 + (lane & 15), slot = 18 + (lane >> 4);
        *reinterpret_cast<uint4*>(dst + dcol * 512 + ((slot ^ (dcol & 15)) << 4)) = make_uint4(0, 0, 0, 0);
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- phase 2
  if (a.dbg & 4) return;
  // ---- FSMN memory on ALL threads: thread = (channel, 24-step time segment); 5 slots of 8 steps (own 3 + one halo slot
  //      each side) come from the V^T image, 11 taps per output, row-major coalesced f32 stores (lanes = channels)
  if (!(a.dbg & 16)) {
    constexpr int PAD = (FTAPS - 1) / 2;
    const int c = tid & (FHD - 1), seg = tid >> 7, cg = h * FHD + c;      // 6 segments x 24 steps = 144
    const int t0 = seg * 24, T16 = n_act * 16;
    if (t0 < T16) {
      const unsigned char* vrow = smem + FVS + c * 512;
      float wc[FTAPS];
#pragma unroll
      for (int j = 0; j < FTAPS; ++j) wc[j] = a.wfsmn[cg * FTAPS + j];
      const float bc = a.bfsmn[cg];
      float x[40];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int sl = seg * 3 - 1 + q;                       // slot index, 8 steps each; -1 and >= 18 are outside the window
        if (sl < 0 || sl * 8 >= T) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[q * 8 + e] = 0.0f;
        } else {
          const uint4 raw = *reinterpret_cast<const uint4*>(vrow + ((sl ^ (c & 15)) << 4));
          const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[q * 8 + 2 * e] = (sl * 8 + 2 * e < T) ? __uint_as_float(wds[e] << 16) : 0.0f;
            x[q * 8 + 2 * e + 1] = (sl * 8 + 2 * e + 1 < T) ? __uint_as_float(wds[e] & 0xffff0000u) : 0.0f;
          }
        }
      }
      float* out = a.mem + (size_t)(row0 + t0) * a.ld_mem + cg;
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        float accv = bc;
#pragma unroll
        for (int j = 0; j < FTAPS; ++j) accv = fmaf(wc[j], x[8 + i + j - PAD], accv);
        if (t0 + i < T16) out[(size_t)i * a.ld_mem] = (t0 + i < T) ? accv : 0.0f;
      }
    }
  }
  // ---- attention, waves 0..8 = query tile. T <= 160 keys: all scores of a tile live in registers (10 independent MFMA
  //      accumulators), so ONE soft-max pass (no running max / rescale) and then 8 independent P.V chains
  if (wave < FMI) {
    if (a.dbg & 8) return;
    const int q0 = wave * 16;
    if (q0 >= T) return;
    const unsigned char* Qs = smem + FQS;
    const unsigned char* Ks = smem + FKS;
    const unsigned char* Vs = smem + FVS;
    const int fq = frow, g = fgrp;
    const int qrow = q0 + fq;
    const int n_sub = (T + 31) >> 5;                          // 32-key sub-tiles in use (<= 5)
    bf16x8_t qf[FHD / 32];
#pragma unroll
    for (int ks = 0; ks < FHD / 32; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + qrow * 256 + (((ks * 4 + g) ^ (qrow & 15)) << 4));
    f32x4_t st[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s) { st[s][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; st[s][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (s < n_sub) {
        const int key0 = s * 32 + fq, key1 = key0 + 16;
#pragma unroll
        for (int ks = 0; ks < FHD / 32; ++ks) {
          const int c = ks * 4 + g;
          const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t*>(Ks + key0 * 256 + ((c ^ (key0 & 15)) << 4));
          const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t*>(Ks + key1 * 256 + ((c ^ (key1 & 15)) << 4));
          st[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[ks], st[s][0], 0, 0, 0);
          st[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[ks], st[s][1], 0, 0, 0);
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = s * 32 + hlf * 16 + g * 4 + r;
          if (key >= T) st[s][hlf][r] = -INFINITY;
          mx = fmaxf(mx, st[s][hlf][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.0f;
    bf16x8_t pf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      float p[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { p[r] = __expf(st[s][r >> 2][r & 3] - mx); l += p[r]; }
      union { bf16x8_t v; uint32_t w[4]; } u8;
#pragma unroll
      for (int r = 0; r < 4; ++r) u8.w[r] = pack_bf16x2(p[2 * r], p[2 * r + 1]);
      pf[s] = u8.v;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4_t ot[FHD / 16];
#pragma unroll
    for (int dt = 0; dt < FHD / 16; ++dt) ot[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (s < n_sub) {
#pragma unroll
        for (int dt = 0; dt < FHD / 16; ++dt) {
          const int d = dt * 16 + fq;
          const unsigned char* vr = Vs + d * 512 + (g & 1) * 8;
          union { bf16x8_t v; uint2 h2[2]; } vf;
          vf.h2[0] = *reinterpret_cast<const uint2*>(vr + (((s * 4 + (g >> 1)) ^ (d & 15)) << 4));
          vf.h2[1] = *reinterpret_cast<const uint2*>(vr + (((s * 4 + 2 + (g >> 1)) ^ (d & 15)) << 4));
          ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[s], ot[dt], 0, 0, 0);
        }
      }
    }
    if (qrow < T) {
      bf16_t* op = reinterpret_cast<bf16_t*>(a.ctx) + (size_t)(row0 + qrow) * a.ld_ctx + h * FHD + g * 4;
#pragma unroll
      for (int dt = 0; dt < FHD / 16; ++dt) {
        uint2 w;
        w.x = pack_bf16x2(ot[dt][0] * inv, ot[dt][1] * inv);
        w.y = pack_bf16x2(ot[dt][2] * inv, ot[dt][3] * inv);
        *reinterpret_cast<uint2*>(op + dt * 16) = w;
      }
    }
  }
}

}  // namespace

bool sanm_fused_supported(int max_T, int d_head, int n_heads, int d, int fsmn_taps, int K) {
  return max_T <= FR && d_head == FHD && n_heads * FHD == d && fsmn_taps == FTAPS && K % 64 == 0;
}

void launch_sanm_qkv_attn(const SanmFusedArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.K % 64 == 0 && a.ld_h % 8 == 0 && a.ldw % 8 == 0 && a.n_heads * FHD == a.d && a.ld_mem % 4 == 0 && a.ld_ctx % 4 == 0,
              "sanm_fused: bad geometry (K=%d d=%d heads=%d)", a.K, a.d, a.n_heads);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sanm_qkv_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FLDS));
  }
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("ASR_FUSED_DBG"); dbg = e ? atoi(e) : 0; }     // kernel ablation switches (timing only)
  SanmFusedArgs b = a;
  b.dbg = dbg;
  hipLaunchKernelGGL(sanm_qkv_attn_kernel, dim3(a.n_utts * a.n_heads), dim3(FNW * 64), FLDS, s, b);
  HIP_CHECK(hipGetLastError());
}
