// Paraformer online decoder, every layer of ONE chunk step as ONE launch (Paraformer/Streaming/Export_Paraformer_Streaming.py:508-553: per layer
// FFN (LayerNorm -> w_1 + relu -> LayerNorm over 2048 -> w_2), LayerNorm -> FSMN over [history | fired tokens] + residual, LayerNorm -> q; k|v of the
// chunk's encoder rows; soft-max attention over [K/V history | chunk rows]; out-projection + residual; the history of a stream that fired rolls on the
// way; the last block is FFN only). bf16 sessions, d = 512 / 4 heads of 128 / d_ffn = 2048.
//
// Same machine as the encoder launch (stream_layers.hip, helpers in stream_cluster.h): the four workgroups (s, h) of stream s form a cluster that meets
// through memory, a slot is one 16-row MFMA tile, wave w of 8 owns 64 / 16 / 16 / 32 / 16 output columns of the five GEMMs and streams their weights from a
// fragment-major copy straight into registers. A decoder layer has five meetings instead of four -- the LayerNorm over the 2048 FFN channels and the
// LayerNorm in front of the FSMN each need whole rows:
//
//   1  LN(dec) -> FFN-1 columns 512 h ..  (f32 slab)                                                  -- exchange 0: hid, f32 [16][2048] -->
//   2  LN over 2048 -> FFN-2 columns 128 h .. + b2 = x1 slab                                          -- exchange 1: x1 -->
//      (while waiting: k|v of head h from the chunk's encoder rows -- they do not depend on the layer's input -- and the history rows, into the images)
//   3  LN(x1) (affine, own 128 channels) -> FSMN over [10 history rows | tokens] + dec = x2 slab, history advanced      -- exchange 2: x2 -->
//   4  LN(x2) -> q of head h, attention (scores / P V on the matrix pipe), K/V history rolled          -- exchange 3: ctx -->
//   5  out-projection columns 128 h .. + bo + x2 = dec slab of the next layer                          -- exchange 4: dec -->
//
// A stream that fired no token in this chunk step leaves the launch at once (all four of its workgroups: its decoder state stays untouched, as on the
// per-launch path). Rounding points are those of the per-launch path (bf16 GEMM operands, bf16 q / k / v / ctx, f32 everything else).
#include <type_traits>
#include "stream_cluster.h"

namespace {

constexpr int D = 512, DFF = 2048, HD = 128, SLOT = 16, MAXK = 64, TAPS = 11, NHIST = TAPS - 1;
// ---- fragment-major weight copy of one layer: [matrix][head][wave][fragment][64 lanes][16 B]
constexpr size_t PW_1 = 16 * 4 * 1024, PW_2 = 64 * 1024, PW_Q = 16 * 1024, PW_KV = 16 * 2 * 1024, PW_O = 16 * 1024;
constexpr size_t PK_1 = 0, PK_2 = PK_1 + NH * NW * PW_1, PK_Q = PK_2 + NH * NW * PW_2, PK_KV = PK_Q + NH * NW * PW_Q, PK_O = PK_KV + NH * NW * PW_KV,
                 PK_BYTES = PK_O + NH * NW * PW_O;
static_assert(PK_BYTES == (size_t)(2 * DFF * D + D * D + 2 * D * D + D * D) * 2, "the packed copy holds every weight element once");
// ---- LDS map (bytes)
constexpr int AS = D * 2 + 16, HS = DFF * 2 + 16, KS = HD * 2 + 16, PS = MAXK * 2 + 16;
constexpr int XN = 0, CTX = XN + SLOT * AS;    // normalised rows (FFN-1, q) / attention context (out-projection); together: the f32 FFN-1 slab [16][512] on its way out
constexpr int ENC = CTX + SLOT * AS;           // the chunk's encoder rows, bf16 [16][512]: loaded once, the A operand of every layer's k|v projection
constexpr int UNI = ENC + SLOT * AS;           // union: hid [16][2048] bf16 (FFN-2) | K / V images (bf16 [64][128])
constexpr int HID = UNI, KB = UNI, VB = KB + MAXK * KS, UNI_END = HID + SLOT * HS;
constexpr int QB = UNI_END, SF = QB + SLOT * KS, PB = SF + SLOT * (MAXK + 1) * 4;
constexpr int DECS = (PB + SLOT * PS + 15) / 16 * 16;              // own 128 columns, f32 [16][128]: the layer's input (residual of the FSMN)
constexpr int CATS = DECS + SLOT * HD * 4;                         // f32 [10 + 16][128]: FSMN history rows, then the x1 slab (raw, then LN(x1) in place)
constexpr int X1S = CATS + NHIST * HD * 4, X2S = X1S + SLOT * HD * 4, LDS_BYTES = X2S + SLOT * HD * 4;
static_assert(SLOT * 512 * 4 <= ENC && VB + MAXK * KS <= UNI_END && LDS_BYTES <= 160 * 1024, "LDS map");

// full rows of one of the cluster's [rows][512] f32 buffers -> two-pass LayerNorm statistics; thread = (row, 16 columns)
//   MODE 0: bf16(normalised) -> the XN operand rows; the raw own 128 columns kept in f32 at `keep` when given
//   MODE 1: own 128 columns only: normalised * g + b in f32 -> `keep`
template <int MODE>
__device__ __forceinline__ void norm_rows(const float* src, unsigned char* smem, int tid, int h, float eps, float* keep, const float* g, const float* b) {
  // thread = (row, column pairs 64 e + 2 j, e < 8): lane-contiguous 8-byte loads (256 contiguous bytes per row and instruction)
  const int row = tid >> 5, j = tid & 31;
  const float* p = src + (size_t)row * D + 2 * j;
  float v[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const u64 t = get8(p + 64 * e);
    v[2 * e] = __uint_as_float((unsigned)t);
    v[2 * e + 1] = __uint_as_float((unsigned)(t >> 32));
  }
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; e += 4) s += (v[e] + v[e + 1]) + (v[e + 2] + v[e + 3]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / D);
  float q = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    const float a0 = v[e] - mean, a1 = v[e + 1] - mean, a2 = v[e + 2] - mean, a3 = v[e + 3] - mean;
    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool own = (e >> 1) == h;
    const int c = 64 * e + 2 * j;
    if constexpr (MODE == 0) {
      *reinterpret_cast<unsigned*>(smem + XN + row * AS + c * 2) = pack_bf16x2((v[2 * e] - mean) * rstd, (v[2 * e + 1] - mean) * rstd);
      if (keep && own) *reinterpret_cast<float2*>(keep + row * HD + (c & 127)) = make_float2(v[2 * e], v[2 * e + 1]);
    } else {
      if (own) {
        const f32x2_t gg = *glob(reinterpret_cast<const f32x2_t*>(g + c)), bb = *glob(reinterpret_cast<const f32x2_t*>(b + c));
        *reinterpret_cast<float2*>(keep + row * HD + (c & 127)) = make_float2((v[2 * e] - mean) * rstd * gg[0] + bb[0], (v[2 * e + 1] - mean) * rstd * gg[1] + bb[1]);
      }
    }
  }
}

// own [16][128] f32 slab in LDS -> the cluster's [rows][512] f32 buffer
__device__ __forceinline__ void put_slab_f32(float* dst_rows, const unsigned char* slab, int tid, int h) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + NT * e, row = idx >> 6, off = (idx & 63) * 8;
    put8(reinterpret_cast<unsigned char*>(dst_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(slab + row * HD * 4 + off));
  }
}

#define STAMP(k) do { if (a.times && li == a.times_layer && threadIdx.x == 0) a.times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(NT) void stream_dec_kernel(const StreamDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid_0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid_0 >> 6);
  // cluster members: workgroup ids 8 apart (one XCD). a.opt & 8: placement by head as in the encoder launch (XCDs 2 h, 2 h + 1 host head h) -- measured SLOWER here
  // (0.73 against 0.69 ms: the 128 KB f32 hid meeting is read by four workgroups that then sit on four XCDs)
  int cl, h;
  if (!(a.opt & 8)) { const int idx = blockIdx.x >> 3; cl = ((idx >> 2) << 3) + (blockIdx.x & 7); h = idx & 3; }
  else { const int x = blockIdx.x & 7; h = x >> 1; cl = (blockIdx.x >> 3) * 2 + (x & 1); }
  if (cl >= a.n_streams) return;
  const UttPlan tp = a.token_plan[cl];
  const int T = __builtin_amdgcn_readfirstlane(tp.T), sid = tp.lang, row0 = tp.row_off, n_cur = a.n_cur;
  if (T <= 0) return;                                       // no fired frame: the stream's decoder state is not touched (all four workgroups agree)
  const int len = __builtin_amdgcn_readfirstlane(a.cache_len[sid]), nk = len + n_cur;
  float* Sf = reinterpret_cast<float*>(smem + SF);
  float* decs = reinterpret_cast<float*>(smem + DECS);
  float* cats = reinterpret_cast<float*>(smem + CATS);
  float* x1s = reinterpret_cast<float*>(smem + X1S);
  float* x2s = reinterpret_cast<float*>(smem + X2S);
  float* dec_rows = a.dec + (size_t)row0 * D;
  float* x1_rows = a.x1 + (size_t)row0 * D;
  float* x2_rows = a.x2 + (size_t)row0 * D;
  float* hid_rows = a.hid + (size_t)row0 * DFF;
  bf16_t* ctx_rows = a.ctx + (size_t)row0 * D;
  const size_t wave_frag = (size_t)(h * NW + wave);
  const bool by_head = (a.opt & 8) != 0;
  const int xcd = blockIdx.x & 7;
  const int n_wg_xcd = by_head ? (a.n_streams - (xcd & 1) + 1) >> 1 : ((a.n_streams - xcd + 7) >> 3) * NH, wg_xcd = by_head ? cl >> 1 : (cl >> 3) * NH + h;
  unsigned sink = 0, tw = 0;
  {   // the chunk's encoder rows (bf16, after_norm), once
    const bf16_t* er = a.enc + (size_t)row0 * D;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int s = tid_0 + NT * e, row = s >> 6, c0 = (s & 63) * 8;
      *reinterpret_cast<u32x4*>(smem + ENC + row * AS + c0 * 2) = *reinterpret_cast<const u32x4*>(er + (size_t)row * D + c0);
    }
  }
  u32x4 w1a[16];                                            // batch 0 of FFN-1: requested a layer ahead
  wload<16>(w1a, a.layers[0].wpack + PK_1 + wave_frag * PW_1 + (tid_0 & 63) * 16, 0);

#pragma unroll 1
  for (int li = 0; li < a.n_layers; ++li) {
    const StreamDecLayer& L = a.layers[li];
    const int tid = opaque(tid_0), lane = tid & 63, frow = lane & 15, fgrp = lane >> 4;
    const unsigned char* a_lane = smem + frow * AS + fgrp * 16;
    const unsigned char* hid_lane = smem + HID + frow * HS + fgrp * 16;
    unsigned* flags = a.flags + ((size_t)li * a.n_streams + cl) * 8;
    const unsigned char* wp1 = L.wpack + PK_1 + wave_frag * PW_1 + lane * 16;
    const unsigned char* wp2 = L.wpack + PK_2 + wave_frag * PW_2 + lane * 16;
    const unsigned char* wpq = L.wpack + PK_Q + wave_frag * PW_Q + lane * 16;
    const unsigned char* wpkv = L.wpack + PK_KV + wave_frag * PW_KV + lane * 16;
    const unsigned char* wpo = L.wpack + PK_O + wave_frag * PW_O + lane * 16;
    const bool full = L.full != 0;
    STAMP(0);
    // epilogue constants of the first half, a phase ahead of their use
    float b1v[4], b2v;
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) b1v[j] = glob(L.b1)[h * 512 + wave * 64 + j * 16 + frow];
      b2v = glob(L.b2)[h * HD + wave * 16 + frow];
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 1: LayerNorm of the stream's rows, FFN-1 columns 512 h + 64 wave ..
    if (li > 0) consume(flags - (size_t)a.n_streams * 8 + 4, a.err);
    STAMP(1);
    norm_rows<0>(dec_rows, smem, tid, h, a.ln_eps, decs, nullptr, nullptr);
    lds_barrier();
    STAMP(2);
    {
      f32x4_t acc[4] = {};
      u32x4 w1b[16];
      gemm_phase<4, 16, 4, 1, false>(w1a, w1b, wp1, a_lane + XN, acc);
      lds_barrier();                                         // every wave is done with the operand rows: the f32 slab goes over them
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float*>(smem + XN + ((fgrp * 4 + i) * 512 + wave * 64 + j * 16 + frow) * 4) = fmaxf(acc[j][i] + b1v[j], 0.0f);
      }
    }
    lds_barrier();
    STAMP(3);
    u32x4 w2a[16], w2b[16];
    {   // exchange 0: own [16][512] f32 slab out; all four back in, LayerNorm over the 2048 channels (two passes, f32), bf16 -> hid
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int s = tid + NT * e, row = s >> 8, off = (s & 255) * 8;
        put8(reinterpret_cast<unsigned char*>(hid_rows + (size_t)row * DFF + h * 512) + off, *reinterpret_cast<const u64*>(smem + XN + row * 2048 + off));
      }
      publish(flags + 0);
      sink ^= tw;
      consume(flags + 0, a.err);
      STAMP(4);
      const int row = tid >> 5, j = tid & 31;              // thread = (row, column pairs 64 e + 2 j of every slab): lane-contiguous 8-byte loads
      float v[64];
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const u64 t = get8(hid_rows + (size_t)row * DFF + 64 * e + 2 * j);
        v[2 * e] = __uint_as_float((unsigned)t);
        v[2 * e + 1] = __uint_as_float((unsigned)(t >> 32));
      }
      float s = 0.0f;
#pragma unroll
      for (int e = 0; e < 64; e += 4) s += (v[e] + v[e + 1]) + (v[e + 2] + v[e + 3]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s * (1.0f / DFF);
      float qq = 0.0f;
#pragma unroll
      for (int e = 0; e < 64; e += 4) {
        const float a0 = v[e] - mean, a1 = v[e + 1] - mean, a2 = v[e + 2] - mean, a3 = v[e + 3] - mean;
        qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / DFF) + a.ln_eps);
#pragma unroll
      for (int e = 0; e < 32; ++e)
        *reinterpret_cast<unsigned*>(smem + HID + row * HS + (64 * e + 2 * j) * 2) = pack_bf16x2((v[2 * e] - mean) * rstd, (v[2 * e + 1] - mean) * rstd);
      wload<16>(w2a, wp2, 0);                                 // (64 f32 of the row are live above: no room to request this across the wait)
    }
    lds_barrier();
    STAMP(5);
    // ---- 2: FFN-2 columns 128 h + 16 wave .. + b2 -> x1 slab
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 64, 16, 2, false>(w2a, w2b, wp2, hid_lane, acc);
      const int col = wave * 16 + frow;
#pragma unroll
      for (int i = 0; i < 4; ++i) x1s[(fgrp * 4 + i) * HD + col] = (acc[0][i] + acc[1][i]) + b2v;
    }
    lds_barrier();
    STAMP(6);
    if (!full) {                                             // FFN-only block: x1 is the stream's rows of what follows
      put_slab_f32(dec_rows, smem + X1S, tid, h);
      publish(flags + 4);
      if (li + 1 < a.n_layers) wload<16>(w1a, a.layers[li + 1].wpack + PK_1 + wave_frag * PW_1 + lane * 16, 0);
      continue;
    }
    put_slab_f32(x1_rows, smem + X1S, tid, h);
    publish(flags + 1);
    // ---- while the cluster gathers x1: history rows and k|v of head h from the chunk's encoder rows -> the images (hid is dead)
    bf16_t* ck = L.cache_k + ((size_t)sid * NH + h) * a.cap * HD;
    bf16_t* cv = L.cache_v + ((size_t)sid * NH + h) * a.cap * HD;
    u32x4 rk[2], rv[2];
    float wf[TAPS], hist[3];                                  // FSMN taps of this thread's channel; history rows g, g + 4, g + 8 of it
    {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
        rk[it] = u32x4{0, 0, 0, 0}; rv[it] = u32x4{0, 0, 0, 0};
        if (p < len) {
          rk[it] = *glob(reinterpret_cast<const u32x4*>(ck + (size_t)p * HD + c0));
          rv[it] = *glob(reinterpret_cast<const u32x4*>(cv + (size_t)p * HD + c0));
        }
      }
      const int hc = h * HD + (tid & 127);
#pragma unroll
      for (int j = 0; j < TAPS; ++j) wf[j] = glob(L.wfsmn)[hc * TAPS + j];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int j = (tid >> 7) + 4 * e;
        hist[e] = j < NHIST ? glob(L.fsmn_hist)[((size_t)sid * NHIST + j) * D + hc] : 0.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      u32x4 wkv[8], wkv1[8];
      wload<8>(wkv, wpkv, 0);
      const float bk = glob(L.bkv)[h * HD + wave * 16 + frow], bv = glob(L.bkv)[D + h * HD + wave * 16 + frow];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
        if (p < len) {
          *reinterpret_cast<u32x4*>(smem + KB + p * KS + c0 * 2) = rk[it];
          *reinterpret_cast<u32x4*>(smem + VB + p * KS + c0 * 2) = rv[it];
        } else if (p >= len + SLOT) {
          *reinterpret_cast<u32x4*>(smem + VB + p * KS + c0 * 2) = u32x4{0, 0, 0, 0};
        }
      }
      f32x4_t acc[2] = {};                                   // tile 0: k columns 16 wave .., tile 1: v columns 16 wave ..
      gemm_phase<2, 16, 4, 1, false>(wkv, wkv1, wpkv, a_lane + ENC, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = len + fgrp * 4 + i, c = wave * 16 + frow;
        *reinterpret_cast<bf16_t*>(smem + KB + r * KS + c * 2) = (bf16_t)(pack_bf16x2(acc[0][i] + bk, 0.0f) & 0xffffu);
        *reinterpret_cast<bf16_t*>(smem + VB + r * KS + c * 2) = (bf16_t)(pack_bf16x2(acc[1][i] + bv, 0.0f) & 0xffffu);
      }
    }
    u32x4 wq[16];
    wload<16>(wq, wpq, 0);
    const float bqv = glob(L.bq)[h * HD + wave * 16 + frow], bov = glob(L.bo)[h * HD + wave * 16 + frow];
    __builtin_amdgcn_sched_barrier(0);
    consume(flags + 1, a.err);
    STAMP(7);
    // ---- 3: LayerNorm(x1) with its affine on the own 128 channels, FSMN over [history | tokens] + dec -> x2 slab; the history takes the tokens in
    norm_rows<1>(x1_rows, smem, tid, h, a.ln_eps, x1s, L.n2_g, L.n2_b);
    {
      const int c = tid & 127, g = tid >> 7;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (g + 4 * e < NHIST) cats[(g + 4 * e) * HD + c] = hist[e];
    }
    lds_barrier();
    {   // thread = (channel, four rows): x2[t] = dec[t] + sum_j w[j] cat[t + j] over cat = [10 history rows | LN(x1) of the tokens]; rows past the tokens are zero
      const int c = tid & 127, g = tid >> 7, hc = h * HD + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = g * 4 + i;
        float m = decs[t * HD + c];
#pragma unroll
        for (int j = 0; j < TAPS; ++j) m = fmaf(wf[j], cats[(t + j) * HD + c], m);
        x2s[t * HD + c] = t < T ? m : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {                           // history <- the last 10 of [history | tokens]
        const int j = g + 4 * e;
        if (j < NHIST) glob(L.fsmn_hist)[((size_t)sid * NHIST + j) * D + hc] = cats[(T + j) * HD + c];
      }
    }
    lds_barrier();
    STAMP(8);
    put_slab_f32(x2_rows, smem + X2S, tid, h);
    publish(flags + 2);
    consume(flags + 2, a.err);
    STAMP(9);
    // ---- 4: LayerNorm(x2) -> q of head h, attention over [history | chunk rows]
    norm_rows<0>(x2_rows, smem, tid, h, a.ln_eps, nullptr, nullptr, nullptr);
    lds_barrier();
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 16, 16, 2, false>(wq, wq, wpq, a_lane + XN, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<bf16_t*>(smem + QB + (fgrp * 4 + i) * KS + (wave * 16 + frow) * 2) = (bf16_t)(pack_bf16x2((acc[0][i] + acc[1][i]) + bqv, 0.0f) & 0xffffu);
    }
    u32x4 wo[16];
    wload<16>(wo, wpo, 0);
    lds_barrier();
    STAMP(10);
    if (wave * 16 < nk) {
      f32x4_t sc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks)
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(smem + QB + frow * KS + fgrp * 16 + ks * 64),
                                                     *reinterpret_cast<const bf16x8_t*>(smem + KB + (wave * 16 + frow) * KS + fgrp * 16 + ks * 64), sc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) Sf[(fgrp * 4 + i) * (MAXK + 1) + wave * 16 + frow] = sc[i];
    }
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = wave + 8 * e;
      const float sc = lane < nk ? Sf[q * (MAXK + 1) + lane] : -INFINITY;
      const float mx = wave_max(sc);
      const float ex = lane < nk ? expf(sc - mx) : 0.0f;
      const float sum = wave_sum(ex);
      *reinterpret_cast<bf16_t*>(smem + PB + q * PS + lane * 2) = (bf16_t)(pack_bf16x2(ex / sum, 0.0f) & 0xffffu);
    }
    lds_barrier();
    {
      f32x4_t o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks * 32 < nk) {
          bf16x8_t vf;
#pragma unroll
          for (int e = 0; e < 8; ++e) vf[e] = *reinterpret_cast<const short*>(smem + VB + (ks * 32 + fgrp * 8 + e) * KS + (wave * 16 + frow) * 2);
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(smem + PB + frow * PS + fgrp * 16 + ks * 64), vf, o, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = fgrp * 4 + i;
        *reinterpret_cast<bf16_t*>(smem + CTX + q * AS + (h * HD + wave * 16 + frow) * 2) = q < T ? (bf16_t)(pack_bf16x2(o[i], 0.0f) & 0xffffu) : (bf16_t)0;
      }
    }
    {   // history <- last cap of (history ++ the chunk's n_cur rows): old rows move from the registers they were read into, new rows come from the images
      const int total = len + n_cur, new_len = min(total, a.cap), drop = total - new_len;
      if (drop > 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
          if (p < len && p >= drop) {
            *glob(reinterpret_cast<u32x4*>(ck + (size_t)(p - drop) * HD + c0)) = rk[it];
            *glob(reinterpret_cast<u32x4*>(cv + (size_t)(p - drop) * HD + c0)) = rv[it];
          }
        }
      }
      const int r = tid >> 4, c0 = (tid & 15) * 8, p = len + r - drop;
      if (r < n_cur && p >= 0) {
        *glob(reinterpret_cast<u32x4*>(ck + (size_t)p * HD + c0)) = *reinterpret_cast<const u32x4*>(smem + KB + (len + r) * KS + c0 * 2);
        *glob(reinterpret_cast<u32x4*>(cv + (size_t)p * HD + c0)) = *reinterpret_cast<const u32x4*>(smem + VB + (len + r) * KS + c0 * 2);
      }
    }
    lds_barrier();
    STAMP(11);
    {   // exchange 3: own 128 ctx columns out, the other three heads' in
      const int row = tid >> 5, off = (tid & 31) * 8;
      put8(reinterpret_cast<unsigned char*>(ctx_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(smem + CTX + row * AS + h * 256 + off));
      publish(flags + 3);
      consume(flags + 3, a.err);
#pragma unroll
      for (int q = 1; q < NH; ++q) {
        const int hq = (h + q) & 3;
        *reinterpret_cast<u64*>(smem + CTX + row * AS + hq * 256 + off) = get8(reinterpret_cast<const unsigned char*>(ctx_rows + (size_t)row * D + hq * HD) + off);
      }
    }
    lds_barrier();
    STAMP(12);
    // ---- 5: out-projection columns 128 h + 16 wave .. + bo + x2 -> the stream's rows of the next layer
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 16, 16, 2, false>(wo, wo, wpo, a_lane + CTX, acc);
      const int col = wave * 16 + frow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = fgrp * 4 + i;
        decs[row * HD + col] = (acc[0][i] + acc[1][i]) + bov + x2s[row * HD + col];
      }
    }
    lds_barrier();
    put_slab_f32(dec_rows, smem + DECS, tid, h);
    publish(flags + 4);
    STAMP(13);
    if (li + 1 < a.n_layers) {
      wload<16>(w1a, a.layers[li + 1].wpack + PK_1 + wave_frag * PW_1 + lane * 16, 0);
      if (!(a.opt & 1)) {
        if (by_head) tw = warm(a.layers[li + 1].wpack + PK_1 + (size_t)h * NW * PW_1, (int)(NW * PW_1 / 128), wg_xcd, n_wg_xcd, tid);
        else tw = warm(a.layers[li + 1].wpack + PK_1, (int)((PK_2 - PK_1) / 128), wg_xcd, n_wg_xcd, tid);
      }
    }
  }
  if (sink == 0x9e3779b9u && a.n_layers < 0) a.err[1] = sink;            // (keeps the warm-up loads; never true)
}

// one thread per 16-byte slot of the packed copy: where in the arena's row-major matrices its eight elements live
__global__ __launch_bounds__(256) void stream_dec_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, const bf16_t* __restrict__ wq,
                                                              const bf16_t* __restrict__ wkv, const bf16_t* __restrict__ wo, int full, unsigned char* __restrict__ dst) {
  const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x, o = slot * 16;
  if (o >= PK_BYTES) return;
  const int lane = (int)((o >> 4) & 63), frow = lane & 15, kq = (lane >> 4) * 8;
  const bf16_t* src = nullptr;
  if (o < PK_2) {
    const size_t r = o - PK_1;
    const int hw = (int)(r / PW_1), frag = (int)((r % PW_1) >> 10), ks = frag >> 2, j = frag & 3, h = hw >> 3, w = hw & 7;
    src = w1 + (size_t)(h * 512 + w * 64 + j * 16 + frow) * D + ks * 32 + kq;
  } else if (o < PK_Q) {
    const size_t r = o - PK_2;
    const int hw = (int)(r / PW_2), ks = (int)((r % PW_2) >> 10), h = hw >> 3, w = hw & 7;
    src = w2 + (size_t)(h * HD + w * 16 + frow) * DFF + ks * 32 + kq;
  } else if (!full) {
    src = nullptr;
  } else if (o < PK_KV) {
    const size_t r = o - PK_Q;
    const int hw = (int)(r / PW_Q), ks = (int)((r % PW_Q) >> 10), h = hw >> 3, w = hw & 7;
    src = wq + (size_t)(h * HD + w * 16 + frow) * D + ks * 32 + kq;
  } else if (o < PK_O) {
    const size_t r = o - PK_KV;
    const int hw = (int)(r / PW_KV), frag = (int)((r % PW_KV) >> 10), ks = frag >> 1, j = frag & 1, h = hw >> 3, w = hw & 7;
    src = wkv + (size_t)(j * D + h * HD + w * 16 + frow) * D + ks * 32 + kq;
  } else {
    const size_t r = o - PK_O;
    const int hw = (int)(r / PW_O), ks = (int)((r % PW_O) >> 10), h = hw >> 3, w = hw & 7;
    src = wo + (size_t)(h * HD + w * 16 + frow) * D + ks * 32 + kq;
  }
  *reinterpret_cast<uint4*>(dst + o) = src ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
}

}  // namespace

size_t stream_dec_pack_bytes() { return PK_BYTES; }

void launch_stream_dec_pack(const bf16_t* w1, const bf16_t* w2, const bf16_t* wq, const bf16_t* wkv, const bf16_t* wo, bool full, void* dst, hipStream_t s) {
  hipLaunchKernelGGL(stream_dec_pack_kernel, dim3((unsigned)((PK_BYTES / 16 + 255) / 256)), dim3(256), 0, s, w1, w2, wq, wkv, wo, full ? 1 : 0, (unsigned char*)dst);
  HIP_CHECK(hipGetLastError());
}

bool stream_dec_supported(int d, int d_ffn, int n_heads, int cap, int n_cur, int ktaps) {
  return d == D && d_ffn == DFF && n_heads == NH && cap + SLOT <= MAXK && n_cur <= SLOT && ktaps == TAPS;
}

void launch_stream_dec(const StreamDecArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_streams >= 1 && a.n_layers >= 1 && a.cap + SLOT <= MAXK && a.n_cur <= SLOT, "stream_dec: bad geometry");
  static PerDeviceOnce attr_once;
  if (attr_once.first())
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_dec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int n_wgs = !(a.opt & 8) ? (a.n_streams + 7) / 8 * 32 : (a.n_streams + 1) / 2 * 8;
  hipLaunchKernelGGL(stream_dec_kernel, dim3(n_wgs), dim3(NT), LDS_BYTES, s, a);
  HIP_CHECK(hipGetLastError());
}
