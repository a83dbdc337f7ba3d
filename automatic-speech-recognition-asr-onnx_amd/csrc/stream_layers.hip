// Paraformer online encoder, layers 1 .. n of ONE chunk step as ONE launch (Paraformer/Streaming/Export_Paraformer_Streaming.py:400-435:
// per layer LayerNorm -> q|k|v -> soft-max attention over [K/V history | chunk rows] + FSMN over the chunk rows -> out-projection + residual
// -> LayerNorm -> FFN + residual, with the history rolled on the way). bf16 sessions, d = 512 / 4 heads of 128 / d_ffn = 2048.
//
// A chunk step is 13 live rows per stream in a 16-row slot: at 64 streams every GEMM of the layer is 1024 x {1536, 512, 2048, 512} against
// 6.3 MB of weights -- weight-streaming work, and seven launches of it per layer spend their time filling and draining the chip (round 3:
// 520 launches, 5.2 ms per step). Here the step's layer loop is one launch of (streams x 4) workgroups:
//
//   * the four workgroups (s, h) of stream s form a CLUSTER; workgroup h owns head h of the attention and column slab h of every GEMM; the
//     cluster meets four times per layer (ctx, x1, hid, x) through memory: agent-scope write-through stores and agent-scope loads of the
//     payload, one relaxed counter per exchange -- no fence, no L2 invalidate. Placement is by HEAD (XCDs 2 h and 2 h + 1 host head h's
//     workgroups): the meetings do not care where a stream's heads sit, and an XCD's L2 then streams a quarter of every matrix;
//   * a slot is ONE 16-row MFMA tile: wave w of 8 multiplies the slot by its own 48 / 16 / 64 / 16 output columns with the whole K in its
//     own registers' stream: weights come straight from a fragment-major copy of the layer (`launch_stream_layers_pack`: one contiguous KB
//     per wave instruction, in consumption order), 12-16 KB per wave in flight, and the first batch of the NEXT phase is requested before the
//     workgroup waits for its cluster -- the exchange latency hides the weight latency and the other way round;
//   * rounding points are those of the per-launch path (bf16 operand of every GEMM, bf16 q|k|v and ctx, f32 residual stream, f32 soft-max
//     and FSMN in LDS) with ONE addition: the soft-max weights enter P V as bf16 MFMA operands here, where stream_attn_kernel multiplies them in f32. The
//     fixtures of the per-launch path hold for this one within the bars of tests/test_paraformer_streaming_gpu.py::test_fused_launches_vs_per_launch_path.
//
// Give-up: a cluster that waits 0.2 s on a counter raises `err` and the launch runs out (results void, histories half-rolled). Round 5: when other sessions exist on the
// GPU the host snapshots the active streams' state in front of the step, restores it and redoes the step on the per-launch path (SvSession::stream_step); without a
// snapshot the step fails and its streams must be reset.
#include <type_traits>
#include "stream_cluster.h"

namespace {

constexpr int D = 512, DFF = 2048, HD = 128, SLOT = 16, MAXK = 64, TAPS = 11;
// ---- fragment-major weight copy of one layer: [phase][head][wave][fragment][64 lanes][16 B]
constexpr size_t PW_A = 16 * 3 * 1024, PW_B = 16 * 1024, PW_C = 16 * 4 * 1024, PW_D = 64 * 1024;
constexpr size_t PK_A = 0, PK_B = PK_A + NH * NW * PW_A, PK_C = PK_B + NH * NW * PW_B, PK_D = PK_C + NH * NW * PW_C, PK_BYTES = PK_D + NH * NW * PW_D;
static_assert(PK_BYTES == (size_t)(3 * D * D + D * D + 2 * DFF * D) * 2, "the packed copy holds every weight element once");
// ---- LDS map (bytes)
constexpr int AS = D * 2 + 16;                 // row stride of a 512-wide bf16 operand (16 B of padding: the 16 rows of a fragment read start in different banks)
constexpr int HS = DFF * 2 + 16;               // row stride of hid
constexpr int XN = 0, CTX = XN + SLOT * AS;    // normalised rows (phases A, C) / attention context (phase B)
constexpr int UNI = CTX + SLOT * AS;           // union: K / V images of the attention (bf16 [64][128], row-major like the cache) | hid [16][2048] bf16
constexpr int KS = HD * 2 + 16;                // row stride of the q / k / v images
constexpr int KB = UNI, VB = KB + MAXK * KS, HID = UNI;
constexpr int UNI_END = HID + SLOT * HS;
static_assert(VB + MAXK * KS <= UNI_END, "the K / V images live inside hid's bytes");
constexpr int QB = UNI_END, SF = QB + SLOT * KS;                   // q rows bf16, scores f32 [16][65]
constexpr int PS = MAXK * 2 + 16, PB = SF + SLOT * (MAXK + 1) * 4; // probabilities bf16 [16][64]
constexpr int XRES = (PB + SLOT * PS + 15) / 16 * 16;                   // own 128 columns of the residual stream, f32 [16][128]
constexpr int XB = XRES + SLOT * HD * 4, MEM = XB + SLOT * HD * 4, LDS_BYTES = MEM + SLOT * HD * 4;
static_assert(LDS_BYTES <= 160 * 1024 && XRES % 16 == 0 && UNI % 16 == 0, "LDS map");

// full rows of the cluster's f32 stream (16 x 512, exchanged) -> plain normalisation (the affine is folded into the next weights) -> bf16 operand rows;
// thread = (row, 16 columns); the own 128 columns are kept in f32 at `keep` (the residual of the next epilogue) when asked
template <typename F>
__device__ __forceinline__ void norm_rows(const float* src, unsigned char* smem, int tid, int h, bool keep_own, float eps, F&& behind_loads) {
  // thread = (row, column pairs 64 e + 2 j, e < 8): a wave instruction reads 256 contiguous bytes of each of its two rows (16 columns per thread in a row would be
  // 64 lanes 64 bytes apart, one 64-byte segment each: 2 us of address processing per LayerNorm)
  const int row = tid >> 5, j = tid & 31;
  const float* p = src + (size_t)row * D + 2 * j;
  float v[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const u64 t = get8(p + 64 * e);
    v[2 * e] = __uint_as_float((unsigned)t);
    v[2 * e + 1] = __uint_as_float((unsigned)(t >> 32));
  }
  __builtin_amdgcn_sched_barrier(0);
  behind_loads();                                          // (more requests for the queue: they line up behind the rows, the arithmetic below waits for the rows only)
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; e += 4) s += (v[e] + v[e + 1]) + (v[e + 2] + v[e + 3]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / D);
  float q = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    const float a = v[e] - mean, b = v[e + 1] - mean, c = v[e + 2] - mean, d = v[e + 3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    *reinterpret_cast<unsigned*>(smem + XN + row * AS + (64 * e + 2 * j) * 2) = pack_bf16x2((v[2 * e] - mean) * rstd, (v[2 * e + 1] - mean) * rstd);
    if (keep_own && (e >> 1) == h) *reinterpret_cast<float2*>(smem + XRES + (row * HD + 64 * (e & 1) + 2 * j) * 4) = make_float2(v[2 * e], v[2 * e + 1]);
  }
}

// own [16][128] f32 slab in LDS -> the cluster's [rows][512] f32 buffer (two 8-byte write-through stores per thread)
__device__ __forceinline__ void put_slab_f32(float* dst_rows, const unsigned char* slab, int tid, int h) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + NT * e, row = idx >> 6, off = (idx & 63) * 8;
    put8(reinterpret_cast<unsigned char*>(dst_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(slab + row * HD * 4 + off));
  }
}

#define STAMP(k) do { if (a.times && li == a.times_layer && threadIdx.x == 0) a.times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

template <int PFM>          // phases whose second weight batch is requested right behind the phase's exchanged rows (under their arithmetic): bit 0 = A, 1 = C, 2 = D
__global__ __launch_bounds__(NT) void stream_layers_kernel(const StreamLayersArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid_0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid_0 >> 6);
  // cluster placement: workgroup b runs on XCD b % 8 (observed; nothing but speed depends on it)
  // (round 4, late: placement BY HEAD instead -- XCDs 2 h and 2 h + 1 host the workgroups of head h, so an XCD's L2 streams a quarter of every matrix; the
  //  meetings go through the memory side and do not care where a stream's four heads sit; a stream's workgroups stay within one group of eight ids.
  //  a.opt & 8: the old placement, a stream's heads on one XCD)
  int cl, h;
  if (a.opt & 8) { const int idx = blockIdx.x >> 3; cl = ((idx >> 2) << 3) + (blockIdx.x & 7); h = idx & 3; }
  else { const int x = blockIdx.x & 7; h = x >> 1; cl = (blockIdx.x >> 3) * 2 + (x & 1); }
  if (cl >= a.n_streams) return;
  const UttPlan up = a.plan[cl];
  const int sid = up.lang, row0 = up.row_off, n_cur = a.n_cur;
  const int len = __builtin_amdgcn_readfirstlane(a.cache_len[sid]), nk = len + n_cur;
  float* Sf = reinterpret_cast<float*>(smem + SF);
  float* xres = reinterpret_cast<float*>(smem + XRES);
  float* xbs = reinterpret_cast<float*>(smem + XB);
  float* mem = reinterpret_cast<float*>(smem + MEM);
  float* x_rows = a.x + (size_t)row0 * D;
  float* xb_rows = a.xb + (size_t)row0 * D;
  bf16_t* ctx_rows = a.ctx + (size_t)row0 * D;
  bf16_t* hid_rows = a.hid + (size_t)row0 * DFF;
  const size_t wave_frag = (size_t)(h * NW + wave);
  u32x4 wa[12];                                                            // batch 0 of phase A: requested a layer ahead
  wload<12>(wa, a.layers[0].wpack + PK_A + wave_frag * PW_A + (tid_0 & 63) * 16, 0);
  // this workgroup's place among the workgroups of its XCD (workgroup b runs on XCD b % 8): its share of every L2 warm-up
  const bool by_head = !(a.opt & 8);
  const int xcd = blockIdx.x & 7;
  const int n_wg_xcd = by_head ? (a.n_streams - (xcd & 1) + 1) >> 1 : ((a.n_streams - xcd + 7) >> 3) * NH, wg_xcd = by_head ? cl >> 1 : (cl >> 3) * NH + h;
  unsigned sink = 0, tw = 0, twb = 0;

#pragma unroll 1
  for (int li = 0; li < a.n_layers; ++li) {
    const StreamLayer& L = a.layers[li];
    const int tid = opaque(tid_0), lane = tid & 63, frow = lane & 15, fgrp = lane >> 4;      // (per layer: nothing per-lane is carried around the loop)
    const unsigned char* a_lane = smem + frow * AS + fgrp * 16;             // this lane's A-fragment bytes inside a 512-wide operand
    const unsigned char* hid_lane = smem + HID + frow * HS + fgrp * 16;
    STAMP(0);
    unsigned* flags = a.flags + ((size_t)li * a.n_streams + cl) * 4;
    const unsigned char* wpA = L.wpack + PK_A + wave_frag * PW_A + lane * 16;
    const unsigned char* wpB = L.wpack + PK_B + wave_frag * PW_B + lane * 16;
    const unsigned char* wpC = L.wpack + PK_C + wave_frag * PW_C + lane * 16;
    const unsigned char* wpD = L.wpack + PK_D + wave_frag * PW_D + lane * 16;
    // ---- K / V history of (stream, head) requested now (nothing on the way depends on it): 16 threads per row, two rows of each per thread
    const bf16_t* ck = L.cache_k + ((size_t)sid * NH + h) * a.cap * HD;
    const bf16_t* cv = L.cache_v + ((size_t)sid * NH + h) * a.cap * HD;
    u32x4 rk[2], rv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
      rk[it] = u32x4{0, 0, 0, 0}; rv[it] = u32x4{0, 0, 0, 0};
      if (p < len) {
        rk[it] = *glob(reinterpret_cast<const u32x4*>(ck + (size_t)p * HD + c0));
        rv[it] = *glob(reinterpret_cast<const u32x4*>(cv + (size_t)p * HD + c0));
      }
    }
    // epilogue constants of the first half: requested here, a layer's worth of latency ahead of their use (in the epilogue each would cost a round trip)
    float bA[3], wc[TAPS], bc;
    {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int lc = wave * 48 + j * 16;
        bA[j] = glob(L.bqkv)[(lc >> 7) * D + h * HD + (lc & 127) + frow];
      }
      const int hc = h * HD + (tid & 127);
#pragma unroll
      for (int j = 0; j < TAPS; ++j) wc[j] = glob(L.wfsmn)[hc * TAPS + j];
      bc = glob(L.bfsmn)[hc];
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase A: LayerNorm of the stream's rows (the previous layer's exchange 3, or the rows the launch was given), q|k|v of head h
    if (li > 0) consume(flags - (size_t)a.n_streams * 4 + 3, a.err);
    STAMP(1);
    u32x4 wa1[12];
    norm_rows(x_rows, smem, tid, h, true, a.ln_eps,
              [&]() __attribute__((always_inline)) { if constexpr ((PFM & 1) != 0) wload<12>(wa1, wpA, 12); });     // batch 1 of phase A under the LayerNorm
#pragma unroll
    for (int it = 0; it < 2; ++it) {        // history rows -> the K / V images (row-major bf16, as they are cached); rows past the chunk's slot are zero for P V
      const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
      if (p < len) {
        *reinterpret_cast<u32x4*>(smem + KB + p * KS + c0 * 2) = rk[it];
        *reinterpret_cast<u32x4*>(smem + VB + p * KS + c0 * 2) = rv[it];
      } else if (p >= len + SLOT) {
        *reinterpret_cast<u32x4*>(smem + VB + p * KS + c0 * 2) = u32x4{0, 0, 0, 0};
      }
    }
    lds_barrier();
    STAMP(2);
    {
      f32x4_t acc[3] = {};
      gemm_phase<3, 16, 4, 1, (PFM & 1) != 0>(wa, wa1, wpA, a_lane + XN, acc);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int lc = wave * 48 + j * 16, part = lc >> 7, within = (lc & 127) + frow;          // (a 16-column tile never straddles q | k | v)
        const float bias = bA[j];
        const int base = part == 0 ? QB : part == 1 ? KB : VB, r0 = part == 0 ? 0 : len;           // (wave-uniform selects)
#pragma unroll
        for (int i = 0; i < 4; ++i)                                                                // q|k|v are bf16 tensors on the per-launch path too
          *reinterpret_cast<bf16_t*>(smem + base + (r0 + fgrp * 4 + i) * KS + within * 2) = (bf16_t)(pack_bf16x2(acc[j][i] + bias, 0.0f) & 0xffffu);
      }
    }
    u32x4 wb[16];
    wload<16>(wb, wpB, 0);                                                  // the whole of phase B, under the attention
    lds_barrier();
    STAMP(3);
    float bC[4], bD;                                          // epilogue constants of the second half, in front of the warm-up
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) bC[j] = glob(L.b1)[h * 512 + wave * 64 + j * 16 + frow];
      bD = glob(L.b2)[h * HD + wave * 16 + frow];
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned tw2 = 0;
    unsigned tw2b = 0;
    if (!(a.opt & 3)) {                                      // both FFN matrices (by-head placement: head h's slices), under the attention (nothing in it waits on the vector queue)
      if (by_head) {
        tw2 = warm(L.wpack + PK_C + (size_t)h * NW * PW_C, (int)(NW * PW_C / 128), wg_xcd, n_wg_xcd, tid);
        tw2b = warm(L.wpack + PK_D + (size_t)h * NW * PW_D, (int)(NW * PW_D / 128), wg_xcd, n_wg_xcd, tid);
      } else tw2 = warm(L.wpack + PK_C, (int)((PK_BYTES - PK_C) / 128), wg_xcd, n_wg_xcd, tid);
    }
    // ---- attention of the slot's 16 rows over [history | chunk rows]: scores on the matrix pipe (wave = one 16-key tile), soft-max in f32 (wave = two rows),
    //      P (bf16) V on the matrix pipe (wave = 16 channels; the V fragment is gathered down the key axis of the row-major image)
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
        *reinterpret_cast<bf16_t*>(smem + CTX + q * AS + (h * HD + wave * 16 + frow) * 2) = q < n_cur ? (bf16_t)(pack_bf16x2(o[i], 0.0f) & 0xffffu) : (bf16_t)0;
      }
    }
    {   // FSMN memory term of the slot (taps outside the chunk rows are zero, rows past the chunk are zero), the head's 128 channels; thread = (channel, 4 rows)
      const int c = tid & 127, t0 = (tid >> 7) * 4;
      constexpr int PAD = (TAPS - 1) / 2, NV = 4 + TAPS - 1;
      float vr[NV];                                        // rows t0 - PAD .. t0 + 3 + PAD of the chunk's V (zero outside the chunk)
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int tt = t0 + r - PAD;
        vr[r] = (tt >= 0 && tt < n_cur) ? bf16_to_f32(*reinterpret_cast<const bf16_t*>(smem + VB + (len + tt) * KS + c * 2)) : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float m = bc;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) m = fmaf(wc[j], vr[i + j], m);
        mem[(t0 + i) * HD + c] = t0 + i < n_cur ? m : 0.0f;
      }
    }
    {   // history <- last cap of (history ++ the first roll_rows chunk rows): old rows move inside the cache from the registers they were read into,
        // the new rows come from the images
      const int total = len + a.roll_rows, new_len = min(total, a.cap), drop = total - new_len;
      bf16_t* wk = const_cast<bf16_t*>(ck);
      bf16_t* wv = const_cast<bf16_t*>(cv);
      if (drop > 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int s = tid + it * NT, p = s >> 4, c0 = (s & 15) * 8;
          if (p < len && p >= drop) {
            *glob(reinterpret_cast<u32x4*>(wk + (size_t)(p - drop) * HD + c0)) = rk[it];
            *glob(reinterpret_cast<u32x4*>(wv + (size_t)(p - drop) * HD + c0)) = rv[it];
          }
        }
      }
      {
        const int r = tid >> 4, c0 = (tid & 15) * 8, p = len + r - drop;          // chunk row r -> history row p
        if (r < a.roll_rows && p >= 0) {
          *glob(reinterpret_cast<u32x4*>(wk + (size_t)p * HD + c0)) = *reinterpret_cast<const u32x4*>(smem + KB + (len + r) * KS + c0 * 2);
          *glob(reinterpret_cast<u32x4*>(wv + (size_t)p * HD + c0)) = *reinterpret_cast<const u32x4*>(smem + VB + (len + r) * KS + c0 * 2);
        }
      }
    }
    lds_barrier();
    STAMP(4);
    {   // exchange 0: own 128 ctx columns out, the other three heads' in
      const int row = tid >> 5, off = (tid & 31) * 8;
      put8(reinterpret_cast<unsigned char*>(ctx_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(smem + CTX + row * AS + h * 256 + off));
      publish(flags + 0, a.fault != 0 && li == 0 && blockIdx.x == 5);
      consume(flags + 0, a.err);
#pragma unroll
      for (int q = 1; q < NH; ++q) {
        const int hq = (h + q) & 3;
        *reinterpret_cast<u64*>(smem + CTX + row * AS + hq * 256 + off) = get8(reinterpret_cast<const unsigned char*>(ctx_rows + (size_t)row * D + hq * HD) + off);
      }
    }
    lds_barrier();
    STAMP(5);
    // ---- phase B: out-projection columns 128 h + 16 wave .., + FSMN term + residual -> x1 slab
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 16, 16, 2, false>(wb, wb, wpB, a_lane + CTX, acc);
      const int col = wave * 16 + frow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = fgrp * 4 + i;
        xbs[row * HD + col] = (acc[0][i] + acc[1][i]) + mem[row * HD + col] + xres[row * HD + col];
      }
    }
    lds_barrier();
    STAMP(6);
    put_slab_f32(xb_rows, smem + XB, tid, h);
    publish(flags + 1);
    sink ^= tw ^ twb ^ tw2 ^ tw2b;                                               // (drained by the publish)
    u32x4 wc0[16], wc1[16];
    wload<16>(wc0, wpC, 0);
    if ((a.opt & 3) == 2) tw = warm(L.wpack + PK_C, (int)((PK_D - PK_C) / 128), wg_xcd, n_wg_xcd, tid);
    consume(flags + 1, a.err);
    STAMP(7);
    // ---- phase C: LayerNorm of x1, FFN-1 columns 512 h + 64 wave ..
    norm_rows(xb_rows, smem, tid, h, false, a.ln_eps, [&]() __attribute__((always_inline)) { if constexpr ((PFM & 2) != 0) wload<16>(wc1, wpC, 16); });
    lds_barrier();
    STAMP(8);
    {
      f32x4_t acc[4] = {};
      gemm_phase<4, 16, 4, 1, (PFM & 2) != 0>(wc0, wc1, wpC, a_lane + XN, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = h * 512 + wave * 64 + j * 16 + frow;
        const float bias = bC[j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<bf16_t*>(smem + HID + (fgrp * 4 + i) * HS + col * 2) = (bf16_t)(pack_bf16x2(fmaxf(acc[j][i] + bias, 0.0f), 0.0f) & 0xffffu);
      }
    }
    lds_barrier();
    STAMP(9);
    u32x4 wd[16], wd1[16];
    {   // exchange 2: own 512 hid columns out, the other three quarters in
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = tid + NT * e, row = s >> 7, off = (s & 127) * 8;
        put8(reinterpret_cast<unsigned char*>(hid_rows + (size_t)row * DFF + h * 512) + off, *reinterpret_cast<const u64*>(smem + HID + row * HS + h * 1024 + off));
      }
      publish(flags + 2);
      sink ^= tw;
      wload<16>(wd, wpD, 0);
      if ((a.opt & 3) == 2) tw = warm(L.wpack + PK_D, (int)((PK_BYTES - PK_D) / 128), wg_xcd, n_wg_xcd, tid);
      consume(flags + 2, a.err);
#pragma unroll
      for (int q = 1; q < NH; ++q) {
        const int hq = (h + q) & 3;
        u64 t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = tid + NT * e, row = s >> 7, off = (s & 127) * 8;
          t[e] = get8(reinterpret_cast<const unsigned char*>(hid_rows + (size_t)row * DFF + hq * 512) + off);
        }
        if constexpr ((PFM & 4) != 0) { if (q == NH - 1) { __builtin_amdgcn_sched_barrier(0); wload<16>(wd1, wpD, 16); } }      // batch 1 of phase D behind the last quarter
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = tid + NT * e, row = s >> 7, off = (s & 127) * 8;
          *reinterpret_cast<u64*>(smem + HID + row * HS + hq * 1024 + off) = t[e];
        }
      }
    }
    lds_barrier();
    STAMP(10);
    // ---- phase D: FFN-2 columns 128 h + 16 wave .. + b2 + x1 -> the stream's rows of the next layer
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 64, 16, 2, (PFM & 4) != 0>(wd, wd1, wpD, hid_lane, acc);
      const int col = wave * 16 + frow;
      const float bias = bD;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = fgrp * 4 + i;
        xres[row * HD + col] = (acc[0][i] + acc[1][i]) + bias + xbs[row * HD + col];
      }
    }
    lds_barrier();
    STAMP(11);
    put_slab_f32(x_rows, smem + XRES, tid, h);
    publish(flags + 3);
    sink ^= tw;
    STAMP(12);
    if (li + 1 < a.n_layers) {
      wload<12>(wa, a.layers[li + 1].wpack + PK_A + wave_frag * PW_A + lane * 16, 0);
      if (!(a.opt & 1)) {                                    // q|k|v and out-projection of the next layer
        const unsigned char* nw = a.layers[li + 1].wpack;
        if (by_head) {
          tw = warm(nw + PK_A + (size_t)h * NW * PW_A, (int)(NW * PW_A / 128), wg_xcd, n_wg_xcd, tid);
          twb = warm(nw + PK_B + (size_t)h * NW * PW_B, (int)(NW * PW_B / 128), wg_xcd, n_wg_xcd, tid);
        } else tw = warm(nw + PK_A, (int)((PK_C - PK_A) / 128), wg_xcd, n_wg_xcd, tid);
      }
    }
  }
  if (sink == 0x9e3779b9u && a.n_layers < 0) a.err[1] = sink;            // (keeps the warm-up loads; never true)
}

// one thread per 16-byte slot of the packed copy: where in the arena's row-major matrices its eight elements live
__global__ __launch_bounds__(256) void stream_layers_pack_kernel(const bf16_t* __restrict__ wqkv, const bf16_t* __restrict__ wout, const bf16_t* __restrict__ w1,
                                                                 const bf16_t* __restrict__ w2, unsigned char* __restrict__ dst) {
  const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x, o = slot * 16;
  if (o >= PK_BYTES) return;
  const int lane = (int)((o >> 4) & 63), frow = lane & 15, kq = (lane >> 4) * 8;
  const bf16_t* src;
  if (o < PK_B) {
    const size_t r = o - PK_A;
    const int hw = (int)(r / PW_A), frag = (int)((r % PW_A) >> 10), ks = frag / 3, j = frag % 3, h = hw >> 3, w = hw & 7;
    const int lc = w * 48 + j * 16 + frow;
    src = wqkv + (size_t)((lc >> 7) * D + h * HD + (lc & 127)) * D + ks * 32 + kq;
  } else if (o < PK_C) {
    const size_t r = o - PK_B;
    const int hw = (int)(r / PW_B), ks = (int)((r % PW_B) >> 10), h = hw >> 3, w = hw & 7;
    src = wout + (size_t)(h * HD + w * 16 + frow) * D + ks * 32 + kq;
  } else if (o < PK_D) {
    const size_t r = o - PK_C;
    const int hw = (int)(r / PW_C), frag = (int)((r % PW_C) >> 10), ks = frag >> 2, j = frag & 3, h = hw >> 3, w = hw & 7;
    src = w1 + (size_t)(h * 512 + w * 64 + j * 16 + frow) * D + ks * 32 + kq;
  } else {
    const size_t r = o - PK_D;
    const int hw = (int)(r / PW_D), ks = (int)((r % PW_D) >> 10), h = hw >> 3, w = hw & 7;
    src = w2 + (size_t)(h * HD + w * 16 + frow) * DFF + ks * 32 + kq;
  }
  *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src);
}

}  // namespace

template <int M> static void static_for_attr() {          // every instance may use the large dynamic LDS size (set once per device)
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_layers_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  if constexpr (M < 7) static_for_attr<M + 1>();
}

size_t stream_layers_pack_bytes() { return PK_BYTES; }

void launch_stream_layers_pack(const bf16_t* wqkv, const bf16_t* wout, const bf16_t* w1, const bf16_t* w2, void* dst, hipStream_t s) {
  hipLaunchKernelGGL(stream_layers_pack_kernel, dim3((unsigned)((PK_BYTES / 16 + 255) / 256)), dim3(256), 0, s, wqkv, wout, w1, w2, (unsigned char*)dst);
  HIP_CHECK(hipGetLastError());
}

bool stream_layers_supported(int d, int d_ffn, int n_heads, int cap, int n_cur, int ktaps) {
  return d == D && d_ffn == DFF && n_heads == NH && cap + SLOT <= MAXK && n_cur <= SLOT && ktaps == TAPS;
}

void launch_stream_layers(const StreamLayersArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_streams >= 1 && a.n_layers >= 1 && a.cap + SLOT <= MAXK && a.n_cur <= SLOT && a.ktaps == TAPS, "stream_layers: bad geometry");
  static PerDeviceOnce attr_once;
  const bool first = attr_once.first();
  const int n_wgs = (a.opt & 8) ? (a.n_streams + 7) / 8 * 32 : (a.n_streams + 1) / 2 * 8;
  const int pfm = 7 & (a.opt >> 4);
  auto go = [&](auto tag) {
    constexpr int PFM = decltype(tag)::value;
    if (first) {
      static_for_attr<0>();
    }
    hipLaunchKernelGGL(stream_layers_kernel<PFM>, dim3(n_wgs), dim3(NT), LDS_BYTES, s, a);
  };
  switch (pfm) {
    case 0: go(std::integral_constant<int, 0>{}); break;
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 3: go(std::integral_constant<int, 3>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    case 5: go(std::integral_constant<int, 5>{}); break;
    case 6: go(std::integral_constant<int, 6>{}); break;
    default: go(std::integral_constant<int, 7>{}); break;
  }
  HIP_CHECK(hipGetLastError());
}
