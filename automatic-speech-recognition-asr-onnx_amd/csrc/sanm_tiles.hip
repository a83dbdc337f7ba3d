// SANM encoder blocks (Export_SenseVoice.py:227-258; the same block in Export_Paraformer.py) for SMALL batches of <= 144-row windows, a run of blocks as ONE
// launch: the batch-1 point of the metric. The batch block kernel (sanm_block8.hip) gives a window four workgroups -- right for 64 windows, four CUs of 256
// for one. Here a window is cut into its 16-row MFMA tiles and every (tile, head) is a workgroup of the streaming machine (stream_layers.hip, helpers in
// stream_cluster.h): 9 tiles x 4 heads = 36 workgroups per 8 s window, each streaming a quarter of the block's weights from the fragment-major copy straight
// into registers, the four heads of a tile meeting four times per block (ctx, x1, hid, x) through memory.
//
// What a window needs beyond a stream's slot: soft-max attention and the FSMN run over the WHOLE window, so the nine workgroups of a head meet once more per
// block -- every one writes the k and v rows of its tile, waits for the window's count and reads the other tiles' rows into its LDS images ([144][128] k,
// [160][128] v, row-major bf16). Scores are 16 queries x <= 9 key tiles on the matrix pipe, the soft-max runs in f32 over <= 144 keys (three per lane), P V
// over <= 5 k-steps with the V fragment gathered down the key axis; keys and FSMN taps past the window's T rows are masked by index (pad rows hold finite
// values, as everywhere in the engine).
//
// Rounding points are those of the four-launch path in its un-fused form (separate LayerNorm, bf16 q|k|v / ctx / hid, f32 residual stream), which is what a
// batch below the block kernel's threshold ran before. A workgroup that waits 0.2 s raises `err`: the host redoes the pass on the four-launch path.
// Tried and dropped: 16 extra workgroups (two per XCD) that do nothing but touch the next phase's weights one phase ahead of tile cluster 0 -- with 220 idle CUs
// at one window it looked free, but the GEMM phases got 15 % SLOWER (5.4 / 5.5 / 6.2 us against 4.6 / 4.8 / 5.7 with every workgroup touching its 1 / n).
#include "stream_cluster.h"
#include "gemm.h"

namespace {

constexpr int D = 512, DFF = 2048, HD = 128, SLOT = 16, MAXT = 144, MAXKT = MAXT / 16, VROWS = 160, TAPS = 11;
// ---- fragment-major weight copy of one block: the streaming encoder's (launch_stream_layers_pack), [phase][head][wave][fragment][64 lanes][16 B]
constexpr size_t PW_A = 16 * 3 * 1024, PW_B = 16 * 1024, PW_C = 16 * 4 * 1024, PW_D = 64 * 1024;
constexpr size_t PK_A = 0, PK_B = PK_A + NH * NW * PW_A, PK_C = PK_B + NH * NW * PW_B, PK_D = PK_C + NH * NW * PW_C, PK_BYTES = PK_D + NH * NW * PW_D;
// ---- LDS map (bytes)
constexpr int AS = D * 2 + 16, HS = DFF * 2 + 16, KS = HD * 2 + 16;
constexpr int XN = 0, CTX = XN + SLOT * AS;
constexpr int UNI = CTX + SLOT * AS;           // union: k / v images of the window (bf16 [144][128], [160][128]) | hid [16][2048] bf16
constexpr int KB = UNI, VB = KB + MAXT * KS, HID = UNI;
constexpr int UNI_END = (VB + VROWS * KS > HID + SLOT * HS) ? VB + VROWS * KS : HID + SLOT * HS;
constexpr int QB = UNI_END, SF = QB + SLOT * KS;                   // q rows bf16, scores f32 [16][161]
constexpr int SFS = VROWS + 1, PS = VROWS * 2 + 16, PB = SF + SLOT * SFS * 4;      // probabilities bf16 [16][160]
constexpr int XRES = (PB + SLOT * PS + 15) / 16 * 16;
constexpr int XB = XRES + SLOT * HD * 4, MEM = XB + SLOT * HD * 4, LDS_BYTES = MEM + SLOT * HD * 4;
static_assert(LDS_BYTES <= 160 * 1024 && XRES % 16 == 0 && UNI % 16 == 0, "LDS map");

// full rows of the tile's f32 stream (16 x 512, exchanged) -> plain normalisation (the affine is folded into the next weights) -> bf16 operand rows
__device__ __forceinline__ void norm_rows(const float* src, unsigned char* smem, int tid, int h, bool keep_own, float eps) {
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

__device__ __forceinline__ void put_slab_f32(float* dst_rows, const unsigned char* slab, int tid, int h) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + NT * e, row = idx >> 6, off = (idx & 63) * 8;
    put8(reinterpret_cast<unsigned char*>(dst_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(slab + row * HD * 4 + off));
  }
}

#define STAMP(k) do { if (a.times && li == a.times_layer && threadIdx.x == 0) a.times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(NT) void sanm_tiles_kernel(const SanmTilesArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid_0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid_0 >> 6);
  // Placement BY HEAD (workgroup b runs on XCD b % 8): XCDs 2 h and 2 h + 1 host the workgroups of head h, so an XCD's L2 streams a QUARTER of every matrix
  // (1.57 MB per block instead of 6.3 MB) -- the meetings go through the memory side anyway, they do not care where a tile's four heads sit. The four
  // workgroups of a tile are within one group of eight consecutive ids: dispatched together. (a.opt & 2: the older placement, a tile's heads on one XCD.)
  int cl, h;
  if (a.opt & 2) { const int idx = blockIdx.x >> 3; cl = ((idx >> 2) << 3) + (blockIdx.x & 7); h = idx & 3; }
  else { const int x = blockIdx.x & 7; h = x >> 1; cl = (blockIdx.x >> 3) * 2 + (x & 1); }
  if (cl >= a.n_tiles) return;
  const int win = a.tile_win[cl], tile = a.tile_idx[cl];
  const UttPlan up = a.plan[win];
  const int T = __builtin_amdgcn_readfirstlane(up.T), wrow0 = up.row_off, row0 = wrow0 + tile * SLOT, nkt = (T + SLOT - 1) / SLOT;
  float* Sf = reinterpret_cast<float*>(smem + SF);
  float* xres = reinterpret_cast<float*>(smem + XRES);
  float* xbs = reinterpret_cast<float*>(smem + XB);
  float* mem = reinterpret_cast<float*>(smem + MEM);
  float* x_rows = a.x + (size_t)row0 * D;
  float* xb_rows = a.xb + (size_t)row0 * D;
  bf16_t* ctx_rows = a.ctx + (size_t)row0 * D;
  bf16_t* hid_rows = a.hid + (size_t)row0 * DFF;
  bf16_t* kv_win0 = a.kv + (size_t)wrow0 * 2 * D;                          // [block parity][window row][k | v][512]
  const size_t wave_frag = (size_t)(h * NW + wave);
  u32x4 wa[12];
  wload<12>(wa, a.layers[0].wpack + PK_A + wave_frag * PW_A + (tid_0 & 63) * 16, 0);
  // this workgroup's share of an L2 warm-up: by-head placement -> the XCD needs head h's slice of a phase only, shared by the tiles of its parity
  const bool by_head = !(a.opt & 2);
  const int xcd = blockIdx.x & 7;
  const int n_wg_xcd = by_head ? (a.n_tiles - (xcd & 1) + 1) >> 1 : ((a.n_tiles - xcd + 7) >> 3) * NH, wg_xcd = by_head ? cl >> 1 : (cl >> 3) * NH + h;
  unsigned sink = 0, tw = 0, twb = 0;          // (warm-up values: consumed only behind a drain of the vector queue)
  for (int e = tid_0; e < (VROWS - MAXT) * (KS / 16); e += NT)              // v rows past the last tile: zero once (P is zero there, the product must be too)
    *reinterpret_cast<u32x4*>(smem + VB + MAXT * KS + e * 16) = u32x4{0, 0, 0, 0};

#pragma unroll 1
  for (int li = 0; li < a.n_layers; ++li) {
    const StreamLayer& L = a.layers[li];
    const int tid = opaque(tid_0), lane = tid & 63, frow = lane & 15, fgrp = lane >> 4;
    const unsigned char* a_lane = smem + frow * AS + fgrp * 16;
    const unsigned char* hid_lane = smem + HID + frow * HS + fgrp * 16;
    STAMP(0);
    unsigned* lflags = a.flags + (size_t)li * a.flag_stride;
    unsigned* flags = lflags + cl * 4;
    unsigned* kvflag = lflags + a.n_tiles * 4 + win * NH + h;
    // two k/v buffers, by block parity: a tile that is through block li must not write block li + 1's rows over what a slower tile of its head still reads
    // (it cannot be two blocks ahead: the next meeting needs the slow tile's count)
    bf16_t* kv_win = kv_win0 + (size_t)(li & 1) * a.kv_parity_stride;
    const unsigned char* wpA = L.wpack + PK_A + wave_frag * PW_A + lane * 16;
    const unsigned char* wpB = L.wpack + PK_B + wave_frag * PW_B + lane * 16;
    const unsigned char* wpC = L.wpack + PK_C + wave_frag * PW_C + lane * 16;
    const unsigned char* wpD = L.wpack + PK_D + wave_frag * PW_D + lane * 16;
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
    // ---- phase A: LayerNorm of the tile's rows, q|k|v of head h
    if (li > 0) consume(flags - a.flag_stride + 3, a.err);
    STAMP(1);
    norm_rows(x_rows, smem, tid, h, true, a.ln_eps);
    lds_barrier();
    STAMP(2);
    {
      f32x4_t acc[3] = {};
      u32x4 wa1[12];
      gemm_phase<3, 16, 4, 1, false>(wa, wa1, wpA, a_lane + XN, acc);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int lc = wave * 48 + j * 16, part = lc >> 7, within = (lc & 127) + frow;
        const int base = part == 0 ? QB : part == 1 ? KB : VB, r0 = part == 0 ? 0 : tile * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<bf16_t*>(smem + base + (r0 + fgrp * 4 + i) * KS + within * 2) = (bf16_t)(pack_bf16x2(acc[j][i] + bA[j], 0.0f) & 0xffffu);
      }
    }
    u32x4 wb[16];
    wload<16>(wb, wpB, 0);
    lds_barrier();
    STAMP(3);
    float bC[4], bD;
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) bC[j] = glob(L.b1)[h * 512 + wave * 64 + j * 16 + frow];
      bD = glob(L.b2)[h * HD + wave * 16 + frow];
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // the window's meeting per head: own k / v rows out, the other tiles' in
      const int row = tid >> 5, off = (tid & 31) * 8;
      unsigned char* gk = reinterpret_cast<unsigned char*>(kv_win + (size_t)(tile * SLOT + row) * 2 * D + h * HD) + off;
      put8(gk, *reinterpret_cast<const u64*>(smem + KB + (tile * SLOT + row) * KS + off));
      put8(gk + D * 2, *reinterpret_cast<const u64*>(smem + VB + (tile * SLOT + row) * KS + off));
      publish(kvflag);
      if (nkt & 1) {                                         // P V walks 32 keys at a time: the odd last tile's partner rows must be zero, not what hid left there
        const int s = tid;                                   // 16 rows x 17 16-byte pieces
        if (s < SLOT * (KS / 16)) *reinterpret_cast<u32x4*>(smem + VB + nkt * SLOT * KS + s * 16) = u32x4{0, 0, 0, 0};
      }
      consume(kvflag, a.err, (unsigned)nkt);
      u64 kk[MAXKT], vv[MAXKT];                              // every foreign tile's words in flight at once (a loop of load -> store pairs is a round trip per tile)
#pragma unroll
      for (int tt = 0; tt < MAXKT; ++tt) {
        kk[tt] = 0; vv[tt] = 0;
        if (tt < nkt && tt != tile) {
          const unsigned char* sk = reinterpret_cast<const unsigned char*>(kv_win + (size_t)(tt * SLOT + row) * 2 * D + h * HD) + off;
          kk[tt] = get8(sk); vv[tt] = get8(sk + D * 2);
        }
      }
#pragma unroll
      for (int tt = 0; tt < MAXKT; ++tt) {
        if (tt < nkt && tt != tile) {
          *reinterpret_cast<u64*>(smem + KB + (tt * SLOT + row) * KS + off) = kk[tt];
          *reinterpret_cast<u64*>(smem + VB + (tt * SLOT + row) * KS + off) = vv[tt];
        }
      }
    }
    unsigned tw2 = 0, tw2b = 0;
    if (!(a.opt & 1)) {
      if (by_head) {                                         // head h's slices of FFN-1 and FFN-2
        tw2 = warm(L.wpack + PK_C + (size_t)h * NW * PW_C, (int)(NW * PW_C / 128), wg_xcd, n_wg_xcd, tid);
        tw2b = warm(L.wpack + PK_D + (size_t)h * NW * PW_D, (int)(NW * PW_D / 128), wg_xcd, n_wg_xcd, tid);
      } else tw2 = warm(L.wpack + PK_C, (int)((PK_BYTES - PK_C) / 128), wg_xcd, n_wg_xcd, tid);
    }     // both FFN matrices, under the attention (nothing in it waits on the vector queue)
    lds_barrier();
    STAMP(4);
    // ---- attention of the tile's 16 rows over the window: scores on the matrix pipe (wave = key tiles w, w + 8), soft-max in f32, P V on the matrix pipe
    for (int kt = wave; kt < nkt; kt += NW) {
      f32x4_t sc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks)
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(smem + QB + frow * KS + fgrp * 16 + ks * 64),
                                                     *reinterpret_cast<const bf16x8_t*>(smem + KB + (kt * 16 + frow) * KS + fgrp * 16 + ks * 64), sc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) Sf[(fgrp * 4 + i) * SFS + kt * 16 + frow] = sc[i];
    }
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = wave + 8 * e;
      float sc[3], mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = lane + 64 * c;
        sc[c] = k < T ? Sf[q * SFS + k] : -INFINITY;
        mx = fmaxf(mx, sc[c]);
      }
      mx = wave_max(mx);
      float ex[3], sum = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { ex[c] = lane + 64 * c < T ? expf(sc[c] - mx) : 0.0f; sum += ex[c]; }
      sum = wave_sum(sum);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = lane + 64 * c;
        if (k < VROWS) *reinterpret_cast<bf16_t*>(smem + PB + q * PS + k * 2) = (bf16_t)(pack_bf16x2(ex[c] / sum, 0.0f) & 0xffffu);
      }
    }
    lds_barrier();
    {
      f32x4_t o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < VROWS / 32; ++ks) {
        if (ks * 32 < T) {
          bf16x8_t vf;
#pragma unroll
          for (int e = 0; e < 8; ++e) vf[e] = *reinterpret_cast<const short*>(smem + VB + (ks * 32 + fgrp * 8 + e) * KS + (wave * 16 + frow) * 2);
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(smem + PB + frow * PS + fgrp * 16 + ks * 64), vf, o, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)       // rows past the window stay zero in the ctx buffer (block 0's out-projection of the NEXT run reads its pad rows)
        *reinterpret_cast<bf16_t*>(smem + CTX + (fgrp * 4 + i) * AS + (h * HD + wave * 16 + frow) * 2) =
            tile * SLOT + fgrp * 4 + i < T ? (bf16_t)(pack_bf16x2(o[i], 0.0f) & 0xffffu) : (bf16_t)0;
    }
    {   // FSMN memory term of the tile's rows: taps reach 5 rows into the neighbouring tiles, rows outside [0, T) are zero; thread = (channel, 4 rows)
      const int c = tid & 127, t0 = tile * SLOT + (tid >> 7) * 4;
      constexpr int PAD = (TAPS - 1) / 2, NV = 4 + TAPS - 1;
      float vr[NV];
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int tt = t0 + r - PAD;
        vr[r] = (tt >= 0 && tt < T) ? bf16_to_f32(*reinterpret_cast<const bf16_t*>(smem + VB + tt * KS + c * 2)) : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float m = bc;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) m = fmaf(wc[j], vr[i + j], m);
        mem[((tid >> 7) * 4 + i) * HD + c] = t0 + i < T ? m : 0.0f;
      }
    }
    lds_barrier();
    STAMP(5);
    {   // exchange 0: own 128 ctx columns out, the other three heads' in
      const int row = tid >> 5, off = (tid & 31) * 8;
      put8(reinterpret_cast<unsigned char*>(ctx_rows + (size_t)row * D + h * HD) + off, *reinterpret_cast<const u64*>(smem + CTX + row * AS + h * 256 + off));
      publish(flags + 0);
      sink ^= tw2 ^ tw2b;                                    // (drained by the publish)
      consume(flags + 0, a.err);
#pragma unroll
      for (int q = 1; q < NH; ++q) {
        const int hq = (h + q) & 3;
        *reinterpret_cast<u64*>(smem + CTX + row * AS + hq * 256 + off) = get8(reinterpret_cast<const unsigned char*>(ctx_rows + (size_t)row * D + hq * HD) + off);
      }
    }
    lds_barrier();
    STAMP(6);
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
    STAMP(7);
    put_slab_f32(xb_rows, smem + XB, tid, h);
    publish(flags + 1);
    u32x4 wc0[16], wc1[16];
    wload<16>(wc0, wpC, 0);
    consume(flags + 1, a.err);
    STAMP(8);
    // ---- phase C: LayerNorm of x1, FFN-1 columns 512 h + 64 wave ..
    norm_rows(xb_rows, smem, tid, h, false, a.ln_eps);
    lds_barrier();
    STAMP(9);
    {
      f32x4_t acc[4] = {};
      gemm_phase<4, 16, 4, 1, false>(wc0, wc1, wpC, a_lane + XN, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = h * 512 + wave * 64 + j * 16 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<bf16_t*>(smem + HID + (fgrp * 4 + i) * HS + col * 2) = (bf16_t)(pack_bf16x2(fmaxf(acc[j][i] + bC[j], 0.0f), 0.0f) & 0xffffu);
      }
    }
    lds_barrier();
    STAMP(10);
    u32x4 wd[16], wd1[16];
    {   // exchange 2: own 512 hid columns out, the other three quarters in
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = tid + NT * e, row = s >> 7, off = (s & 127) * 8;
        put8(reinterpret_cast<unsigned char*>(hid_rows + (size_t)row * DFF + h * 512) + off, *reinterpret_cast<const u64*>(smem + HID + row * HS + h * 1024 + off));
      }
      publish(flags + 2);
      wload<16>(wd, wpD, 0);
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = tid + NT * e, row = s >> 7, off = (s & 127) * 8;
          *reinterpret_cast<u64*>(smem + HID + row * HS + hq * 1024 + off) = t[e];
        }
      }
    }
    lds_barrier();
    STAMP(11);
    // ---- phase D: FFN-2 columns 128 h + 16 wave .. + b2 + x1 -> the tile's rows of the next block
    {
      f32x4_t acc[2] = {};
      gemm_phase<1, 64, 16, 2, false>(wd, wd1, wpD, hid_lane, acc);
      const int col = wave * 16 + frow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = fgrp * 4 + i;
        xres[row * HD + col] = (acc[0][i] + acc[1][i]) + bD + xbs[row * HD + col];
      }
    }
    lds_barrier();
    STAMP(12);
    put_slab_f32(x_rows, smem + XRES, tid, h);
    publish(flags + 3);
    sink ^= tw ^ twb;
    STAMP(13);
    if (li + 1 < a.n_layers) {
      wload<12>(wa, a.layers[li + 1].wpack + PK_A + wave_frag * PW_A + lane * 16, 0);
      if (!(a.opt & 1)) {
        const unsigned char* nw = a.layers[li + 1].wpack;
        if (by_head) {
          tw = warm(nw + PK_A + (size_t)h * NW * PW_A, (int)(NW * PW_A / 128), wg_xcd, n_wg_xcd, tid);
          twb = warm(nw + PK_B + (size_t)h * NW * PW_B, (int)(NW * PW_B / 128), wg_xcd, n_wg_xcd, tid);
        } else tw = warm(nw + PK_A, (int)((PK_C - PK_A) / 128), wg_xcd, n_wg_xcd, tid);
      }
    }
  }
  if (sink == 0x9e3779b9u && a.n_layers < 0) a.err[1] = sink;
}

}  // namespace

bool sanm_tiles_supported(int max_T, int d, int d_ffn, int n_heads, int d_head, int ktaps) {
  return max_T <= MAXT && d == D && d_ffn == DFF && n_heads == NH && d_head == HD && ktaps == TAPS;
}
int sanm_tiles_max_tiles() { return gemm_env_cus() / 8 * 2; }   // one workgroup per CU, tiles in pairs: 64 tiles x 4 heads on 256 CUs (the device's count as of the last gemm_reload_env(); a larger batch takes the other paths)

void launch_sanm_tiles(const SanmTilesArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_tiles >= 1 && a.n_tiles <= sanm_tiles_max_tiles() && a.n_layers >= 1, "sanm_tiles: %d tiles", a.n_tiles);
  static PerDeviceOnce attr_once;
  if (attr_once.first())
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sanm_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int n_wgs = (a.opt & 2) ? (a.n_tiles + 7) / 8 * 32 : (a.n_tiles + 1) / 2 * 8;
  hipLaunchKernelGGL(sanm_tiles_kernel, dim3(n_wgs), dim3(NT), LDS_BYTES, s, a);
  HIP_CHECK(hipGetLastError());
}
