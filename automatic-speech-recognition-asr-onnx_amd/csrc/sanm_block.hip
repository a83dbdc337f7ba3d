// One SANM encoder block (Export_SenseVoice.py:227-258: LayerNorm -> q|k|v -> soft-max attention + FSMN -> out-projection + residual
// -> LayerNorm -> FFN + residual) as ONE launch for batches of <= 144-row windows (8 s chunks): the bf16 headline path.
//
// Decomposition. A window's block is independent of every other window, and 64 windows x 4 heads = 256 = one workgroup per CU:
// the four workgroups (w, h) of window w form a CLUSTER. Workgroup h owns head h of the attention half and column slab h of every
// GEMM of the block, so each GEMM phase is a [144 rows] x [N / 4 columns] tile whose A operand (all 512 / 2048 input columns of the
// window) is what the cluster exchanged at the previous phase boundary:
//
//   A  q|k|v projection of head h (LayerNorm folded in: raw bf16 rows + row statistics + column sums), attention and FSMN in LDS
//        -> ctx[:, 128 h ..] (bf16)                         ── exchange 0 ──>
//   B  out-projection slab: ctx[144][512] x Wout[128 h.. ][512]^T + FSMN term (LDS, never in memory) + residual (f32)
//        -> x1 slab: f32 stays in REGISTERS until phase D, bf16 copy + row statistics  ── exchange 1 ──>
//   C  FFN-1 slab, two halves of 256 columns: relu(LN(x1) W1[512 h..]^T + b1) -> hid[:, 512 h ..] (bf16)   ── exchange 2 ──>
//   D  FFN-2 slab: hid[144][2048] x W2[128 h..][2048]^T + b2 + x1 (registers) -> x (f32), bf16 copy + row statistics of the next block
//
// What this removes against the four-launch path (fused attention half, out-proj, FFN-1, FFN-2 GEMM launches): three launch
// boundaries with their single-round prologue / epilogue bubbles, the f32 round trip of the FSMN term and of x1 (2 x 19 MB written and
// read per block at batch 64), the second read of the f32 residual, and -- because the cluster's producers and consumers share one
// XCD (workgroup b runs on XCD b % 8: placement is used for speed only) -- most exchanged rows are read back from that XCD's L2.
//
// Exchange protocol (cdna_hip_programming.md Guideline 16, R1): payload stores are write-through (sc1), every wave drains its stores,
// one lane counts the workgroup in on the (window, exchange) flag with a relaxed agent-scope atomic; a consumer polls that word
// relaxed, issues ONE agent-scope acquire and then reads with ordinary (LDS-DMA) loads. Flags are zeroed by a memset node ahead of
// the first block of a forward pass; every (block, window, exchange) has its own word. Every spin is bounded: a workgroup that gives up
// raises err[0] and carries on, so a broken launch produces garbage and an error status, never a hung GPU.
//
// All four workgroups of a cluster must be resident together, so a launch takes at most (CUs / 4) windows; the host splits larger
// batches into several launches per block.
#include <algorithm>
#include <type_traits>
#include "kernels.h"

namespace {

constexpr int R = 144, RF = 9;                  // rows / row fragments of a window tile
constexpr int HD = 128, NH = 4, D = 512, DFF = 2048;
constexpr int NW = 12, NT = NW * 64;            // waves / threads per workgroup
constexpr int TAPS = 11;
// ---- LDS map (bytes). Phase A: two projection stages, then the q / k / v^T images; later phases reuse the space (see each phase)
constexpr int A_ABYTES = R * 64, A_WBYTES = 3 * HD * 64, A_STAGE = A_ABYTES + A_WBYTES, A_NS = 4;    // K-step 32: 9216 + 24576 per stage, 4 stages
constexpr int QS = 0, KS = QS + R * 256, KEYS = 160, VS = KS + KEYS * 256, IMG_END = VS + HD * 512;    // 0, 36864, 77824, 143360
constexpr int LDS_BYTES = 160 * 1024;                                      // the whole CU: one workgroup per CU
constexpr int ST_F = LDS_BYTES - R * 8;                                    // (mean, rstd) [144] float2, at the very end
constexpr int ST_P = ST_F - 4 * R * 8;                                     // statistics partials [4][144] float2
static_assert(R * HD * 4 <= ST_P, "the FSMN term (f32 [144][128] at offset 0, over the dead q / k images) overlaps the statistics");
static_assert(A_NS * A_STAGE <= ST_P && IMG_END <= ST_P, "phase A overlaps the statistics");
static_assert(6 * (R + 256) * 64 <= ST_P && 9 * (R + 128) * 64 <= ST_P, "phase B / C / D rings overlap the statistics");

__device__ __forceinline__ int frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }
__device__ __forceinline__ int w_swz(int r) { return (((r >> 3) & 3) << 1) | ((r >> 1) & 1); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// write-through (sc1) stores: the payload of an exchange must be in memory, not dirty in this XCD's L2, when the flag is raised
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
// (The hooks live in a second instantiation of the kernel, ABL = true, launched only when ASR_SANM_BLOCK_ABL is set: in the product instance
//  they would put branches between the fragment reads and the MFMAs, and the compiler then falls back to s_waitcnt lgkmcnt(0) everywhere.)
__device__ int g_loop_dbg;          // timing-only ablations of the GEMM loops (ASR_SANM_BLOCK_ABL): 1 = no refills after the prologue, 2 = no MFMA, 4 = no fragment reads, 8 = weight rows alias one 64-row region (every weight load hits L2), 16 = payload / output stores skipped
template <bool ABL = false>
__device__ __forceinline__ void store16_wt(void* p, uint4 v) {
  if (ABL && (g_loop_dbg & 16)) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); return; }
  const u32x4_t w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void store8_wt(void* p, float2 v) {
  const f32x2_t w = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

// count this workgroup in on an exchange flag: every wave's stores drained, then one relaxed agent-scope add
__device__ __forceinline__ void publish(unsigned* flag, bool withhold = false) {     // withhold: fault injection (tests), the count never arrives
  wait_vm<0>();
  __syncthreads();
  if (threadIdx.x == 0 && !withhold) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until all `need` workgroups of the cluster are in; ONE acquire (drops this CU's stale L1 lines), then ordinary loads
__device__ __forceinline__ void consume(unsigned* flag, unsigned need, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(2);
      // a cluster in step waits 0.4 .. 2.5 us here; 2^13 polls (about 8 ms: a poll is a sleep + an L2 round trip, about 1 us) means the cluster is not
      // co-resident: give up, never hang -- the host redoes the pass on the four-launch path and keeps the session there for a while
      if (++spins > (1u << 13)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if ((spins & 255u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;           // somebody already gave up: the launch is void anyway
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// Exact counted wait: at most n loads of this wave still in flight (n = P x stages issued after the one about to be read)
template <int P, int MAXS> __device__ __forceinline__ void wait_stages(int stages_after) {
  if constexpr (MAXS <= 0) { wait_vm<0>(); }
  else { if (stages_after >= MAXS) wait_vm<P * MAXS>(); else wait_stages<P, MAXS - 1>(stages_after); }
}

// 64-byte stage rows (K-step 32, phase A): chunk c of row r sits at position c ^ hk((r >> 2) & 3) for rows read as natural 16-row
// fragments -- every ds_read_b128 service group then touches each bank once (checked by enumeration).
__device__ __forceinline__ int hk(int sel) { return (0x78 >> (2 * (sel & 3))) & 3; }          // {0, 2, 3, 1}[sel]

// ---- [144 x (NJ * 16 * 4)] tile = A[144][K] (rows a0.., pitch lda) x W[RW][K]^T (rows w0.., pitch ldw), K-step 64, ring of NS stages filled
// by LDS-DMA with the 16-byte-slot swizzle on the global side. 12 waves = 3 row groups (48 rows) x 4 column groups (NJ fragments = 16 NJ
// columns); W rows are fetched in the paired-fragment order so that a lane ends with 8 consecutive output columns per fragment pair.
// Every wave issues P = ceil(pieces / 12) loads per stage (the last piece is re-loaded by the spare slots: same bytes, same place).
template <int RW, int NS, int NJ, bool FULL, bool PIPE, bool ABL>     // FULL: all 9 row fragments active (an 8 s window): straight-line MFMA body, no per-fragment branches
__device__ __forceinline__ void tile_loop_impl(unsigned char* ring, const bf16_t* a0, int lda, int a_rows_left, const bf16_t* w0, int ldw, int nk,
                                          int n_act, f32x4_t (&acc)[3][NJ], int lane, int wave) {
  constexpr int NP = (R + RW) / 8, P = (NP + NW - 1) / NW, STAGE = (R + RW) * 128;
  static_assert(RW == NJ * 16 * 4, "four column groups");
  static_assert(NS >= 3 || !PIPE, "the software pipeline reads stage kt + 1 one iteration after its DMA was issued");
  const int srow = lane >> 3, frow = lane & 15, fgrp = lane >> 4, rg = wave >> 2, cg = wave & 3;
  const bf16_t* src[P];
  int dst[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int ii = min(wave + NW * t, NP - 1);
    dst[t] = ii * 1024;
    if (ii < R / 8) {
      const int r = ii * 8 + srow;
      src[t] = a0 + (size_t)min(r, a_rows_left - 1) * lda + (((lane & 7) ^ srow) << 3);
    } else {
      const int wr = (ii - R / 8) * 8 + srow;
      src[t] = w0 + (size_t)((ABL && (g_loop_dbg & 8)) ? (wr & 63) : wr) * ldw + (((lane & 7) ^ w_swz(wr)) << 3);
    }
  }
  auto stage = [&](int slot, int k0) {
    unsigned char* base = ring + slot * STAGE;
#pragma unroll
    for (int t = 0; t < P; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                       (__attribute__((address_space(3))) void*)(base + dst[t]), 16, 0, 0);
  };
  int a_off[2], w_off[2][NJ];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c = kk * 4 + fgrp;
    a_off[kk] = (rg * 48 + frow) * 128 + ((c ^ (frow & 7)) << 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = cg * (NJ * 16) + frag_col(j, frow); w_off[kk][j] = R * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }
  }
  const int my_frags = min(3, max(0, n_act - rg * 3));         // active row fragments of this wave (wave-uniform)
  const int dbg = ABL ? g_loop_dbg : 0;
  // Software pipeline across the per-K-step barrier: the fragments of (stage kt, k-half 1) are read while the MFMAs of k-half 0 run,
  // the barrier that makes stage kt + 1 visible sits between the two MFMA groups, and the fragments of (stage kt + 1, k-half 0) are read
  // while the MFMAs of k-half 1 run -- a wave never waits on an LDS round trip with an idle matrix pipe.
  bf16x8_t wf0[NJ], af0[3], wf1[NJ], af1[3];
  auto read_frags = [&](const unsigned char* St, int kk, bf16x8_t (&wf)[NJ], bf16x8_t (&af)[3]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(St + w_off[kk][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (FULL || i < my_frags) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_off[kk] + i * 2048);
  };
  auto mfmas = [&](const bf16x8_t (&wf)[NJ], const bf16x8_t (&af)[3]) {
    if (dbg & 2) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(wf[j]));
#pragma unroll
      for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(af[i]));
      return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (FULL || i < my_frags) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s * 64);
  if constexpr (!PIPE) {           // plain form (fewer live registers): one barrier per K-step, fragments read after it
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NS - 2 < nk) wait_vm<P * (NS - 2)>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + NS - 1 < nk && !(dbg & 1)) stage((kt + NS - 1) % NS, (kt + NS - 1) * 64);
      const unsigned char* St = ring + (kt % NS) * STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        read_frags(St, kk, wf0, af0);
        mfmas(wf0, af0);
      }
    }
    __syncthreads();
    return;
  }
  if (NS - 2 < nk) wait_vm<P * (NS - 2)>(); else wait_vm<0>();       // stage 0 landed (this wave's pieces) ...
  __builtin_amdgcn_s_barrier();                                      // ... and everyone else's
  read_frags(ring, 0, wf0, af0);
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* St = ring + (kt % NS) * STAGE;
    read_frags(St, 1, wf1, af1);
    mfmas(wf0, af0);
    if (kt + 1 < nk) {
      // stage kt + 1 must have landed: the loads issued after it are the stages kt + 2 .. kt + NS - 2 (those that exist)
      if (kt + NS - 2 < nk) wait_vm<P * (NS > 2 ? NS - 3 : 0)>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();        // every wave has finished READING stage kt - 1 (its MFMAs ran), so that slot may be refilled
      if (kt + NS - 1 < nk && !(dbg & 1)) stage((kt + NS - 1) % NS, (kt + NS - 1) * 64);
      read_frags(ring + ((kt + 1) % NS) * STAGE, 0, wf0, af0);
    }
    mfmas(wf1, af1);
  }
  __syncthreads();          // every wave is done with the ring (and no DMA is in flight): the caller may reuse the space
}
template <int RW, int NS, int NJ, bool PIPE, bool ABL>
__device__ __forceinline__ void tile_loop(unsigned char* ring, const bf16_t* a0, int lda, int a_rows_left, const bf16_t* w0, int ldw, int nk,
                                          int n_act, f32x4_t (&acc)[3][NJ], int lane, int wave) {
  if (n_act == RF) tile_loop_impl<RW, NS, NJ, true, PIPE, ABL>(ring, a0, lda, a_rows_left, w0, ldw, nk, n_act, acc, lane, wave);
  else tile_loop_impl<RW, NS, NJ, false, false, ABL>(ring, a0, lda, a_rows_left, w0, ldw, nk, n_act, acc, lane, wave);
}

__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 w;
  w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]); w.z = pack_bf16x2(v[4], v[5]); w.w = pack_bf16x2(v[6], v[7]);
  return w;
}
// (sum, sum of squares) of the 8 bf16-ROUNDED values of a packed group
__device__ __forceinline__ void stats8(const uint4& w, float& s1, float& s2) {
  const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = __uint_as_float(wd[e] << 16), hi = __uint_as_float(wd[e] & 0xffff0000u);
    s1 += lo + hi;
    s2 = fmaf(lo, lo, fmaf(hi, hi, s2));
  }
}

// Per-phase views: the kernel arguments are re-read from the kernarg segment and the lane id is made opaque at every phase start, so
// the compiler cannot hoist a later phase's pointers / per-lane offsets above the projection loop (they would be spilled around it:
// 168 registers at three waves per SIMD, and a scratch reload next to hand-placed LDS-DMA costs a full vmcnt(0) drain)
typedef const __attribute__((address_space(4))) SanmBlockArgs* KernArgs;
__device__ __forceinline__ KernArgs phase_args() {
  KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

#define STAMP(k) do { if (a->times && threadIdx.x == 0) a->times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

template <bool ABL>
__global__ __launch_bounds__(NT) void sanm_block_kernel(const SanmBlockArgs a_byval) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  KernArgs a = phase_args();
  int lane = tid & 63;
  int frow = lane & 15, fgrp = lane >> 4;
  // cluster placement: workgroup b runs on XCD b % 8 (observed, not guaranteed): the four workgroups of a window get ids 8 apart
  int cl, h;
  if (a->scatter) { cl = blockIdx.x >> 2; h = blockIdx.x & 3; }            // test mode: a cluster spread over four XCDs
  else { const int idx = blockIdx.x >> 3; cl = ((idx >> 2) << 3) + (blockIdx.x & 7); h = idx & 3; }
  if (cl >= a->n_utts) return;
  const int u = a->utt0 + cl;
  const UttPlan up = a->plan[u];
  const int T = up.T, row0 = up.row_off;
  const int n_act = (T + 15) >> 4;                        // active row fragments (cluster-uniform)
  const int rows_left = a->n_rows_alloc - row0;            // readable rows from row0 on
  unsigned* flags = a->flags + (size_t)cl * 4;
  const int rg = wave >> 2, cg = wave & 3;
  STAMP(0);

  f32x4_t xpre[3][2];                                      // this lane's part of the f32 residual slab, requested while the FSMN runs
  // ================================================================ phase A: q|k|v projection of head h, attention, FSMN
  {
    const int grp = wave >> 2, sub = wave & 3;              // grp: 0 q, 1 k, 2 v
    constexpr int A_AI = R / 16, A_WI = 3 * HD / 16, A_NP = A_AI + A_WI, A_P = (A_NP + NW - 1) / NW;      // 9 + 24 pieces of 16 rows x 64 B: 3 per wave
    const int r16 = lane >> 2, ch = ((lane & 3) ^ hk(r16 >> 2)) << 3;
    const bf16_t* hb = a->x_lo;
    const bf16_t* wb = a->wqkv;
    const bf16_t* src[A_P];
    int dst[A_P];
#pragma unroll
    for (int t = 0; t < A_P; ++t) {
      const int ii = min(wave + t * NW, A_NP - 1);
      dst[t] = ii * 1024;
      if (ii < A_AI) {
        src[t] = hb + (size_t)(row0 + min(ii * 16 + r16, rows_left - 1)) * D + ch;
      } else {
        const int wr = (ii - A_AI) * 16 + r16;              // 0..383: q | k | v rows of this head
        src[t] = wb + (size_t)((ABL && (g_loop_dbg & 8)) ? (wr & 63) : (wr >> 7) * D + h * HD + (wr & 127)) * D + ch;
      }
    }
    auto stage = [&](int slot, int k0) {
      unsigned char* base = smem + slot * A_STAGE;
#pragma unroll
      for (int t = 0; t < A_P; ++t)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                         (__attribute__((address_space(3))) void*)(base + dst[t]), 16, 0, 0);
    };
    f32x4_t acc[RF][2];
#pragma unroll
    for (int i = 0; i < RF; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    constexpr int nk = D / 32;
    const int fsw = (fgrp ^ hk(frow >> 2)) << 4;            // this lane's 16-byte chunk position inside a 64-byte stage row
    const int a_lane = frow * 64 + fsw, w_lane = A_ABYTES + (grp * HD + sub * 32 + frow) * 64 + fsw;
    const bool ln_here = !a->st_in;                          // no producer statistics (first block after a stand-alone LayerNorm): derive them here
    const int st_row = tid % R, st_sg = tid / R;            // threads < 576: (row, one of the four 16-byte chunk positions)
    float st_s = 0.0f, st_ss = 0.0f;
    float2* st_part = reinterpret_cast<float2*>(smem + ST_P);
    float2* st_fin = reinterpret_cast<float2*>(smem + ST_F);
    if (!ln_here && tid < R) {
      const float2 ss = sum_row_partials(a->st_in + (size_t)(row0 + min(tid, rows_left - 1)) * (D / 32), D / 32);
      const float mean = ss.x * (1.0f / D);
      const float var = fmaxf(ss.y * (1.0f / D) - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + a->ln_eps));
    }
    auto k_loop = [&](auto full_tag, auto swap_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      constexpr bool SWAP = decltype(swap_tag)::value;
#pragma unroll
      for (int s = 0; s < A_NS - 1; ++s) stage(s, s * 32);
      for (int kt = 0; kt < nk; ++kt) {
        wait_stages<A_P, A_NS - 2>(nk - 1 - kt);
        __builtin_amdgcn_s_barrier();
        if (kt + A_NS - 1 < nk) stage((kt + A_NS - 1) % A_NS, (kt + A_NS - 1) * 32);
        const unsigned char* St = smem + (kt % A_NS) * A_STAGE;
        if (ln_here && wave < RF) {
          const uint4 raw = *reinterpret_cast<const uint4*>(St + st_row * 64 + st_sg * 16);
          const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(wds[e] << 16), hi = __uint_as_float(wds[e] & 0xffff0000u);
            st_s += lo + hi;
            st_ss = fmaf(lo, lo, fmaf(hi, hi, st_ss));
          }
        }
        const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(St + w_lane);
        const bf16x8_t w1 = *reinterpret_cast<const bf16x8_t*>(St + w_lane + 16 * 64);
        bf16x8_t af[RF];
#pragma unroll
        for (int i = 0; i < RF; ++i)
          if (FULL || i < n_act) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_lane + i * 1024);
#pragma unroll
        for (int i = 0; i < RF; ++i) {
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
    };
    if (n_act == RF) {
      if (grp < 2) k_loop(std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{});
    } else {
      if (grp < 2) k_loop(std::false_type{}, std::true_type{}); else k_loop(std::false_type{}, std::false_type{});
    }
    if (ln_here && wave < RF) st_part[st_sg * R + st_row] = make_float2(st_s, st_ss);
    __syncthreads();                                        // ring is dead: the q / k / v^T images may overwrite it
    STAMP(1);
    if (ln_here) {
      if (tid < R) {
        const float2 p0 = st_part[tid], p1 = st_part[R + tid], p2 = st_part[2 * R + tid], p3 = st_part[3 * R + tid];
        const float mean = ((p0.x + p1.x) + (p2.x + p3.x)) * (1.0f / D);
        const float var = fmaxf(((p0.y + p1.y) + (p2.y + p3.y)) * (1.0f / D) - mean * mean, 0.0f);
        st_fin[tid] = make_float2(mean, rsqrtf(var + a->ln_eps));
      }
      __syncthreads();
    }
    if (grp < 2) {            // acc[i][j][r] = C[16 i + frow][32 sub + 16 j + 4 fgrp + r] -> row-major image, 8-byte writes
      unsigned char* dst = smem + (grp == 0 ? QS : KS);
      const float* bias = a->bqkv + grp * D + h * HD;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = sub * 32 + j * 16 + fgrp * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
        const float4 c4 = *reinterpret_cast<const float4*>(a->cqkv + grp * D + h * HD + col);
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          const int row = i * 16 + frow;
          const float2 mr = st_fin[row];
          const float v0 = (acc[i][j][0] - mr.x * c4.x) * mr.y, v1 = (acc[i][j][1] - mr.x * c4.y) * mr.y;
          const float v2 = (acc[i][j][2] - mr.x * c4.z) * mr.y, v3 = (acc[i][j][3] - mr.x * c4.w) * mr.y;
          uint2 w;
          w.x = pack_bf16x2(v0 + b4.x, v1 + b4.y);
          w.y = pack_bf16x2(v2 + b4.z, v3 + b4.w);
          *reinterpret_cast<uint2*>(dst + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + ((col >> 2) & 1) * 8) = w;
        }
      }
    } else {                  // acc[i][j][r] = C[16 i + 4 fgrp + r][32 sub + 16 j + frow] -> V^T[d][t] image
      unsigned char* dst = smem + VS;
      const float* bias = a->bqkv + 2 * D + h * HD;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int dcol = sub * 32 + j * 16 + frow;
        const float b = bias[dcol];
        const float cs = a->cqkv[2 * D + h * HD + dcol];
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float2 mr = st_fin[i * 16 + fgrp * 4 + r]; v[r] = (v[r] - mr.x * cs) * mr.y; }
          uint2 w;
          w.x = pack_bf16x2(v[0] + b, v[1] + b);
          w.y = pack_bf16x2(v[2] + b, v[3] + b);
          *reinterpret_cast<uint2*>(dst + dcol * 512 + (((2 * i + (fgrp >> 1)) ^ (dcol & 15)) << 4) + (fgrp & 1) * 8) = w;
        }
      }
      if (sub == 0 && lane < 32) {      // keys 144..159 of the last 32-key sub-tile: zeros so that 0 * v stays 0
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int dcol = rr * 16 + (lane & 15), slot = 18 + (lane >> 4);
          *reinterpret_cast<uint4*>(dst + dcol * 512 + ((slot ^ (dcol & 15)) << 4)) = make_uint4(0, 0, 0, 0);
        }
      }
    }
    __syncthreads();

    // ---- attention, waves 0..8 = one 16-query tile each; all <= 160 scores of a tile live in registers (one soft-max pass). The
    //      context tile goes back into the wave's OWN (dead) q rows of the LDS image, so the workgroup can store whole rows.
    if (wave < RF && wave * 16 < T) {
      const int q0 = wave * 16;
      const unsigned char* Qs = smem + QS;
      const unsigned char* Ks = smem + KS;
      const unsigned char* Vs = smem + VS;
      const int fq = frow, g = fgrp;
      const int qrow = q0 + fq;
      const int n_sub = (T + 31) >> 5;
      bf16x8_t qf[HD / 32];
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + qrow * 256 + (((ks * 4 + g) ^ (qrow & 15)) << 4));
      f32x4_t st[5][2];
#pragma unroll
      for (int s = 0; s < 5; ++s) { st[s][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; st[s][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        if (s < n_sub) {
          const int key0 = s * 32 + fq, key1 = key0 + 16;
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks) {
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
      f32x4_t ot[HD / 16];
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) ot[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        if (s < n_sub) {
#pragma unroll
          for (int dt = 0; dt < HD / 16; ++dt) {
            const int d = dt * 16 + fq;
            const unsigned char* vr = Vs + d * 512 + (g & 1) * 8;
            union { bf16x8_t v; uint2 h2[2]; } vf;
            vf.h2[0] = *reinterpret_cast<const uint2*>(vr + (((s * 4 + (g >> 1)) ^ (d & 15)) << 4));
            vf.h2[1] = *reinterpret_cast<const uint2*>(vr + (((s * 4 + 2 + (g >> 1)) ^ (d & 15)) << 4));
            ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[s], ot[dt], 0, 0, 0);
          }
        }
      }
      // O^T[d = 16 dt + 4 g + r][q = fq] -> ctx image row qrow (only this wave reads / writes these q rows), 16-byte chunk dt * 2 + g / 2
      unsigned char* crow = smem + QS + qrow * 256;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        uint2 w;
        w.x = pack_bf16x2(ot[dt][0] * inv, ot[dt][1] * inv);
        w.y = pack_bf16x2(ot[dt][2] * inv, ot[dt][3] * inv);
        *reinterpret_cast<uint2*>(crow + (((dt * 2 + (g >> 1)) ^ (qrow & 15)) << 4) + (g & 1) * 8) = w;
      }
    }
    __syncthreads();
    STAMP(2);
    // ---- context rows of this head -> memory (whole 16-byte chunks, write-through) and count the workgroup in on exchange 0;
    //      the FSMN below then runs while the other three heads finish
    {
      bf16_t* cg_ = a->ctx + (size_t)row0 * D + h * HD;
      for (int c = tid; c < n_act * 16 * 16; c += NT) {
        const int row = c >> 4, ch = c & 15;
        uint4 v = *reinterpret_cast<const uint4*>(smem + QS + row * 256 + ((ch ^ (row & 15)) << 4));
        if (row >= T) v = make_uint4(0, 0, 0, 0);          // alignment rows past the window: zeros, like the separate kernels leave them
        store16_wt<ABL>(cg_ + (size_t)row * D + ch * 8, v);
      }
    }
    publish(flags + 0, ABL && a->fault != 0 && blockIdx.x == 5);     // fault injection exists in the test / ablation instance only
    STAMP(3);
    // the residual rows of this workgroup's slab depend on nobody else: request them now, they arrive under the FSMN / the exchange wait
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = min((wave >> 2) * 48 + i * 16 + (lane & 15), rows_left - 1);
      const float* xr = a->x + (size_t)(row0 + row) * D + h * HD + (wave & 3) * 32 + (lane >> 4) * 8;
      xpre[i][0] = *reinterpret_cast<const f32x4_t*>(xr); xpre[i][1] = *reinterpret_cast<const f32x4_t*>(xr + 4);
    }
    // ---- FSMN memory (thread = channel x 24-step segment, 11 taps from the V^T image) -> f32 [144][128] image at offset 0 (the q / k
    //      images are dead), 32-byte granules XOR-swizzled by the row so that the out-projection epilogue reads without conflicts
    {
      constexpr int PAD = (TAPS - 1) / 2;
      const int c = tid & (HD - 1), seg = tid >> 7, cgl = h * HD + c;
      const int t0 = seg * 24, T16 = n_act * 16;
      if (t0 < T16) {
        const unsigned char* vrow = smem + VS + c * 512;
        float wc[TAPS];
#pragma unroll
        for (int j = 0; j < TAPS; ++j) wc[j] = a->wfsmn[cgl * TAPS + j];
        const float bc = a->bfsmn[cgl];
        float x[40];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const int sl = seg * 3 - 1 + q;
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
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          float accv = bc;
#pragma unroll
          for (int j = 0; j < TAPS; ++j) accv = fmaf(wc[j], x[8 + i + j - PAD], accv);
          const int t = t0 + i;
          if (t < T16) *reinterpret_cast<float*>(smem + t * 512 + (((c >> 3) ^ (t & 15)) << 5) + (c & 7) * 4) = (t < T) ? accv : 0.0f;
        }
      }
    }
    __syncthreads();
  }

  // ================================================================ phase B: out-projection slab + FSMN term + residual -> x1
  float xres[3][8];                                        // x1 (then the FFN residual) of this lane: rows 48 rg + 16 i + frow, 8 columns
  {
    a = phase_args();
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    STAMP(4);
    consume(flags + 0, NH, a->err);
    STAMP(5);
    // the accumulators start from FSMN term (f32 image at offset 0) + residual (registers, requested before the FSMN), so the image is
    // dead before the first DMA lands and the operand ring of this phase can take the whole LDS (four stages in flight instead of two)
    const int n = cg * 32 + fgrp * 8;                       // this lane's 8 consecutive columns inside the slab
    f32x4_t acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = rg * 48 + i * 16 + frow;
      const unsigned char* mrow = smem + row * 512 + ((((n >> 3) ^ (row & 15))) << 5);
      if (rg * 3 + i < n_act) { acc[i][0] = *reinterpret_cast<const f32x4_t*>(mrow) + xpre[i][0]; acc[i][1] = *reinterpret_cast<const f32x4_t*>(mrow + 16) + xpre[i][1]; }
      else { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    }
    __syncthreads();
    tile_loop<128, 4, 2, true, ABL>(smem, a->ctx + (size_t)row0 * D, D, rows_left, a->wout + (size_t)h * HD * D, D, D / 64, n_act, acc, lane, wave);
    STAMP(6);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = rg * 48 + i * 16 + frow;
      if (rg * 3 + i < n_act) {
        float* v = xres[i];                                 // (the accumulators started from FSMN term + residual)
        v[0] = acc[i][0][0]; v[1] = acc[i][0][1]; v[2] = acc[i][0][2]; v[3] = acc[i][0][3];
        v[4] = acc[i][1][0]; v[5] = acc[i][1][1]; v[6] = acc[i][1][2]; v[7] = acc[i][1][3];
        const uint4 pk = pack8(xres[i]);
        store16_wt<ABL>(a->x1_lo + (size_t)(row0 + row) * D + h * HD + n, pk);
        float s1 = 0.0f, s2 = 0.0f;
        stats8(pk, s1, s2);
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (fgrp == 0) store8_wt(a->st1 + (size_t)(row0 + row) * (D / 32) + h * 4 + cg, make_float2(s1, s2));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xres[i][e] = 0.0f;
      }
    }
    publish(flags + 1);
    STAMP(7);
  }

  // ================================================================ phase C: FFN-1 slab (two halves of 256 columns), LayerNorm folded in
  {
    a = phase_args();
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    consume(flags + 1, NH, a->err);
    STAMP(8);
    float2* st_fin = reinterpret_cast<float2*>(smem + ST_F);
    if (tid < R) {
      const float2 ss = sum_row_partials(a->st1 + (size_t)(row0 + min(tid, rows_left - 1)) * (D / 32), D / 32);
      const float mean = ss.x * (1.0f / D);
      const float var = fmaxf(ss.y * (1.0f / D) - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + a->ln_eps));
    }
    __syncthreads();
#pragma unroll 1
    for (int hc = 0; hc < 2; ++hc) {
      const int col0 = h * 512 + hc * 256;                  // first hidden column of this half
      f32x4_t acc[3][4];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const unsigned long long tc0 = a->times ? wall_clock64() : 0ull;
      tile_loop<256, 3, 4, false, ABL>(smem, a->x1_lo + (size_t)row0 * D, D, rows_left, a->w1 + (size_t)col0 * D, D, D / 64, n_act, acc, lane, wave);
      if (a->times && threadIdx.x == 0) a->times[(size_t)blockIdx.x * 16 + 14] += wall_clock64() - tc0;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = col0 + cg * 64 + p * 32 + fgrp * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(a->b1 + n), b1v = *reinterpret_cast<const float4*>(a->b1 + n + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(a->c1 + n), c1v = *reinterpret_cast<const float4*>(a->c1 + n + 4);
        const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
        const float c8[8] = {c0.x, c0.y, c0.z, c0.w, c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (rg * 3 + i < n_act) {
            const int row = rg * 48 + i * 16 + frow;
            const float2 mr = st_fin[row];
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r]; v[4 + r] = acc[i][2 * p + 1][r]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf((v[e] - mr.x * c8[e]) * mr.y + b8[e], 0.0f);
            store16_wt<ABL>(a->hid + (size_t)(row0 + row) * DFF + n, pack8(v));
          }
        }
      }
    }
    STAMP(9);
    publish(flags + 2);
    STAMP(10);
  }

  // ================================================================ phase D: FFN-2 slab + bias + x1 -> x (f32), bf16 copy + row statistics
  {
    a = phase_args();
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    consume(flags + 2, NH, a->err);
    STAMP(11);
    f32x4_t acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    tile_loop<128, 4, 2, true, ABL>(smem, a->hid + (size_t)row0 * DFF, DFF, rows_left, a->w2 + (size_t)h * HD * DFF, DFF, DFF / 64, n_act, acc, lane, wave);
    STAMP(12);
    const int n = cg * 32 + fgrp * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(a->b2 + h * HD + n), b1v = *reinterpret_cast<const float4*>(a->b2 + h * HD + n + 4);
    const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (rg * 3 + i < n_act) {
        const int row = rg * 48 + i * 16 + frow;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][1][r]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e] + xres[i][e];
        float* xo = a->x + (size_t)(row0 + row) * D + h * HD + n;
        const uint4 pk = pack8(v);
        if (!(ABL && (g_loop_dbg & 16))) {
          *reinterpret_cast<float4*>(xo) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(xo + 4) = make_float4(v[4], v[5], v[6], v[7]);
          *reinterpret_cast<uint4*>(a->x_lo_out + (size_t)(row0 + row) * D + h * HD + n) = pk;
        }
        float s1 = 0.0f, s2 = 0.0f;
        stats8(pk, s1, s2);
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (fgrp == 0) a->st_out[(size_t)(row0 + row) * (D / 32) + h * 4 + cg] = make_float2(s1, s2);
      }
    }
    wait_vm<0>();
    STAMP(13);
  }
}

__global__ void rows_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    uint4 w;
    w.x = pack_bf16x2(a.x, a.y); w.y = pack_bf16x2(a.z, a.w); w.z = pack_bf16x2(b.x, b.y); w.w = pack_bf16x2(b.z, b.w);
    reinterpret_cast<uint4*>(y)[i] = w;
  }
}

}  // namespace

void launch_rows_to_bf16(const float* x, bf16_t* y, size_t n, hipStream_t s) {
  ASR_REQUIRE(n % 8 == 0, "rows_to_bf16: element count must be a multiple of 8");
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, s, x, y, n8);
  HIP_CHECK(hipGetLastError());
}

bool sanm_block_supported(int max_T, int d_head, int n_heads, int d, int d_ffn, int fsmn_taps) {
  return max_T <= R && d_head == HD && n_heads == NH && d == D && d_ffn == DFF && fsmn_taps == TAPS;
}

int sanm_block_max_utts() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return (cus / 32) * 8;          // whole groups of 8 windows (one per XCD), four workgroups each, one workgroup per CU
}

void launch_sanm_block(const SanmBlockArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_utts > 0 && a.n_utts <= sanm_block_max_utts(), "sanm_block: %d windows per launch (max %d)", a.n_utts, sanm_block_max_utts());
  ASR_REQUIRE(a.x_lo && a.x && a.ctx && a.x1_lo && a.st1 && a.hid && a.x_lo_out && a.st_out && a.flags && a.err && a.plan, "sanm_block: null buffer");
  static PerDeviceOnce attr_once;
  static int abl = 0;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sanm_block_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sanm_block_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    abl = getenv("ASR_SANM_BLOCK_ABL") ? atoi(getenv("ASR_SANM_BLOCK_ABL")) : 0;
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_loop_dbg), &abl, sizeof(int)));
  }
  const int grid = a.scatter ? a.n_utts * 4 : ((a.n_utts + 7) / 8) * 32;
  if (abl || a.fault) hipLaunchKernelGGL(sanm_block_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, s, a);     // the ablation / fault-injection build (same arithmetic with no ablation bit set)
  else hipLaunchKernelGGL(sanm_block_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, s, a);
  HIP_CHECK(hipGetLastError());
}
