// One SANM encoder block (Export_SenseVoice.py:227-258: LayerNorm -> q|k|v -> soft-max attention + FSMN -> out-projection + residual
// -> LayerNorm -> FFN + residual) as ONE launch for batches of <= 144-row windows (8 s chunks): the bf16 headline path (multiplication scheme of round 4, FFN pair of round 6).
//
// Decomposition (as in round 2): 64 windows x 4 heads = 256 = one workgroup per CU; the four workgroups (w, h) of window w form a CLUSTER,
// workgroup h owns head h of the attention half and column slab h of every GEMM, and the A operand of each GEMM phase is what the cluster
// exchanged at the previous phase boundary. What changed is how a workgroup multiplies:
//
//   * 8 waves (512 threads, 256 registers per lane) instead of 12: a wave tile is 144 rows x 48 / 32 / 64 columns, so one A fragment read
//     from LDS feeds 3 / 2 / 4 MFMAs (the 48 x 32 and 48 x 64 tiles of the 12-wave kernel needed 5 reads per 6 MFMAs and ran the LDS port at 83 %).
//   * WEIGHTS NEVER TOUCH LDS. Every wave streams the fragments of its own output columns straight from global memory into registers, from a
//     fragment-major copy of the block's weights made once per session (`launch_sanm_block8_pack`: one contiguous KB per wave instruction, in
//     the order the wave consumes them). Nothing about W is shared between the waves of a workgroup, so there is nothing to stage.
//   * The A operand lives in LDS as four 36 KB CHUNKS of 128 columns ([144 rows][256 B], 16-byte slots XOR-swizzled by the row). An exchanged
//     operand arrives by LDS-DMA, one chunk per `s_barrier`; the workgroup's OWN quarter of every exchange is written into its chunk slot by
//     the producing epilogue and never comes back from memory: a phase starts multiplying its own chunk while the three foreign ones are in
//     flight. (Round-4 form of FFN-2: the 16 chunks of `hid` starting with the own four, the ring refilling behind; round-6 form: only the own four.)
//   * No barrier inside a chunk, hand-counted `s_waitcnt vmcnt(N)` for both streams (see `Sched`): the compiler's own bookkeeping gives up
//     (vmcnt(0) / lgkmcnt(0)) while an LDS-DMA is pending, so the W loads are inline asm and their waits are tied to the registers they fill.
//
//   A  q|k|v projection of head h (LayerNorm folded in) -> q / k / v^T images in LDS, attention (wave = one 16-query tile; the ninth tile is
//      shared by all waves: scores redundantly, context split over the head dimension), FSMN        ctx[:, 128 h ..] (bf16)  -- exchange 0 -->
//   B  out-projection slab (wave = K-half x 32 columns; accumulators start from FSMN term + residual)
//        -> x1 slab: f32 to memory (own rows, read back in D), bf16 copy + row statistics                                   -- exchange 1 -->
//   C  FFN-1 slab: relu(LN(x1) W1[512 h ..]^T + b1) -> hid[:, 512 h ..] (bf16), the four own chunks of phase D, in LDS
//   D  (FFNK = 1, the default since round 6) FFN-2 over the workgroup's OWN 512 hidden columns for ALL 512 output columns: hid never leaves the CU; the own 128-column slab
//      of the partial stays in LDS (f32), the three foreign slabs go out as f16 chunk images                                  -- exchange 2 -->
//      sum of the four partials (source-head order) + b2 + x1 -> x (f32), bf16 copy + row statistics of the next block        -- exchange 3 -->
//      (FFNK = 0, ASR_SANM_BLOCK_FFNK=0: the round-4 form -- hid is exchanged, FFN-2 is a column slab over the 16 chunks of hid, wave = K-half x 32 columns)
//
// Exchange protocol, give-up bound, placement and launch limits are those of round 2 (the 12-wave kernel of rounds 2-3 was deleted in round 6; DESIGN.md history table).
#include <algorithm>
#include <type_traits>
#include <utility>
#include "kernels.h"

namespace {

constexpr int R = 144, RF = 9;                  // rows / row fragments of a window tile
constexpr int HD = 128, NH = 4, D = 512, DFF = 2048;
constexpr int NW = 8, NT = NW * 64;             // waves / threads per workgroup
constexpr int TAPS = 11;
constexpr int LDS_BYTES = 160 * 1024;
// ---- LDS map (bytes)
constexpr int CH = R * 256;                      // one chunk: 144 rows x 128 bf16 columns = 36864
constexpr int NSLOT = 4;                         // chunk slots 0 .. 3 at q * CH
constexpr int QS = 0, KS = CH, KEYS = 160, VS = KS + KEYS * 256, IMG_END = VS + HD * 512;       // phase A images: q (= slot 0, later the ctx chunk), k, v^T
constexpr int TERM = CH;                         // FSMN term f32 [144][128] over slots 1, 2 (the k image and the head of v^T are dead by then)
constexpr int RED = 2 * CH;                      // K-half exchange of phases B / D over slots 2, 3 (72 KB)
constexpr int ST_F = NSLOT * CH;                 // (mean, rstd) [144] float2
constexpr int ST_P = ST_F + R * 8;               // statistics partials [4 slots][144] float2
constexpr int DUMMY = ST_P + NSLOT * R * 8;          // landing area of the L2 warm-up loads (256 B per wave, never read)
static_assert(IMG_END <= ST_F && TERM + R * HD * 4 <= ST_F && DUMMY + NW * 256 <= LDS_BYTES, "LDS map");

// ---- fragment-major weight copy of one block, bytes from the block's base (see launch_sanm_block8_pack for the element order)
constexpr size_t PK_QKV = 0, PK_QKV_WAVE = 16 * 3 * 1024;                       // [h][wave][16 steps][3 frags][64 lanes][16 B]
constexpr size_t PK_OUT = PK_QKV + (size_t)NH * NW * PK_QKV_WAVE, PK_OUT_WAVE = 8 * 2 * 1024;
constexpr size_t PK_W1 = PK_OUT + (size_t)NH * NW * PK_OUT_WAVE, PK_W1_WAVE = 16 * 4 * 1024;
constexpr size_t PK_W2 = PK_W1 + (size_t)NH * NW * PK_W1_WAVE, PK_W2_WAVE = 32 * 2 * 1024;
constexpr size_t PK_BYTES = PK_W2 + (size_t)NH * NW * PK_W2_WAVE;
static_assert(PK_BYTES == (size_t)(3 * D * D + D * D + 2 * DFF * D) * 2, "the packed copy holds every weight element once");

__device__ __host__ constexpr int frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
// write-through (sc1) stores: the payload of an exchange must be in memory, not dirty in this XCD's L2, when the flag is raised
__device__ __forceinline__ void store16_wt(void* p, uint4 v, bool plain = false) {
  if (plain) { *reinterpret_cast<uint4*>(p) = v; return; }
  const u32x4_t w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void store8_wt(void* p, float2 v, bool plain = false) {
  if (plain) { *reinterpret_cast<float2*>(p) = v; return; }
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
// fence = false (only when the whole cluster shares this XCD's L2 and the payload is then read by sc1 loads, which bypass the CU's L1): no acquire. On gfx950 the
// agent-scope acquire (buffer_inv sc1) does not only drop this CU's L1 lines, and four of them per cluster and exchange keep emptying the XCD's L2 of what its
// 32 workgroups share (weights, the cluster's exchanged lines).
__device__ __forceinline__ void consume(unsigned* flag, unsigned need, unsigned* err, bool fence = true) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(2);
      // a cluster in step waits a few us here; 2^15 polls (about 30 ms with the sleep and the L2 round trip: a launch now walks up to 49 blocks, and a
      // cluster may wait behind another session's whole launch) means the cluster is not co-resident: give up, never hang -- the host redoes the pass
      // on the four-launch path and keeps the session there for a while
      if (++spins > (1u << 15)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if ((spins & 255u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;           // somebody already gave up: the launch is void anyway
    }
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
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

// f16 images of the FFN-2 partials (FFNK form): 8 f32 -> 16 bytes (round to nearest even) and back
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint4 pack8_f16(const float (&v)[8]) {
  uint4 w;
  w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[0], v[1]}, f16x2_t)); w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[2], v[3]}, f16x2_t));
  w.z = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[4], v[5]}, f16x2_t)); w.w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[6], v[7]}, f16x2_t));
  return w;
}
__device__ __forceinline__ void add8_f16(float (&v)[8], const u32x4_t w) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2_t f = __builtin_convertvector(__builtin_bit_cast(f16x2_t, (uint32_t)w[e]), f32x2_t);
    v[2 * e] += f[0]; v[2 * e + 1] += f[1];
  }
}

template <int... I, typename F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define GLDS(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define GLDS_SC1(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 16)

// one chunk (144 rows x 256 B of a row-major bf16 matrix) -> an LDS slot. One wave instruction lands four rows; 36 pieces over 8 waves = five per
// wave (the spare slots reload the last piece: same bytes, same place), so every wave's queue sees exactly five operations per chunk.
constexpr int DMA_PER_CHUNK = 5;
__device__ __forceinline__ void issue_chunk(const unsigned char* src, int ld_bytes, int rows_left, unsigned char* slot, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < DMA_PER_CHUNK; ++t) {
    const int piece = min(wave + NW * t, R / 4 - 1);
    const int m = piece * 4 + (lane >> 4);
    GLDS(src + (size_t)min(m, rows_left - 1) * ld_bytes + (((lane & 15) ^ (m & 15)) << 4), slot + piece * 1024);
  }
}

// The cluster's exchange buffers (ctx, x1, the FFN-2 partials / hid, x) hold CHUNK IMAGES: inside the window's own rows of the buffer, chunk c is the T16 x 256-byte LDS
// image of its 128 columns, swizzled slots and all, at c * T16 * 256 bytes: a consumer's LDS-DMA of one wave instruction lands four rows = one contiguous KB. Producers
// write the image into their LDS slot (it is the own chunk of the next phase anyway) and, by default, store the same 16-byte granules to memory straight from the
// registers (write-through); `store_chunk_img` -- pieces out of the LDS image, one contiguous KB per instruction -- is the ctx image's path and opt bit 8's for the rest
// (measured 1.5 % slower there: the exchange stores are bound by the write path, not by their issue).
__device__ __forceinline__ void issue_chunk_img(const unsigned char* img, int n_pieces, unsigned char* slot, int wave, int lane, bool sc1 = false) {
#pragma unroll
  for (int t = 0; t < DMA_PER_CHUNK; ++t) {
    const int piece = min(wave + NW * t, R / 4 - 1);                       // (pieces past a short window reload its last piece into rows nobody stores)
    const unsigned char* src = img + (size_t)min(piece, n_pieces - 1) * 1024 + lane * 16;
    if (sc1) GLDS_SC1(src, slot + piece * 1024); else GLDS(src, slot + piece * 1024);
  }
}
__device__ __forceinline__ void store_chunk_img(unsigned char* img, const unsigned char* slot, int n_pieces, int wave, int lane, bool plain) {
#pragma unroll
  for (int t = 0; t < DMA_PER_CHUNK; ++t) {
    const int piece = wave + NW * t;
    if (piece < n_pieces) store16_wt(img + (size_t)piece * 1024 + lane * 16, *reinterpret_cast<const uint4*>(slot + piece * 1024 + lane * 16), plain);
  }
}

// L2 warm-up: the 32 workgroups of an XCD stream the same packed weights a phase later, and the first one to ask for a line pays the trip to HBM with
// a one-microsecond prefetch distance. Each workgroup asks for its 1 / 32 of the region a phase ahead: one dword per 128-byte line, landed in a dummy
// LDS area by LDS-DMA (no destination register to keep alive), never waited for on purpose.
__device__ __forceinline__ void l2_touch(const unsigned char* base, int n_lines, unsigned char* smem, int wave, int tid) {
  if (!base) return;
  const int part = blockIdx.x >> 3, n_parts = max(1, (int)(gridDim.x >> 3));
  const int per = (n_lines + n_parts - 1) / n_parts, l1 = min((part + 1) * per, n_lines);
  for (int l = part * per + tid; l < l1; l += NT)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)l * 128),
                                     (__attribute__((address_space(3))) void*)(smem + DUMMY + wave * 256), 4, 0, 0);
}

// ---- the in-order queue of one wave inside a chunk loop, as compile-time arithmetic. A loop walks NQ chunks, this wave multiplies TPC K-steps
// of each with NJ W fragments per step; the W fragment sets are PF deep: the PF first are prefetched by the caller, set (s % PF) is refilled
// right after the MFMAs of step s were issued (for step s + PF). Chunks below SEEDED are already in LDS; chunks SEEDED .. 3 are requested in
// front of the loop (after the prefetch), chunk q >= 4 at the start of iteration q - 3 (behind that iteration's barrier).
//   after_dma(q): operations this wave issued behind chunk q's DMA when iteration q starts  -> vmcnt bound for "chunk q has landed"
//   after_w(s):   operations issued behind the W set of step s when step s starts           -> vmcnt bound for "the set is in its registers"
// Operations the bookkeeping does not know of can only make a bound stricter than necessary, never unsafe.
template <int NJ, int NQ, int TPC, int PF, int SEEDED>
struct Sched {
  static constexpr int NS = NQ * TPC;
  static constexpr bool refill(int s) { return s >= 0 && s + PF < NS; }
  static constexpr bool dma_exists(int q) { return q >= SEEDED && q < NQ; }
  static constexpr bool inloop_dma(int it) { return it >= 1 && it + 3 >= 4 && dma_exists(it + 3); }
  static constexpr int cap(int n) { return n > 63 ? 63 : n; }
  static constexpr int refills_in_iter(int it) { int n = 0; for (int t = 0; t < TPC; ++t) n += refill(it * TPC + t) ? NJ : 0; return n; }
  static constexpr int after_dma(int q) {
    int n = 0;
    if (q < 4) {
      for (int p = q + 1; p < 4; ++p) n += dma_exists(p) ? DMA_PER_CHUNK : 0;
      for (int it = 1; it < q; ++it) n += inloop_dma(it) ? DMA_PER_CHUNK : 0;
      for (int it = 0; it < q; ++it) n += refills_in_iter(it);
    } else {
      for (int it = q - 3 + 1; it < q; ++it) n += inloop_dma(it) ? DMA_PER_CHUNK : 0;
      for (int it = q - 3; it < q; ++it) n += refills_in_iter(it);
    }
    return cap(n);
  }
  static constexpr int after_w(int s) {
    int n = 0;
    if (s < PF) {
      n += (PF - 1 - s) * NJ;
      for (int p = 0; p < 4; ++p) n += dma_exists(p) ? DMA_PER_CHUNK : 0;
      for (int p = 0; p < s; ++p) n += refill(p) ? NJ : 0;
      for (int it = 1; it <= s / TPC; ++it) n += inloop_dma(it) ? DMA_PER_CHUNK : 0;
    } else {
      for (int p = s - PF + 1; p < s; ++p) n += refill(p) ? NJ : 0;
      for (int it = (s - PF) / TPC + 1; it <= s / TPC; ++it) n += inloop_dma(it) ? DMA_PER_CHUNK : 0;
    }
    return cap(n);
  }
};

// W fragment loads by inline asm (see the header): `p` = this lane's position in the wave's stream, one set = NJ consecutive KB
#define ASR_WLOAD(J) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(set[J]) : "v"(p), "n"((J) * 1024) : "memory")
template <int NJ>
__device__ __forceinline__ void wload_set(bf16x8_t (&set)[NJ], const unsigned char* p) {
  ASR_WLOAD(0); ASR_WLOAD(1);
  if constexpr (NJ > 2) ASR_WLOAD(2);
  if constexpr (NJ > 3) ASR_WLOAD(3);
}
#undef ASR_WLOAD
template <int N> __device__ __forceinline__ void wait_set(bf16x8_t (&s)[2]) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s[0]), "+v"(s[1]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_set(bf16x8_t (&s)[3]) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_set(bf16x8_t (&s)[4]) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : "n"(N) : "memory"); }

template <int NJ, int PF>
__device__ __forceinline__ void w_prefetch(bf16x8_t (&wf)[PF][NJ], const unsigned char* wp) {
  static_for<PF>([&](auto s) __attribute__((always_inline)) { wload_set<NJ>(wf[decltype(s)::value], wp + (size_t)decltype(s)::value * NJ * 1024); });
}

// ---- acc[144][16 NJ] += chunks x W. VMASK: fragments j with bit j set multiply in the un-swapped operand order (acc[i][j][r] = C[16 i + 4 (lane >> 4) + r][lane & 15]: the
// time-contiguous V^T); the others in the swapped order (acc[i][j][r] = C[16 i + (lane & 15)][4 (lane >> 4) + r] within the fragment's 16 columns).
// kg0 = first K-step of a chunk this wave multiplies (a K-half wave: 0 or TPC); issue(q) requests chunk q into slot q & 3; gate() runs once, in front of the
// first request (the exchange wait).
// ABL: timing-only ablations (results are garbage by design): 1 no MFMA, 2 no A fragment reads after the first, 4 no W refills, 8 no chunk DMA
template <int NJ, int NQ, int TPC, int PF, int SEEDED, int VMASK, int ABL, typename IssueF, typename GateF>
__device__ __forceinline__ void chunk_gemm(unsigned char* smem, const unsigned char* wp, bf16x8_t (&wf)[PF][NJ], int kg0, int lane, f32x4_t (&acc)[RF][NJ], IssueF&& issue, GateF&& gate) {
  using S = Sched<NJ, NQ, TPC, PF, SEEDED>;
  const int frow = lane & 15, fgrp = lane >> 4;
  const unsigned char* ab[TPC];
#pragma unroll
  for (int t = 0; t < TPC; ++t) ab[t] = smem + frow * 256 + ((((kg0 + t) & 3) ^ (frow >> 2)) << 6) + ((fgrp ^ (frow & 3)) << 4);
  if constexpr (SEEDED < 4) {
    gate();
    static_for<4>([&](auto q) __attribute__((always_inline)) { if constexpr (S::dma_exists(decltype(q)::value) && !(ABL & 8)) issue(decltype(q)::value); });
  }
  static_for<NQ>([&](auto qt) __attribute__((always_inline)) {
    constexpr int q = decltype(qt)::value;
    if constexpr (S::dma_exists(q) && !(ABL & (4 | 8))) wait_vm<S::after_dma(q)>();
    __builtin_amdgcn_s_barrier();          // raw barrier (the fence of a __syncthreads would drain the queues): chunk q is in LDS for everyone, chunk q - 1 is consumed
    if constexpr (S::inloop_dma(q)) {
      if constexpr (SEEDED >= 4 && q == SEEDED - 3) gate();
      if constexpr (!(ABL & 8)) issue(q + 3);
    }
    static_for<TPC>([&](auto tt) __attribute__((always_inline)) {
      constexpr int t = decltype(tt)::value, s = q * TPC + t, set = s % PF;
      bf16x8_t af[RF];
      if constexpr (!(ABL & 2) || s == 0) {
#pragma unroll
        for (int i = 0; i < RF; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(ab[t] + (q & 3) * CH + i * 4096);
      } else {
#pragma unroll
        for (int i = 0; i < RF; ++i) af[i] = wf[set][i % NJ];
      }
      if constexpr (!(ABL & 4)) wait_set<S::after_w(s)>(wf[set]); else if constexpr (s < PF) wait_set<0>(wf[set]);
      if constexpr (ABL & 1) {
#pragma unroll
        for (int i = 0; i < RF; ++i) asm volatile("" ::"v"(af[i]));
        asm volatile("" ::"v"(wf[set][0]), "v"(wf[set][1]));
        if constexpr (NJ > 2) asm volatile("" ::"v"(wf[set][2]));
        if constexpr (NJ > 3) asm volatile("" ::"v"(wf[set][3]));
      } else {
#pragma unroll
        for (int i = 0; i < RF; ++i)
          static_for<NJ>([&](auto jt) __attribute__((always_inline)) {
            constexpr int j = decltype(jt)::value;
            if constexpr ((VMASK >> j) & 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[set][j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][j], af[i], acc[i][j], 0, 0, 0);
          });
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (S::refill(s) && !(ABL & 4)) wload_set<NJ>(wf[set], wp + (size_t)(s + PF) * NJ * 1024);
    });
  });
}

// K-half exchange of phases B / D: wave (kh, cg) holds a [144][32] partial over its half of K; kh = 0 ends up with row fragments 0 .. 4, kh = 1 with 5 .. 8 of the
// sum (always kh 0 + kh 1). `red` = 72 KB of LDS nobody reads any more.
__device__ __forceinline__ void khalf_exchange(unsigned char* red, int kh, int cg, int lane, f32x4_t (&acc)[RF][2]) {
  // (value selects, not branches over the two halves: a branch per half makes the compiler merge the two store sequences through a pointer phi, and the
  //  accumulator array then lives in scratch memory)
  unsigned char* mine = red + (size_t)cg * RF * 2048 + lane * 16;
  const bool lo = kh == 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {             // slot k: fragment 5 + k of K-half 0 (k < 4) / fragment k of K-half 1
    const int i0 = k < 4 ? 5 + k : 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4_t v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = lo ? acc[i0][j][r] : acc[k][j][r];
      if (k < 4 || !lo) *reinterpret_cast<f32x4_t*>(mine + (lo ? i0 : k) * 2048 + j * 1024) = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {             // K-half 0 adds fragments 0 .. 4 of K-half 1; K-half 1 adds fragments 5 .. 8 of K-half 0
    const int i0 = k < 4 ? 5 + k : 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(mine + (lo ? k : i0) * 2048 + j * 1024);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[k][j][r] += lo ? r4[r] : 0.0f;
        if (k < 4) acc[i0][j][r] += lo ? 0.0f : r4[r];
      }
    }
  }
}

// LayerNorm statistics of the 144 bf16 rows whose four chunks sit in the slots (chunk c in slot (c - rot) & 3). One work item = one row of one slot: sixteen
// 16-byte reads into four independent (sum, sum of squares) accumulator pairs; 576 items over 512 threads. The four partials of a row are then summed in
// CHUNK order, so the four workgroups of a cluster get the same bits whatever their rotation; (mean, rstd) -> st_fin. Called between barriers. (The row
// statistics no longer travel with the exchanges: three dependent global loads in front of the loop and the producers' 8-byte write-through stores cost
// more than this pass over LDS. v_dot2c_f32_bf16 would do it in a third of the instructions, but inline asm escapes the compiler's hazard bookkeeping for
// dependent DOT operations: measured, results depended on timing.)
__device__ __forceinline__ void row_stats_from_chunks(unsigned char* smem, int rot, float eps, int tid) {
  float2* st_part = reinterpret_cast<float2*>(smem + ST_P);
  for (int v = tid; v < NSLOT * R; v += NT) {
    const int slot = v / R, row = v - slot * R;
    const unsigned char* src = smem + slot * CH + row * 256;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const uint4 w = *reinterpret_cast<const uint4*>(src + (((p + row) & 15) << 4));       // (rows are 256 B apart: a common position would be a 16-way bank conflict)
      const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(wd[e] << 16), hi = __uint_as_float(wd[e] & 0xffff0000u);
        s1[e] += lo + hi;
        s2[e] = fmaf(lo, lo, fmaf(hi, hi, s2[e]));
      }
    }
    st_part[v] = make_float2((s1[0] + s1[1]) + (s1[2] + s1[3]), (s2[0] + s2[1]) + (s2[2] + s2[3]));
  }
  __syncthreads();
  if (tid < R) {
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int c = 0; c < NSLOT; ++c) { const float2 p = st_part[((c - rot) & 3) * R + tid]; sx += p.x; sy += p.y; }
    const float mean = sx * (1.0f / D);
    const float var = fmaxf(sy * (1.0f / D) - mean * mean, 0.0f);
    reinterpret_cast<float2*>(smem + ST_F)[tid] = make_float2(mean, rsqrtf(var + eps));
  }
  __syncthreads();
}

// ---- row statistics riding the exchanges. A producer epilogue reduces (sum, sum of squares) of its own 128 bf16-rounded columns per row -- lanes -> 32-column
// groups by two shuffles -> st_part[cg][row] -> one thread per row -> ONE float2 per row and workgroup, stored next to the payload (slot h of the row's 16-slot
// record). The consumer asks for the four records of its rows (32 B per row, one thread per row) by inline asm in front of its chunk loop and finishes them
// behind it: no load latency is exposed and nobody passes over the rows a second time.
// (By LDS-DMA into the partials area, which no producer epilogue uses at that time: registers filled by an asynchronous load must not live across a
// 230-register loop -- if the compiler spilled them it would store whatever they held before the data arrived.) Every wave issues exactly one instruction.
__device__ __forceinline__ void row_stats_request(const float2* rec0, int rows_left, unsigned char* smem, int wave, int lane, bool sc1 = false) {
  const int item = min(wave * 64 + lane, 2 * R - 1), row = item >> 1;          // item = (row, 16-byte half of the 32-byte record head)
  const unsigned char* src = reinterpret_cast<const unsigned char*>(rec0 + (size_t)min(row, rows_left - 1) * (D / 32)) + (item & 1) * 16;
  if (sc1) GLDS_SC1(src, smem + ST_P + wave * 1024); else GLDS(src, smem + ST_P + wave * 1024);
}
__device__ __forceinline__ void row_stats_finish(unsigned char* smem, float eps, int tid) {      // tid < R, behind the loop's last barrier
  const float4 a = *reinterpret_cast<const float4*>(smem + ST_P + tid * 32), b = *reinterpret_cast<const float4*>(smem + ST_P + tid * 32 + 16);
  const float sx = (a.x + a.z) + (b.x + b.z), sy = (a.y + a.w) + (b.y + b.w);
  const float mean = sx * (1.0f / D);
  const float var = fmaxf(sy * (1.0f / D) - mean * mean, 0.0f);
  reinterpret_cast<float2*>(smem + ST_F)[tid] = make_float2(mean, rsqrtf(var + eps));
}
// producer side, step 1 (inside the epilogue's row loop): this lane's 8 values -> its 32-column group's partial
__device__ __forceinline__ void row_stats_group(const uint4& pk, unsigned char* smem, int cg, int row, int fgrp) {
  float s1 = 0.0f, s2 = 0.0f;
  stats8(pk, s1, s2);
  s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
  s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
  if (fgrp == 0) reinterpret_cast<float2*>(smem + ST_P)[cg * R + row] = make_float2(s1, s2);
}
// producer side, step 2 (after a barrier): one thread per row sums the four groups and stores the record slot
__device__ __forceinline__ void row_stats_publish(unsigned char* smem, float2* rec_slot, int n_rows, bool write_through, bool plain, int tid) {
  if (tid < n_rows) {
    const float2* part = reinterpret_cast<const float2*>(smem + ST_P);
    const float2 p0 = part[tid], p1 = part[R + tid], p2 = part[2 * R + tid], p3 = part[3 * R + tid];
    const float2 v = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    if (write_through) store8_wt(rec_slot + (size_t)tid * (D / 32), v, plain); else rec_slot[(size_t)tid * (D / 32)] = v;
  }
}

// Per-phase views: the kernel arguments are re-read from the kernarg segment and the lane id is made opaque at every phase start, so the compiler cannot
// hoist a later phase's pointers / per-lane offsets above an earlier loop (they would be spilled around it, and a scratch reload next to hand-counted
// vmcnt waits costs a full drain -- or, worse, falsifies the count)
typedef const __attribute__((address_space(4))) SanmBlockArgs* KernArgs;
__device__ __forceinline__ KernArgs phase_args() {
  KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }          // wave-uniform values

constexpr int GETREG_XCC_ID = 20 | (0 << 6) | ((4 - 1) << 11);          // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4)
// all four workgroups of the cluster on this XCD? (wave-uniform: one relaxed load per wave)
__device__ __forceinline__ bool cluster_on_one_xcd(const unsigned* place) {
  const unsigned w = __builtin_amdgcn_readfirstlane(__hip_atomic_load(place, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const unsigned b = w & 0xffu;
  return b != 0u && w == b * 0x01010101u;
}
// "no fence" form of an exchange (write-through stores, sc1 loads, no acquire): only inside one XCD. opt 4: always fence. (Round 5 also measured the payload
// left dirty in the shared L2 -- ordinary stores, one acquire per consumer: 101.7 vs 96.1 us per block, profiles/r05_sanm_block_l2_exchange.txt; removed.)
__device__ __forceinline__ bool cluster_shares_l2(const unsigned* place, int opt) {
  if (opt & 4) return false;
  return cluster_on_one_xcd(place);
}
__device__ __forceinline__ bool cluster_plain_stores(const unsigned* place, int opt) {
  return (opt & 256) && cluster_shares_l2(place, opt);
}

// opt & 2048: the phase clock looks INSIDE phase A -- stamps 4..7 are taken behind the statistics, the images, the waves' own attention tiles and the shared tile
// (STAMP_A), the phase B stamps of those slots are skipped
#define STAMP(k) do { if (a->times && li == a->times_layer && threadIdx.x == 0 && !((a->opt & 2048) && (k) >= 4 && (k) <= 7)) a->times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define STAMP_A(k) do { if (a->times && li == a->times_layer && threadIdx.x == 0 && (a->opt & 2048)) a->times[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

template <int PFA, int PFB, int PFC, int PFD, int ABL, int FFNK>
__global__ __launch_bounds__(NT) void sanm_block8_kernel(const SanmBlockArgs a_byval) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid_0 = threadIdx.x, wave_0 = __builtin_amdgcn_readfirstlane(tid_0 >> 6);
  KernArgs a = phase_args();
  int lane, frow, fgrp;
  // cluster placement: workgroup b runs on XCD b % 8 (observed, not guaranteed): the four workgroups of a window get ids 8 apart
  int cl_0, h_0;
  if (a->scatter) { cl_0 = blockIdx.x >> 2; h_0 = blockIdx.x & 3; }            // test mode: a cluster spread over four XCDs
  else { const int idx = blockIdx.x >> 3; cl_0 = ((idx >> 2) << 3) + (blockIdx.x & 7); h_0 = idx & 3; }
  if (cl_0 >= a->n_utts) return;
  const UttPlan up = a->plan[a->utt0 + cl_0];
  const int T_0 = __builtin_amdgcn_readfirstlane(up.T), row0_0 = __builtin_amdgcn_readfirstlane(up.row_off);
  // placement word: byte h = this workgroup's XCD + 1. A consumer may skip the acquire (and read the payload with sc1 loads) only when all four bytes are
  // equal, i.e. every writer of the cluster shares its L2; any other reading (a byte still 0: that workgroup has not started) keeps the fenced form.
  if (tid_0 == 0) __hip_atomic_fetch_add(a->place + cl_0, ((unsigned)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xf) + 1u) << (8 * h_0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- the blocks of this launch. A window's blocks depend on nothing but the window's own cluster, so a launch walks n_layers consecutive blocks with no
  // chip-wide synchronisation at all: the fourth exchange of a block (the bf16 copy of its output + its row statistics) takes the place of the kernel boundary.
  // Clusters drift apart (their bursts of exchange traffic no longer hit the memory side together), 68 launch boundaries and their single-wave ramps go,
  // and the q|k|v projection of every block but the first starts on a chunk that is already in LDS.
#pragma unroll 1
  for (int li = 0; li < a->n_layers; ++li) {
  a = phase_args();
  // (per-block copies behind an opaque move: whatever the phases derive from them must not be hoisted out of the block loop and kept alive -- spilled -- across it)
  const int tid = opaque(tid_0);
  const int wave = opaque_s(wave_0), cl = opaque_s(cl_0), h = opaque_s(h_0), T = opaque_s(T_0), row0 = opaque_s(row0_0);
  const int n_act = (T + 15) >> 4;                        // active row fragments (cluster-uniform)
  const int rows_left = a->n_rows_alloc - row0;            // readable rows from row0 on
  const unsigned* place = a->place + cl;
  const int kh = wave >> 2, cg = wave & 3;                 // phases B / D: K-half, 32-column group
  lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
  const SanmBlockLayer* L = a->layers + li;
  unsigned* flags = a->flags + (size_t)li * a->flag_stride + (size_t)cl * 4;
  const bool first = li == 0, last = li + 1 == a->n_layers;
  STAMP(0);

  f32x4_t accb[RF][2];                                     // phase B accumulators: start from the residual (requested before the FSMN), then + FSMN term
  // ================================================================ phase A: q|k|v projection of head h, attention, FSMN
  {
    float2* st_fin = reinterpret_cast<float2*>(smem + ST_F);
    {
      const unsigned char* wp = reinterpret_cast<const unsigned char*>(L->wpack) + PK_QKV + (size_t)(h * NW + wave) * PK_QKV_WAVE + lane * 16;
      bf16x8_t wf[PFA][3];
      w_prefetch<3, PFA>(wf, wp);
      const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a->x_lo + (size_t)row0 * D);
      const bool st4 = !first || (a->st_in && a->st_in_n == 4);        // the rows' producer was this kernel: four partials per row
      f32x4_t acc[RF][3];
#pragma unroll
      for (int i = 0; i < RF; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][2] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
      // chunk q of the loop = columns 128 ((h + q) & 3) .., as in every phase: the own quarter first
      if (first) {           // the rows come from another launch: row-major, all four chunks by DMA
        if (st4) row_stats_request(a->st_in + (size_t)row0 * (D / 32), rows_left, smem, wave, lane);
        chunk_gemm<3, 4, 4, PFA, 0, 4, ABL>(smem, wp, wf, 0, lane, acc,
                                     [&](int q) __attribute__((always_inline)) { issue_chunk(xsrc + ((h + q) & 3) * 256, D * 2, rows_left, smem + q * CH, wave, lane); },
                                     []() __attribute__((always_inline)) {});
      } else {               // the previous block of this launch left its own chunk in slot 0 and the cluster's images + records behind exchange 3
        const bool nofence = cluster_shares_l2(place, a->opt);
        unsigned* pflags = flags - a->flag_stride;
        chunk_gemm<3, 4, 4, PFA, 1, 4, ABL>(smem, wp, wf, 0, lane, acc,
                                     [&](int q) __attribute__((always_inline)) { issue_chunk_img(xsrc + (size_t)((h + q) & 3) * n_act * 4096, n_act * 4, smem + q * CH, wave, lane, nofence); },
                                     [&]() __attribute__((always_inline)) {
                                       consume(pflags + 3, NH, a->err, !nofence);
                                       row_stats_request(a->st_out + (size_t)row0 * (D / 32), rows_left, smem, wave, lane, nofence);
                                     });
      }
      // epilogue constants: requested here, they arrive under the statistics pass
      const int grp = wave >> 2, sub = wave & 3;
      const int col = sub * 32 + fgrp * 8;
      const float* bp = L->bqkv + grp * D + h * HD + col;
      const float* cp = L->cqkv + grp * D + h * HD + col;
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
      const int dcol = wave * 16 + frow;
      const float bv = L->bqkv[2 * D + h * HD + dcol], cs = L->cqkv[2 * D + h * HD + dcol];
      __syncthreads();                                      // every wave is done with the chunks
      STAMP(1);
      // LayerNorm statistics of the bf16 rows: the producer's records (requested in front of the loop), the 16-slot records of a GEMM epilogue (block 1
      // behind the four-launch block 0), or -- behind a stand-alone LayerNorm -- a pass over the chunks
      if (st4) { if (tid < R) row_stats_finish(smem, a->ln_eps, tid); __syncthreads(); }
      else if (a->st_in) {           // (only the first block of a launch gets here)
        if (tid < R) {
          const float2 ss = sum_row_partials(a->st_in + (size_t)(row0 + min(tid, rows_left - 1)) * (D / 32), D / 32);
          const float mean = ss.x * (1.0f / D);
          const float var = fmaxf(ss.y * (1.0f / D) - mean * mean, 0.0f);
          st_fin[tid] = make_float2(mean, rsqrtf(var + a->ln_eps));
        }
        __syncthreads();
      } else row_stats_from_chunks(smem, h, a->ln_eps, tid);
      STAMP_A(4);
      // ---- images. Waves 0-3 hold q columns 32 (w & 3) .., waves 4-7 the k columns (swapped order, fragment pair = 8 consecutive columns per lane);
      //      every wave holds v column 16 w + (lane & 15) for rows 16 i + 4 (lane >> 4) .. + 3 (un-swapped order: time-contiguous V^T)
      {
        unsigned char* dst = smem + (grp == 0 ? QS : KS);
        const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, c8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          const int row = i * 16 + frow;
          const float2 mr = st_fin[row];
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][1][r]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] - mr.x * c8[e]) * mr.y + b8[e];
          *reinterpret_cast<uint4*>(dst + row * 256 + ((((col >> 3)) ^ (row & 15)) << 4)) = pack8(v);
        }
        unsigned char* vdst = smem + VS + dcol * 512 + (fgrp & 1) * 8;
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          float v[4] = {acc[i][2][0], acc[i][2][1], acc[i][2][2], acc[i][2][3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float2 mr = st_fin[i * 16 + fgrp * 4 + r]; v[r] = (v[r] - mr.x * cs) * mr.y + bv; }
          uint2 w;
          w.x = pack_bf16x2(v[0], v[1]);
          w.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(vdst + (((2 * i + (fgrp >> 1)) ^ (dcol & 15)) << 4)) = w;
        }
        if (tid < 2 * HD) {       // keys 144..159 of the last 32-key sub-tile: zeros so that 0 * v stays 0
          const int d = tid & (HD - 1), slot = 18 + (tid >> 7);
          *reinterpret_cast<uint4*>(smem + VS + d * 512 + ((slot ^ (d & 15)) << 4)) = make_uint4(0, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    STAMP_A(5);
    // FSMN taps of this thread's channel: requested before the attention, used after it
    float wc[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) wc[j] = L->wfsmn[(h * HD + (tid & (HD - 1))) * TAPS + j];
    const float bc = L->bfsmn[h * HD + (tid & (HD - 1))];

    // ---- attention: wave w = the 16-query tile w with all <= 160 scores in registers (one soft-max pass); the ninth tile (queries 128..143) is shared:
    //      every wave computes its scores and multiplies ONE 16-wide slice of the head dimension. A context tile goes back into the (dead) q rows of
    //      the image, which is exactly the chunk format phase B reads.
    {
      const unsigned char* Qs = smem + QS;
      const unsigned char* Ks = smem + KS;
      const unsigned char* Vs = smem + VS;
      const int fq = frow, g = fgrp;
      const int n_sub = (T + 31) >> 5;
      const bool shared_tile = T > 128;
      // (round 6: the soft-max body of kernels.hip's attention kernel -- score row i of a sub-tile's two S^T tiles is key 8 (i / 4) + (i % 4) (+ 4), so a lane group holds 8
      //  CONSECUTIVE keys and the V^T fragment is one 16-byte read; keys are masked only in the sub-tile that holds the window's end; p = exp2(fma(s, log2 e, -m log2 e)))
      auto scores = [&](const bf16x8_t (&qf)[HD / 32], bf16x8_t (&pf)[5], float& inv) {
        constexpr float LOG2E = 1.4426950408889634f;
        f32x4_t st[5][2];
#pragma unroll
        for (int s = 0; s < 5; ++s) { st[s][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; st[s][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          if (s < n_sub) {
            const int key0 = s * 32 + ((fq >> 2) << 3) + (fq & 3), key1 = key0 + 4;
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
        for (int s = 0; s < 5; ++s) {
          if (s * 32 + 32 > T) {                               // (wave-uniform: only the sub-tiles at or past the window's end mask; sub-tiles past n_sub are all -inf)
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (s * 32 + g * 8 + hlf * 4 + r >= T) st[s][hlf][r] = -INFINITY;
          }
          mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(st[s][0][0], st[s][0][1])), fmaxf(st[s][0][2], st[s][0][3])), fmaxf(fmaxf(st[s][1][0], st[s][1][1]), fmaxf(st[s][1][2], st[s][1][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const f32x2_t mneg = {-mx * LOG2E, -mx * LOG2E}, l2 = {LOG2E, LOG2E};
        f32x2_t lsum = {0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          f32x2_t p[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2_t e = __builtin_elementwise_fma(f32x2_t{st[s][q >> 1][2 * (q & 1)], st[s][q >> 1][2 * (q & 1) + 1]}, l2, mneg);
            p[q] = f32x2_t{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          }
          lsum += (p[0] + p[1]) + (p[2] + p[3]);
          union { bf16x8_t v; uint32_t w[4]; } u8;
#pragma unroll
          for (int q = 0; q < 4; ++q) u8.w[q] = pack_bf16x2(p[q][0], p[q][1]);
          pf[s] = u8.v;
        }
        float l = lsum[0] + lsum[1];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv = 1.0f / l;
      };
      // O^T[d = 16 dt + 4 g + r][q = fq] for dt in [dt0, dt0 + ND) -> context image row qrow, 16-byte chunk dt * 2 + g / 2
      auto context = [&](const bf16x8_t (&pf)[5], float inv, int qrow, auto dt0_tag, auto nd_tag) {
        constexpr int ND = decltype(nd_tag)::value;
        const int dt0 = dt0_tag;
        f32x4_t ot[ND];
#pragma unroll
        for (int e = 0; e < ND; ++e) ot[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          if (s < n_sub) {
#pragma unroll
            for (int e = 0; e < ND; ++e) {
              const int d = (dt0 + e) * 16 + fq;
              const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + d * 512 + (((s * 4 + g) ^ (d & 15)) << 4));
              ot[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s], ot[e], 0, 0, 0);
            }
          }
        }
        unsigned char* crow = smem + QS + qrow * 256;
#pragma unroll
        for (int e = 0; e < ND; ++e) {
          uint2 w;
          w.x = pack_bf16x2(ot[e][0] * inv, ot[e][1] * inv);
          w.y = pack_bf16x2(ot[e][2] * inv, ot[e][3] * inv);
          *reinterpret_cast<uint2*>(crow + ((((dt0 + e) * 2 + (g >> 1)) ^ (qrow & 15)) << 4) + (g & 1) * 8) = w;
        }
      };
      if (wave * 16 < T) {
        const int qrow = wave * 16 + fq;
        bf16x8_t qf[HD / 32], pf[5];
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + qrow * 256 + (((ks * 4 + g) ^ (qrow & 15)) << 4));
        float inv;
        scores(qf, pf, inv);
        context(pf, inv, qrow, 0, std::integral_constant<int, HD / 16>{});
      }
      STAMP_A(6);
      if (shared_tile) {
        // (the waves' own tiles are rows 0..127: nobody has written rows 128..143 yet, and nobody will before the barrier below)
        bf16x8_t qf8[HD / 32], pf[5];
        const int qrow8 = 128 + fq;
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) qf8[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + qrow8 * 256 + (((ks * 4 + g) ^ (qrow8 & 15)) << 4));
        float inv;
        scores(qf8, pf, inv);
        __syncthreads();                                    // every wave holds the shared tile's q fragments and scores: its rows may now take the context
        context(pf, inv, 128 + fq, wave, std::integral_constant<int, 1>{});
      }
      STAMP_A(7);
    }
    __syncthreads();
    STAMP(2);
    // ---- context rows of this head -> memory (whole 16-byte chunks, write-through) and count the workgroup in on exchange 0; alignment rows past the
    //      window become zeros in memory AND in the image (the own chunk of phase B must equal what the other heads read back)
    {
      const bool plain = cluster_plain_stores(place, a->opt);      // tuning switch: payload left in the shared L2 (ordinary stores)
      if (a->times && tid == 0) a->times[(size_t)blockIdx.x * 16 + 15] = plain ? 1ull : 0ull;
      const int T16 = n_act * 16;
      for (int c = T * 16 + tid; c < T16 * 16; c += NT) *reinterpret_cast<uint4*>(smem + QS + c * 16) = make_uint4(0, 0, 0, 0);
      __syncthreads();
      store_chunk_img(reinterpret_cast<unsigned char*>(a->ctx + (size_t)row0 * D) + (size_t)h * T16 * 256, smem + QS, n_act * 4, wave, lane, plain);
    }
    STAMP(3);               // (the count-in on exchange 0 waits until the FSMN below is done: the stores drain under it)
    // the residual rows of this workgroup's slab depend on nobody else: request them now (K-half 0 starts from them), they arrive under the FSMN
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      if (kh == 0) {
        const float* xr = a->x + (size_t)(row0 + min(i * 16 + frow, rows_left - 1)) * D + h * HD + cg * 32 + fgrp * 8;
        accb[i][0] = *reinterpret_cast<const f32x4_t*>(xr); accb[i][1] = *reinterpret_cast<const f32x4_t*>(xr + 4);
      } else { accb[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accb[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    }
    // ---- FSMN memory (thread = channel x time segment of 40 / 40 / 32 / 32 steps, 11 taps from the V^T image) -> registers; once every thread is done
    //      with the image, -> f32 [144][128] at TERM, 32-byte granules XOR-swizzled by the row so that the phase B prologue reads without conflicts
    {
      constexpr int PAD = (TAPS - 1) / 2;
      const int c = tid & (HD - 1), seg = tid >> 7;
      const int t0 = seg < 2 ? seg * 40 : 80 + (seg - 2) * 32, len = seg < 2 ? 40 : 32;
      const unsigned char* vrow = smem + VS + c * 512;
      float x[56];                                           // time steps t0 - 8 .. t0 + 47
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int sl = (t0 >> 3) - 1 + q;
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
      float y[40];
#pragma unroll
      for (int i = 0; i < 40; ++i) {
        float accv = bc;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) accv = fmaf(wc[j], x[8 + i + j - PAD], accv);
        y[i] = accv;
      }
      __syncthreads();                                        // the V^T image is dead
#pragma unroll
      for (int i = 0; i < 40; ++i) {
        const int t = t0 + i;
        if (i < len) *reinterpret_cast<float*>(smem + TERM + t * 512 + (((c >> 3) ^ (t & 15)) << 5) + (c & 7) * 4) = (t < T) ? y[i] : 0.0f;
      }
    }
    publish(flags + 0, first && a->fault != 0 && blockIdx.x == 5);     // (its barrier also closes the term image)
  }

  // ================================================================ phase B: out-projection slab + FSMN term + residual -> x1
  {
    a = phase_args();
    L = a->layers + li;
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    STAMP(4);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(L->wpack) + PK_OUT + (size_t)(h * NW + wave) * PK_OUT_WAVE + lane * 16;
    bf16x8_t wf[PFB][2];
    w_prefetch<2, PFB>(wf, wp);
    if (kh == 0) {
      const int n = cg * 32 + fgrp * 8;
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const int row = i * 16 + frow;
        const unsigned char* mrow = smem + TERM + row * 512 + (((n >> 3) ^ (row & 15)) << 5);
        accb[i][0] += *reinterpret_cast<const f32x4_t*>(mrow); accb[i][1] += *reinterpret_cast<const f32x4_t*>(mrow + 16);
      }
    }
    __syncthreads();                                          // the term is in registers: slots 1..3 may take the other heads' context chunks
    const unsigned char* csrc = reinterpret_cast<const unsigned char*>(a->ctx + (size_t)row0 * D);
    const bool nofence = cluster_shares_l2(place, a->opt);      // (every workgroup of the cluster has published its placement byte long before it can publish an exchange)
    chunk_gemm<2, 4, 2, PFB, 1, 0, ABL>(smem, wp, wf, 2 * kh, lane, accb,
                                 [&](int q) __attribute__((always_inline)) { issue_chunk_img(csrc + (size_t)((h + q) & 3) * n_act * 4096, n_act * 4, smem + q * CH, wave, lane, nofence); },
                                 [&]() __attribute__((always_inline)) { consume(flags + 0, NH, a->err, !nofence); STAMP(5); });
    __syncthreads();
    STAMP(6);
    khalf_exchange(smem + RED, kh, cg, lane, accb);
    // x1 rows: f32 -> memory (this workgroup's own slab: phase D reads it back), bf16 -> exchange 1 + the own chunk of phase C (slot 0), row statistics
    {
      const bool plain = cluster_plain_stores(place, a->opt);      // tuning switch: payload left in the shared L2 (ordinary stores)
      const int n = cg * 32 + fgrp * 8;
      float* xo = a->x + (size_t)row0 * D + h * HD + n;
      const bool direct = (a->opt & 8) == 0;                  // payload rows go out of the registers (measured 1.5 % faster than image pieces out of LDS: the stores are bound by the write path, not by their issue); opt 8: through the LDS image
      unsigned char* x1img = reinterpret_cast<unsigned char*>(a->x1_lo + (size_t)row0 * D) + (size_t)h * n_act * 4096;
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        if ((kh == 0) == (i < 5) && i < n_act) {
          const int row = i * 16 + frow;
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = accb[i][0][r]; v[4 + r] = accb[i][1][r]; }
          *reinterpret_cast<float4*>(xo + (size_t)row * D) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(xo + (size_t)row * D + 4) = make_float4(v[4], v[5], v[6], v[7]);
          const uint4 pk = pack8(v);
          *reinterpret_cast<uint4*>(smem + row * 256 + ((((n >> 3)) ^ (row & 15)) << 4)) = pk;
          if (direct) store16_wt(x1img + row * 256 + ((((n >> 3)) ^ (row & 15)) << 4), pk, plain);
          row_stats_group(pk, smem, cg, row, fgrp);
        }
      }
      __syncthreads();
      if (!direct) store_chunk_img(x1img, smem, n_act * 4, wave, lane, plain);
      row_stats_publish(smem, a->st1 + (size_t)row0 * (D / 32) + h, n_act * 16, true, plain, tid);
    }
    publish(flags + 1);
    STAMP(7);
  }

  // ================================================================ phase C: FFN-1 slab, LayerNorm folded in -> hid slab (exchange 2 + the four own chunks of phase D)
  {
    a = phase_args();
    L = a->layers + li;
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    float2* st_fin = reinterpret_cast<float2*>(smem + ST_F);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(L->wpack) + PK_W1 + (size_t)(h * NW + wave) * PK_W1_WAVE + lane * 16;
    bf16x8_t wf[PFC][4];
    w_prefetch<4, PFC>(wf, wp);
    f32x4_t acc[RF][4];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a->x1_lo + (size_t)row0 * D);
    const bool nofence = cluster_shares_l2(place, a->opt);
    chunk_gemm<4, 4, 4, PFC, 1, 0, ABL>(smem, wp, wf, 0, lane, acc,
                                 [&](int q) __attribute__((always_inline)) { issue_chunk_img(xsrc + (size_t)((h + q) & 3) * n_act * 4096, n_act * 4, smem + q * CH, wave, lane, nofence); },
                                 [&]() __attribute__((always_inline)) {
                                   consume(flags + 1, NH, a->err, !nofence);
                                   STAMP(8);
                                   row_stats_request(a->st1 + (size_t)row0 * (D / 32), rows_left, smem, wave, lane, nofence);
                                 });
    __syncthreads();                                          // every wave is done multiplying the x1 chunks; the statistics records have landed
    if (tid < R) row_stats_finish(smem, a->ln_eps, tid);
    __syncthreads();
    STAMP(9);
    const bool plain = cluster_plain_stores(place, a->opt);      // tuning switch: payload left in the shared L2 (ordinary stores)
    const bool direct = (a->opt & 8) == 0;
    unsigned char* himg = reinterpret_cast<unsigned char*>(a->hid + (size_t)row0 * DFF) + (size_t)(4 * h) * n_act * 4096;
    {
      unsigned char* slot = smem + (wave >> 1) * CH;           // this wave's 64 hidden columns = half of own chunk wave / 2
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = h * 512 + wave * 64 + p * 32 + fgrp * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(L->b1 + n), b1v = *reinterpret_cast<const float4*>(L->b1 + n + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(L->c1 + n), c1v = *reinterpret_cast<const float4*>(L->c1 + n + 4);
        const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
        const float c8[8] = {c0.x, c0.y, c0.z, c0.w, c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          const int row = i * 16 + frow;
          const float2 mr = st_fin[row];
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r]; v[4 + r] = acc[i][2 * p + 1][r]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf((v[e] - mr.x * c8[e]) * mr.y + b8[e], 0.0f);
          const uint4 pk = pack8(v);
          *reinterpret_cast<uint4*>(slot + row * 256 + (((8 * (wave & 1) + 4 * p + fgrp) ^ (row & 15)) << 4)) = pk;
          if constexpr (!FFNK) { if (direct && i < n_act) store16_wt(himg + (size_t)(wave >> 1) * n_act * 4096 + row * 256 + (((8 * (wave & 1) + 4 * p + fgrp) ^ (row & 15)) << 4), pk, plain); }
        }
      }
    }
    if constexpr (!FFNK) {
      if (!direct) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) store_chunk_img(himg + (size_t)c * n_act * 4096, smem + c * CH, n_act * 4, wave, lane, plain);
      }
      STAMP(10);
      publish(flags + 2);
    } else {
      STAMP(10);
      __syncthreads();                                         // FFNK: hid never leaves the CU -- the four own chunks are phase D's whole A operand
    }
    STAMP(11);
  }

  if constexpr (FFNK) {
  // ================================================================ phase D, K-split form (round 6): FFN-2 over the workgroup's OWN 512 hidden columns for ALL 512 output
  // columns -- hid (590 KB per window) never crosses the fabric; what does is the [144][512] partial: the three foreign 128-column slabs go out as f16 chunk images
  // (110 KB per workgroup out and in, against 147 KB out + 442 KB in for hid), the sum of the four partials + bias + x1 is this workgroup's slab of the block output.
  // f16, not bf16: three more mantissa bits keep the hand-over below the bf16 rounding the path already has (a partial is |x| ~ 1..100, far inside f16's range).
  // r04's probe of this split exchanged f32 partials (221 KB out per workgroup: write-path-bound, slower than hid); r05's 2 x 2 form added a row split that doubled
  // the W stream. This form keeps phase C's tiles (144 x 64 per wave) and phase C's W rate.
  {
    a = phase_args();
    L = a->layers + li;
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(L->wpack) + PK_W2 + (size_t)(h * NW + wave) * PK_W2_WAVE + lane * 16;
    bf16x8_t wf[PFC][4];
    w_prefetch<4, PFC>(wf, wp);
    f32x4_t acc[RF][4];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    chunk_gemm<4, 4, 4, PFC, 4, 0, ABL>(smem, wp, wf, 0, lane, acc, [&](int) __attribute__((always_inline)) {}, [&]() __attribute__((always_inline)) {});
    STAMP(12);
    __syncthreads();                                          // every wave is done reading the hid chunks: slots 2, 3 take the own slab, slot 0 the block output
    // ---- partial out: this wave holds output columns 64 wave .. + 63 = slab (wave >> 1); the own slab -> LDS (f32, 32-byte granules swizzled by the row), the others ->
    //      image (destination slab, source h) of the exchange buffer, f16, write-through
    const int dst_h = wave >> 1;
    const bool plain = cluster_plain_stores(place, a->opt);
    // opt 512 (tuning): the partial images by ORDINARY stores when the cluster shares an XCD -- unlike the all-gathered payloads an image has exactly one reader, so a dirty line is read once
    const bool plain_p = plain || ((a->opt & 512) && cluster_shares_l2(place, a->opt));
    unsigned char* pimg = reinterpret_cast<unsigned char*>(a->hid + (size_t)row0 * DFF) + (size_t)(4 * dst_h + h) * n_act * 4096;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const int row = i * 16 + frow;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r]; v[4 + r] = acc[i][2 * p + 1][r]; }
        const int gi = 8 * (wave & 1) + 4 * p + fgrp;          // 8-column group inside the 128-column slab
        if (dst_h == h) {
          float* o = reinterpret_cast<float*>(smem + RED + row * 512 + ((gi ^ (row & 15)) << 5));
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (i < n_act) {
          store16_wt(pimg + row * 256 + ((gi ^ (row & 15)) << 4), pack8_f16(v), plain_p);
        }
      }
    }
    // the x1 rows this wave finishes (fragments 0..4 of K-half 0, 5..8 of K-half 1; written by this very lane in phase B): they arrive under the publish
    float4 x1r[5][2];
    {
      const float* xr0 = a->x + (size_t)row0 * D + h * HD + cg * 32 + fgrp * 8;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int i = kh == 0 ? k : min(5 + k, RF - 1);
        const float* xr = xr0 + (size_t)min(i * 16 + frow, rows_left - 1) * D;
        x1r[k][0] = *reinterpret_cast<const float4*>(xr); x1r[k][1] = *reinterpret_cast<const float4*>(xr + 4);
      }
    }
    publish(flags + 2);                                       // (its barrier also closes the own slab in LDS)
    STAMP(13);
    const bool nofence = cluster_shares_l2(place, a->opt);
    consume(flags + 2, NH, a->err, !nofence);
    // ---- the three foreign partials of the own slab: 16 bytes per (image, row fragment), sc1 loads (they bypass this CU's L1), one wait for all of them
    const int n = cg * 32 + fgrp * 8, gi = cg * 4 + fgrp;
    u32x4_t fp[3][5];
    {
      const unsigned char* own = reinterpret_cast<const unsigned char*>(a->hid + (size_t)row0 * DFF) + (size_t)(4 * h) * n_act * 4096;
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const unsigned char* img = own + (size_t)((h + 1 + s3) & 3) * n_act * 4096;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int i = kh == 0 ? k : min(5 + k, RF - 1);
          const int row = min(i * 16 + frow, n_act * 16 - 1);
          const unsigned char* src = img + row * 256 + ((gi ^ (row & 15)) << 4);
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(fp[s3][k]) : "v"(src) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(fp[0][0]), "+v"(fp[0][1]), "+v"(fp[0][2]), "+v"(fp[0][3]), "+v"(fp[0][4]), "+v"(fp[1][0]), "+v"(fp[1][1]), "+v"(fp[1][2]), "+v"(fp[1][3]),
                   "+v"(fp[1][4]), "+v"(fp[2][0]), "+v"(fp[2][1]), "+v"(fp[2][2]), "+v"(fp[2][3]), "+v"(fp[2][4]) :: "memory");
    }
    const float4 b0 = *reinterpret_cast<const float4*>(L->b2 + h * HD + n), b1v = *reinterpret_cast<const float4*>(L->b2 + h * HD + n + 4);
    const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
    float* xo = a->x + (size_t)row0 * D + h * HD + n;
    bf16_t* xl = a->x_lo_out + (size_t)row0 * D + h * HD + n;
    unsigned char* ximg = reinterpret_cast<unsigned char*>(a->x_lo_out + (size_t)row0 * D) + (size_t)h * n_act * 4096;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = kh == 0 ? k : 5 + k;
      if (i < n_act && i < RF) {
        const int row = i * 16 + frow;
        const float* o = reinterpret_cast<const float*>(smem + RED + row * 512 + ((gi ^ (row & 15)) << 5));
        const float4 o0 = *reinterpret_cast<const float4*>(o), o1 = *reinterpret_cast<const float4*>(o + 4);
        const float4 r0 = x1r[k][0], r1 = x1r[k][1];
        float v[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        // partials are added in SOURCE-head order (the own one where its head comes), so the four workgroups of a cluster sum in one order whatever h is
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src_h = 0; src_h < NH; ++src_h) {
          if (src_h == h) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc8[e] += v[e];
          } else {
            const int s3 = (src_h - h - 1) & 3;
            add8_f16(acc8, s3 == 0 ? fp[0][k] : s3 == 1 ? fp[1][k] : fp[2][k]);
          }
        }
        const float x1[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc8[e] + (b8[e] + x1[e]);
        *reinterpret_cast<float4*>(xo + (size_t)row * D) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(xo + (size_t)row * D + 4) = make_float4(v[4], v[5], v[6], v[7]);
        const uint4 pk = pack8(v);
        if (last) *reinterpret_cast<uint4*>(xl + (size_t)row * D) = pk;
        else {
          const int pos = (((n >> 3)) ^ (row & 15)) << 4;
          *reinterpret_cast<uint4*>(smem + row * 256 + pos) = pk;
          store16_wt(ximg + row * 256 + pos, pk, plain);
        }
        row_stats_group(pk, smem, cg, row, fgrp);
      }
    }
    __syncthreads();
    row_stats_publish(smem, a->st_out + (size_t)row0 * (D / 32) + h, n_act * 16, !last, plain, tid);
    if (!last) publish(flags + 3);
    else if (a->times) wait_vm<0>();
    STAMP(14);
  }
  } else {
  // ================================================================ phase D: FFN-2 slab + bias + x1 -> x (f32), bf16 copy + row statistics
  {
    a = phase_args();
    L = a->layers + li;
    lane = opaque(tid & 63); frow = lane & 15; fgrp = lane >> 4;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(L->wpack) + PK_W2 + (size_t)(h * NW + wave) * PK_W2_WAVE + lane * 16;
    bf16x8_t wf[PFD][2];
    w_prefetch<2, PFD>(wf, wp);
    f32x4_t acc[RF][2];
#pragma unroll
    for (int i = 0; i < RF; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* hsrc = reinterpret_cast<const unsigned char*>(a->hid + (size_t)row0 * DFF);
    const bool nofence = cluster_shares_l2(place, a->opt);
    chunk_gemm<2, 16, 2, PFD, 4, 0, ABL>(smem, wp, wf, 2 * kh, lane, acc,
                                  [&](int q) __attribute__((always_inline)) { issue_chunk_img(hsrc + (size_t)((4 * h + q) & 15) * n_act * 4096, n_act * 4, smem + (q & 3) * CH, wave, lane, nofence); },
                                  [&]() __attribute__((always_inline)) { consume(flags + 2, NH, a->err, !nofence); STAMP(12); });
    // the x1 rows this wave finishes (fragments 0..4 of K-half 0, 5..8 of K-half 1; written by this very lane in phase B): requested now, they arrive under
    // the K-half exchange (in front of the loop they would have to live across it: measured, the compiler spills them)
    float4 x1r[5][2];
    {
      const float* xr0 = a->x + (size_t)row0 * D + h * HD + cg * 32 + fgrp * 8;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int i = kh == 0 ? k : min(5 + k, RF - 1);
        const float* xr = xr0 + (size_t)min(i * 16 + frow, rows_left - 1) * D;
        x1r[k][0] = *reinterpret_cast<const float4*>(xr); x1r[k][1] = *reinterpret_cast<const float4*>(xr + 4);
      }
    }
    __syncthreads();
    STAMP(13);
    khalf_exchange(smem + RED, kh, cg, lane, acc);
    const int n = cg * 32 + fgrp * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(L->b2 + h * HD + n), b1v = *reinterpret_cast<const float4*>(L->b2 + h * HD + n + 4);
    const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
    float* xo = a->x + (size_t)row0 * D + h * HD + n;
    bf16_t* xl = a->x_lo_out + (size_t)row0 * D + h * HD + n;
    const bool plain_d = cluster_plain_stores(place, a->opt);
    unsigned char* ximg = reinterpret_cast<unsigned char*>(a->x_lo_out + (size_t)row0 * D) + (size_t)h * n_act * 4096;        // (x_lo == x_lo_out inside a multi-block launch)
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      if ((kh == 0) == (i < 5) && i < n_act) {
        const int row = i * 16 + frow;
        const float4 r0 = x1r[i < 5 ? i : i - 5][0], r1 = x1r[i < 5 ? i : i - 5][1];
        const float x1[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][1][r]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e] + x1[e];
        *reinterpret_cast<float4*>(xo + (size_t)row * D) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(xo + (size_t)row * D + 4) = make_float4(v[4], v[5], v[6], v[7]);
        const uint4 pk = pack8(v);
        if (last) *reinterpret_cast<uint4*>(xl + (size_t)row * D) = pk;                      // for another launch: row-major, ordinary stores
        else {                                                                                 // for the next block of this launch: the own chunk (slot 0) + the image, exchange 3
          const int pos = (((n >> 3)) ^ (row & 15)) << 4;
          *reinterpret_cast<uint4*>(smem + row * 256 + pos) = pk;
          store16_wt(ximg + row * 256 + pos, pk, plain_d);
        }
        row_stats_group(pk, smem, cg, row, fgrp);
      }
    }
    __syncthreads();
    row_stats_publish(smem, a->st_out + (size_t)row0 * (D / 32) + h, n_act * 16, !last, plain_d, tid);
    if (!last) publish(flags + 3);
    else if (a->times) wait_vm<0>();
    STAMP(14);
  }
  }        // FFNK == 0: hid exchanged (round-4 form)
  }        // blocks of this launch
}

// ---- fragment-major copy of one block's weights (bf16, torch Linear layout [N][K], K contiguous). One thread = one 16-byte lane slot of one fragment.
// A swapped-order fragment of 16 columns: lane (fr = lane & 15, g = lane >> 4) holds W[n0 + perm(fr)][k0 + 8 g .. + 7]; perm pairs two fragments so that a
// lane ends with 8 consecutive output columns (frag_col); the un-swapped V fragment holds W[n0 + fr][...].
struct PackSrc { const bf16_t* wqkv; const bf16_t* wout; const bf16_t* w1; const bf16_t* w2; };
__global__ void sanm_block8_pack_kernel(PackSrc src, unsigned char* dst, int ffnk) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // 16-byte slot of the packed copy
  if (slot * 16 >= PK_BYTES) return;
  const size_t byte = slot * 16;
  const int lane = (int)(slot & 63), fr = lane & 15, g = lane >> 4;
  const bf16_t* w;
  int row, k0, ld;
  if (byte < PK_OUT) {                                  // [h][wave][16 steps][3 frags]: fragments 0, 1 = q (waves 0-3) / k (waves 4-7) columns 32 (wave & 3) .., fragment 2 = v columns 16 wave ..
    const size_t e = (byte - PK_QKV) / 1024;
    const int j = (int)(e % 3), s = (int)((e / 3) % 16), wave = (int)((e / 48) % NW), h = (int)(e / (48 * NW));
    w = src.wqkv; ld = D; k0 = 128 * ((h + (s >> 2)) & 3) + 32 * (s & 3) + 8 * g;        // chunk order (h + q) & 3, as in every phase
    if (j < 2) row = (wave >> 2) * D + h * HD + (wave & 3) * 32 + frag_col(j, fr);
    else row = 2 * D + h * HD + wave * 16 + fr;
  } else if (byte < PK_W1) {                            // [h][wave = (kh, cg)][8 steps = chunk order (h + q) & 3, K-steps 2 kh + t][2 frags]
    const size_t e = (byte - PK_OUT) / 1024;
    const int j = (int)(e % 2), s = (int)((e / 2) % 8), wave = (int)((e / 16) % NW), h = (int)(e / (16 * NW));
    const int kh = wave >> 2, cg = wave & 3, q = s >> 1, t = s & 1;
    w = src.wout; ld = D; k0 = 128 * ((h + q) & 3) + 32 * (2 * kh + t) + 8 * g;
    row = h * HD + cg * 32 + frag_col(j, fr);
  } else if (byte < PK_W2) {                            // [h][wave][16 steps = chunk order (h + q) & 3, K-steps t][4 frags]: hidden columns 512 h + 64 wave ..
    const size_t e = (byte - PK_W1) / 1024;
    const int j = (int)(e % 4), s = (int)((e / 4) % 16), wave = (int)((e / 64) % NW), h = (int)(e / (64 * NW));
    const int q = s >> 2, t = s & 3;
    w = src.w1; ld = D; k0 = 128 * ((h + q) & 3) + 32 * t + 8 * g;
    row = h * 512 + wave * 64 + frag_col(j, fr);
  } else if (ffnk) {                                    // K-split FFN-2: [h][wave][16 steps = own hid chunk q, K-step t][4 frags]: K = hidden 512 h + 128 q + 32 t .., output columns 64 wave ..
    const size_t e = (byte - PK_W2) / 1024;
    const int j = (int)(e % 4), s = (int)((e / 4) % 16), wave = (int)((e / 64) % NW), h = (int)(e / (64 * NW));
    const int q = s >> 2, t = s & 3;
    w = src.w2; ld = DFF; k0 = 512 * h + 128 * q + 32 * t + 8 * g;
    row = wave * 64 + frag_col(j, fr);
  } else {                                              // [h][wave = (kh, cg)][32 steps = chunk order (4 h + q) & 15, K-steps 2 kh + t][2 frags]
    const size_t e = (byte - PK_W2) / 1024;
    const int j = (int)(e % 2), s = (int)((e / 2) % 32), wave = (int)((e / 64) % NW), h = (int)(e / (64 * NW));
    const int kh = wave >> 2, cg = wave & 3, q = s >> 1, t = s & 1;
    w = src.w2; ld = DFF; k0 = 128 * ((4 * h + q) & 15) + 32 * (2 * kh + t) + 8 * g;
    row = h * HD + cg * 32 + frag_col(j, fr);
  }
  *reinterpret_cast<uint4*>(dst + byte) = *reinterpret_cast<const uint4*>(w + (size_t)row * ld + k0);
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

size_t sanm_block8_pack_bytes() { return PK_BYTES; }

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

void launch_sanm_block8_pack(const bf16_t* wqkv, const bf16_t* wout, const bf16_t* w1, const bf16_t* w2, void* dst, bool ffnk, hipStream_t s) {
  const PackSrc src{wqkv, wout, w1, w2};
  const unsigned n = (unsigned)(PK_BYTES / 16);
  hipLaunchKernelGGL(sanm_block8_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, reinterpret_cast<unsigned char*>(dst), ffnk ? 1 : 0);
  HIP_CHECK(hipGetLastError());
}

void launch_sanm_block8(const SanmBlockArgs& a, hipStream_t s) {
  ASR_REQUIRE(a.n_utts > 0 && a.n_utts <= sanm_block_max_utts(), "sanm_block8: %d windows per launch (max %d)", a.n_utts, sanm_block_max_utts());
  ASR_REQUIRE(a.layers && a.place && a.n_layers >= 1 && (a.n_layers == 1 || a.x_lo == a.x_lo_out) && a.x_lo && a.x && a.ctx && a.x1_lo && a.st1 && a.hid && a.x_lo_out && a.st_out && a.flags && a.err && a.plan, "sanm_block8: null buffer");
  // opt bit 2: deeper W fragment queues; opt bits 4..7: a timing-only ablation of the GEMM loops (chunk_gemm's ABL)
  typedef void (*Kern)(const SanmBlockArgs);
  static const Kern kerns[] = {sanm_block8_kernel<3, 4, 3, 4, 0, 0>, sanm_block8_kernel<6, 8, 4, 8, 0, 0>, sanm_block8_kernel<3, 4, 3, 4, 1, 0>, sanm_block8_kernel<3, 4, 3, 4, 2, 0>,
                               sanm_block8_kernel<3, 4, 3, 4, 4, 0>, sanm_block8_kernel<3, 4, 3, 4, 8, 0>, sanm_block8_kernel<3, 4, 3, 4, 12, 0>, sanm_block8_kernel<3, 4, 3, 4, 14, 0>,
                               sanm_block8_kernel<3, 4, 3, 4, 0, 1>};
  static PerDeviceOnce attr_once;
  if (attr_once.first())
    for (Kern k : kerns) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int grid = a.scatter ? a.n_utts * 4 : ((a.n_utts + 7) / 8) * 32;
  const int abl = (a.opt >> 4) & 15;
  const Kern k = a.ffnk ? kerns[8] : abl == 1 ? kerns[2] : abl == 2 ? kerns[3] : abl == 4 ? kerns[4] : abl == 8 ? kerns[5] : abl == 12 ? kerns[6] : abl == 14 ? kerns[7] : (a.opt & 2) ? kerns[1] : kerns[0];
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), LDS_BYTES, s, a);
  HIP_CHECK(hipGetLastError());
}
