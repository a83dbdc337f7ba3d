// FFN-half probe, second form, for the 8-wave SANM block design (standalone; not part of libasr_mi355x.so): the whole FFN half of a block with the
// REAL cluster exchange, two candidate routes for FFN-2, 256 workgroups = 64 clusters of four, looped over `iters` pseudo-layers.
//   F1 (both routes)  x1[144][512] bf16 DMA'd whole into LDS; hid slab = relu(x1 W1[512 h ..]^T), wave tile 144 x 64, W fragments straight from global
//   route K           hid slab -> LDS image; FFN-2 K-split: part[144][512] = hid_h W2[:, 512 h ..]^T; the three foreign quarters leave as f32
//                     (write-through), the own quarter waits in LDS; reduce-scatter: out slab = sum of four quarters (+ residual) -> x, x_lo, stats
//   route H           hid slab -> memory (write-through, bf16) -> exchange -> hid[144][2048] streamed through four 36 KB LDS chunks under FFN-2
//                     (wave = K-half x 32 columns, tile 144 x 32), K-halves summed through LDS -> x, x_lo, stats
//   hipcc --offload-arch=gfx950 -O3 -o ffn8_probe2 ffn8_probe2.hip && ./ffn8_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <type_traits>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef uint16_t bf16_t;

constexpr int R = 144, RF = 9, D = 512, DFF = 2048, NH = 4;
constexpr int NW = 8, NT = NW * 64;
constexpr int LDS_BYTES = 160 * 1024;
constexpr int IMG = 0;                          // A image: [144][512] bf16, 1 KB rows
constexpr int ST_F = 147456;                    // (mean, rstd)[144] float2 (unused by the probe's arithmetic, kept for the budget)
constexpr int CH_COLS = 128, CH_BYTES = R * CH_COLS * 2, NCH = DFF / CH_COLS, NBUF = 4;      // route H: 36 KB chunks, four buffers

struct Args {
  const bf16_t* x1;        // [layers][windows][144][512]
  const bf16_t* w1f;       // [layers][4 h][8 w][16 ks][4 j][64 lane][8]
  const bf16_t* w2f;       // route K: same shape (K = slab h); route H: [layers][4 h][8 w][32 step][2 j][64][8]
  bf16_t* hid;             // route H: [2][windows][144][2048]
  float* part;             // route K: [2][windows][4 src h][144][512] (the own quarter is never written)
  float* x;                // [windows][144][512] residual stream (read + written)
  bf16_t* x_lo;            // [windows][144][512]
  unsigned* flags;         // [iters][windows][4]
  unsigned* err;
  unsigned long long* times;
  int layers, iters, plain;     // plain: exchange payload with ordinary stores (only valid when a cluster shares an XCD)
};

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 w;
  w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]); w.z = pack_bf16x2(v[4], v[5]); w.w = pack_bf16x2(v[6], v[7]);
  return w;
}
template <bool PLAIN> __device__ __forceinline__ void store16_x(void* p, uint4 v) {
  if (PLAIN) { *reinterpret_cast<uint4*>(p) = v; return; }
  const u32x4_t w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void publish(unsigned* flag) {
  wait_vm<0>();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void consume(unsigned* flag, unsigned need, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 16)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

#define GLDS(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// ---- C[144][16 NJ] += A (LDS image, PITCH-byte rows, 16-byte chunk c of row m at position c ^ (m & 15)) x W (this wave's fragment stream: NJ KB per K-step),
// K-steps kg0 .. kg0 + NKS - 1 of the image. Fragment reads are software-pipelined in two row groups (5 + 4): while one group multiplies the other group's
// reads are in flight. W sets: PF in flight, set s refilled right after its MFMAs were issued.
// Resident images (PITCH 1024) keep row m = 16 i + f at slot 9 f + i, so that the nine row fragments of a lane sit 1 KB apart (immediate offsets);
// chunk images (PITCH 256) are row-major (a DMA instruction lands four consecutive rows).
__device__ __forceinline__ int img_row(int m) { return ((m & 15) * RF + (m >> 4)) * 1024; }
template <int NJ, int NKS, int PITCH, int PF>
struct GemmCore {
  static constexpr int FRAG_STRIDE = PITCH == 1024 ? 1024 : 16 * PITCH, LANE_STRIDE = PITCH == 1024 ? RF * 1024 : PITCH;
  const unsigned char* ab[4];          // lane bases, one per (kg & 3)
  const unsigned char* wbase;          // wave-uniform fragment stream
  unsigned lane16;
  bf16x8_t wf[PF][NJ];
  // kg_base: first image K-step of this wave (a multiple of 4, or NKS <= 4 - kg_base % 4): ab[q] addresses K-step kg_base + q
  __device__ __forceinline__ void init(const unsigned char* img, const unsigned char* wstream, int lane, int kg_base = 0) {
    const int frow = lane & 15, fgrp = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ab[q] = img + frow * LANE_STRIDE + ((((kg_base + q) & 3) ^ (frow >> 2)) << 6) + ((kg_base + q) >> 2) * 256 + ((fgrp ^ (frow & 3)) << 4);
    wbase = wstream;
    lane16 = (unsigned)lane * 16u;
  }
  __device__ __forceinline__ bf16x8_t wload(int step, int j) const { return *reinterpret_cast<const bf16x8_t*>(wbase + (size_t)(step * NJ + j) * 1024 + lane16); }
  __device__ __forceinline__ void prefetch(int first_step, int n_total) {     // the first PF sets
#pragma unroll
    for (int s = 0; s < PF; ++s)
      if (first_step + s < n_total)
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[s][j] = wload(first_step + s, j);
  }
  __device__ __forceinline__ bf16x8_t aread(int t, int i, unsigned off) const { return *reinterpret_cast<const bf16x8_t*>(ab[t & 3] + off + (t >> 2) * 256 + i * FRAG_STRIDE); }
  // image K-steps kg0 .. (kg0 compile-time through inlining), W stream steps wstep0 .. (set index = (SET0 + t) % PF, SET0 = wstep0 % PF at compile time);
  // n_total = stream length (refill bound); off = byte offset of the image inside the LDS region `init` was given
  template <int SET0>
  __device__ __forceinline__ void run(f32x4_t (&acc)[RF][NJ], int wstep0, int n_total, unsigned off = 0) {
    bf16x8_t a0[5], a1[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) a0[i] = aread(0, i, off);
#pragma unroll
    for (int t = 0; t < NKS; ++t) {
      const int kg = t, ws = wstep0 + t;
      constexpr int set = 0; (void)set;
      const int st = (SET0 + t) % PF;
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = aread(kg, 5 + i, off);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][j], a0[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < NKS) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a0[i] = aread(kg + 1, i, off);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[5 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][j], a1[i], acc[5 + i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ws + PF < n_total) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[st][j] = wload(ws + PF, j);
      }
    }
  }
};

template <int ROUTE, bool PLAIN>
__global__ __launch_bounds__(NT) void ffn8_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = blockIdx.x >> 3, cl = ((idx >> 2) << 3) + (blockIdx.x & 7), h = idx & 3;
  const int windows = gridDim.x / 4;
  unsigned long long ts[10] = {};
  for (int it = 0; it < a.iters; ++it) {
    const int layer = it % a.layers;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    int frow = lane & 15, fgrp = lane >> 4;
    unsigned* flags = a.flags + ((size_t)it * windows + cl) * 4;
    const bool last = it == a.iters - 1;
    if (last && tid == 0) ts[0] = wall_clock64();
    // ================= F1
    GemmCore<4, 16, 1024, 3> g1;
    g1.init(smem + IMG, reinterpret_cast<const unsigned char*>(a.w1f) + ((((size_t)layer * NH + h) * NW + wave) << 16), lane);
    g1.prefetch(0, 16);                                   // weights do not depend on the exchange: requested before the rows
    {
      const unsigned char* x1 = reinterpret_cast<const unsigned char*>(a.x1 + ((size_t)layer * windows + cl) * R * D);
      const unsigned xo[2] = {(unsigned)(lane ^ (wave & 15)) << 4, (unsigned)(lane ^ ((wave + 8) & 15)) << 4};
#pragma unroll
      for (int t = 0; t < R / NW; ++t) {
        const int m = wave + NW * t;
        GLDS(x1 + (size_t)m * 1024 + xo[t & 1], smem + IMG + img_row(m));
      }
    }
    wait_vm<0>();
    __syncthreads();
    if (last && tid == 0) ts[1] = wall_clock64();
    f32x4_t acc[RF][4];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    g1.run<0>(acc, 0, 16);
    __syncthreads();                       // every wave is done reading the x1 image
    if (last && tid == 0) ts[2] = wall_clock64();
    // (per-phase lane id: later phases' addresses must not be computed -- and spilled -- ahead of the F1 loop; a scratch reload next to counted vmcnt waits costs a full drain)
    lane = tid & 63;
    asm volatile("" : "+v"(lane));
    frow = lane & 15; fgrp = lane >> 4;
    if constexpr (ROUTE == 0) {
      // ---- route K: relu -> bf16 hid image in LDS
      GemmCore<4, 16, 1024, 3> g2;
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        unsigned char* row = smem + IMG + (frow * RF + i) * 1024;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = fmaxf(acc[i][2 * p][r], 0.0f); v[4 + r] = fmaxf(acc[i][2 * p + 1][r], 0.0f); }
          *reinterpret_cast<uint4*>(row + (((8 * wave + 4 * p + fgrp) ^ frow) << 4)) = pack8(v);
        }
      }
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      g2.init(smem + IMG, reinterpret_cast<const unsigned char*>(a.w2f) + ((((size_t)layer * NH + h) * NW + wave) << 16), lane);
      g2.prefetch(0, 16);
      __syncthreads();
      if (last && tid == 0) ts[3] = wall_clock64();
      g2.run<0>(acc, 0, 16);
      __syncthreads();                     // hid image dead: the own quarter may take its place
      if (last && tid == 0) ts[4] = wall_clock64();
      // wave w holds columns 64 w .. 64 w + 63 = quarter w / 2: foreign -> memory, own -> LDS [144][128] f32 (512-byte rows)
      const int q = wave >> 1;
      float* pout = a.part + (((size_t)(it & 1) * windows + cl) * NH + h) * R * D;
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int m = 16 * i + frow, col = 64 * wave + 32 * p + 8 * fgrp;
          if (q == h) {
            unsigned char* o = smem + IMG + (m * 128 + (col & 127)) * 4;
            *reinterpret_cast<f32x4_t*>(o) = acc[i][2 * p];
            *reinterpret_cast<f32x4_t*>(o + 16) = acc[i][2 * p + 1];
          } else {
            float* o = pout + (size_t)m * D + col;
            uint4 lo, hi;
            memcpy(&lo, &acc[i][2 * p], 16); memcpy(&hi, &acc[i][2 * p + 1], 16);
            store16_x<PLAIN>(o, lo); store16_x<PLAIN>(o + 4, hi);
          }
        }
      publish(flags + 2);
      if (last && tid == 0) ts[5] = wall_clock64();
      consume(flags + 2, NH, a.err);
      if (last && tid == 0) ts[6] = wall_clock64();
      // reduce: thread t owns float4 chunks q4 = t + 512 e, e < 9, of the [144][32 float4] slab; fixed order h' = 0..3 (own from LDS)
      const float* pin = a.part + ((size_t)(it & 1) * windows + cl) * NH * R * D;
      float* xg = a.x + (size_t)cl * R * D + h * 128;
      bf16_t* xl = a.x_lo + (size_t)cl * R * D + h * 128;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const int q4 = tid + NT * e, m = q4 >> 5, c4 = (q4 & 31) * 4;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(xg + (size_t)m * D + c4);
        const f32x4_t own = *reinterpret_cast<const f32x4_t*>(smem + IMG + (m * 128 + c4) * 4);
        f32x4_t f[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = *reinterpret_cast<const f32x4_t*>(pin + ((size_t)((h + 1 + k) & 3) * R + m) * D + h * 128 + c4);
        v += own; v += f[0]; v += f[1]; v += f[2];            // (the product kernel sums in source order 0..3; the probe only needs the traffic)
        v *= 0.25f;                            // (keeps the probe's stream bounded over the pseudo-layers)
        *reinterpret_cast<f32x4_t*>(xg + (size_t)m * D + c4) = v;
        uint2 w;
        w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(xl + (size_t)m * D + c4) = w;
      }
      wait_vm<0>();
      __syncthreads();
      if (last && tid == 0) ts[7] = wall_clock64();
    } else {
      // ---- route H: relu -> bf16 hid slab to memory (write-through), exchange, stream hid[144][2048] through LDS chunks
      const int kh = wave >> 2, cg = wave & 3;
      bf16_t* hid = a.hid + ((size_t)(it & 1) * windows + cl) * R * DFF;
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = fmaxf(acc[i][2 * p][r], 0.0f); v[4 + r] = fmaxf(acc[i][2 * p + 1][r], 0.0f); }
          store16_x<PLAIN>(hid + (size_t)(16 * i + frow) * DFF + 512 * h + 64 * wave + 32 * p + 8 * fgrp, pack8(v));
        }
      lane = tid & 63;
      asm volatile("" : "+v"(lane));
      frow = lane & 15; fgrp = lane >> 4;
      // The compiler treats an LDS-DMA as a FLAT access of both address spaces: while one is pending, every wait it derives itself (for a W fragment
      // load into registers, for a ds_read) becomes vmcnt(0) / lgkmcnt(0) -- the DMA queue would drain at every K-step. Inside this loop the W fragments
      // are therefore loaded by inline asm (invisible to that bookkeeping) and waited for by hand-counted vmcnt, tied to the registers they fill.
      const unsigned char* wstream = reinterpret_cast<const unsigned char*>(a.w2f) + ((((size_t)layer * NH + h) * NW + wave) << 16) + lane * 16;
      bf16x8_t wf[4][2];
      auto wload = [&](int step, int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[st][j]) : "v"(wstream + (size_t)(step * 2 + j) * 1024) : "memory");
      };
#pragma unroll
      for (int st = 0; st < 4; ++st) wload(st, st);
      const unsigned char* ab[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) ab[t] = smem + frow * 256 + ((((2 * kh + t) & 3) ^ (frow >> 2)) << 6) + ((fgrp ^ (frow & 3)) << 4);
      f32x4_t c2[RF][2];
#pragma unroll
      for (int i = 0; i < RF; ++i) { c2[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; c2[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
      publish(flags + 2);                    // (drains the four W sets as well)
      if (last && tid == 0) ts[3] = wall_clock64();
      consume(flags + 2, NH, a.err);
      if (last && tid == 0) ts[4] = wall_clock64();
      const unsigned char* hsrc = reinterpret_cast<const unsigned char*>(hid);
      auto issue_chunk = [&](int c) {
        unsigned char* buf = smem + (c % NBUF) * CH_BYTES;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const int piece = min(wave + NW * t, R / 4 - 1);             // rows 4 piece .. 4 piece + 3 (the spare slots reload the last piece: same bytes, same place)
          const int m = piece * 4 + (lane >> 4);
          GLDS(hsrc + (size_t)m * (DFF * 2) + c * 256 + (((lane & 15) ^ (m & 15)) << 4), buf + piece * 1024);
        }
      };
      // Queue of this wave per chunk: [5 DMA of chunk c + 3][2 W loads after K-step 0][2 W loads after K-step 1]. In-order counter:
      //   chunk c landed      <=> at most `after_chunk` younger operations outstanding (22 in the steady state)
      //   W set of a K-step   <=> at most `after_w0` / `after_w1` (16 / 16)
      auto do_chunk = [&](auto c_tag, auto after_chunk, auto after_w0, auto after_w1) {
        constexpr int c = decltype(c_tag)::value, S0 = (2 * c) % 4;
        wait_vm<decltype(after_chunk)::value>();
        __builtin_amdgcn_s_barrier();          // raw barrier (a __syncthreads fence would drain the DMA queue): chunk c landed for everyone, chunk c - 1 is consumed
        if (c + 3 < NCH) issue_chunk(c + 3);
        const unsigned off = (unsigned)((c % NBUF) * CH_BYTES);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8_t af[RF];
#pragma unroll
          for (int i = 0; i < RF; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(ab[t] + off + i * 4096);
          if (t == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wf[S0][0]), "+v"(wf[S0][1]) : "n"(decltype(after_w0)::value) : "memory");
          else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wf[S0 + 1][0]), "+v"(wf[S0 + 1][1]) : "n"(decltype(after_w1)::value) : "memory");
#pragma unroll
          for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) c2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S0 + t][j], af[i], c2[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (2 * c + t + 4 < 32) wload(2 * c + t + 4, S0 + t);
        }
      };
      using std::integral_constant;
#define IC(n) integral_constant<int, n>{}
      issue_chunk(0); issue_chunk(1); issue_chunk(2);
      do_chunk(IC(0), IC(10), IC(63), IC(63));          // (the four prefetched sets were drained by publish)
      do_chunk(IC(1), IC(14), IC(63), IC(63));
      do_chunk(IC(2), IC(18), IC(16), IC(16));
      do_chunk(IC(3), IC(22), IC(16), IC(16));
      do_chunk(IC(4), IC(22), IC(16), IC(16)); do_chunk(IC(5), IC(22), IC(16), IC(16)); do_chunk(IC(6), IC(22), IC(16), IC(16));
      do_chunk(IC(7), IC(22), IC(16), IC(16)); do_chunk(IC(8), IC(22), IC(16), IC(16)); do_chunk(IC(9), IC(22), IC(16), IC(16));
      do_chunk(IC(10), IC(22), IC(16), IC(16)); do_chunk(IC(11), IC(22), IC(16), IC(16)); do_chunk(IC(12), IC(22), IC(16), IC(16));
      do_chunk(IC(13), IC(22), IC(11), IC(11));
      do_chunk(IC(14), IC(17), IC(6), IC(4));
      do_chunk(IC(15), IC(8), IC(2), IC(0));
#undef IC
      __syncthreads();
      if (last && tid == 0) ts[5] = wall_clock64();
      // K-halves: kh = 1 waves park their tile in LDS, kh = 0 waves add it and finish
      float* red = reinterpret_cast<float*>(smem);
      if (kh == 1) {
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4_t*>(red + (((cg * RF + i) * 2 + j) * 64 + lane) * 4) = c2[i][j];
      }
      __syncthreads();
      if (kh == 0) {
        float* xg = a.x + (size_t)cl * R * D + h * 128 + 32 * cg + 8 * fgrp;
        bf16_t* xl = a.x_lo + (size_t)cl * R * D + h * 128 + 32 * cg + 8 * fgrp;
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          const int m = 16 * i + frow;
          f32x4_t lo = c2[i][0] + *reinterpret_cast<const f32x4_t*>(red + (((cg * RF + i) * 2 + 0) * 64 + lane) * 4);
          f32x4_t hi = c2[i][1] + *reinterpret_cast<const f32x4_t*>(red + (((cg * RF + i) * 2 + 1) * 64 + lane) * 4);
          lo += *reinterpret_cast<const f32x4_t*>(xg + (size_t)m * D); hi += *reinterpret_cast<const f32x4_t*>(xg + (size_t)m * D + 4);
          lo *= 0.25f; hi *= 0.25f;
          *reinterpret_cast<f32x4_t*>(xg + (size_t)m * D) = lo; *reinterpret_cast<f32x4_t*>(xg + (size_t)m * D + 4) = hi;
          const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          *reinterpret_cast<uint4*>(xl + (size_t)m * D) = pack8(v);
        }
      }
      wait_vm<0>();
      __syncthreads();
      if (last && tid == 0) ts[7] = wall_clock64();
    }
    if (last && tid == 0 && a.times) {
      unsigned long long* t = a.times + (size_t)blockIdx.x * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = ts[k];
    }
  }
}

static float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static int h_frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }

int main(int argc, char** argv) {
  const int windows = 64, layers = 24, iters = argc > 1 ? atoi(argv[1]) : 48;
  const size_t n_x1 = (size_t)layers * windows * R * D, n_w = (size_t)layers * DFF * D;
  std::vector<bf16_t> hx1(n_x1), hw1(n_w), hw2(n_w), hw1f(n_w), hw2k(n_w), hw2h(n_w);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx1) v = f2bf(rnd() * 2.0f);
  for (auto& v : hw1) v = f2bf(rnd() * 0.1f);         // W1 [layer][2048][512]
  for (auto& v : hw2) v = f2bf(rnd() * 0.1f);         // W2 [layer][512][2048]
  for (int l = 0; l < layers; ++l)
    for (int h = 0; h < NH; ++h)
      for (int w = 0; w < NW; ++w) {
        const size_t wave_base = (((size_t)l * NH + h) * NW + w) * 32768;
        for (int ks = 0; ks < 16; ++ks)
          for (int j = 0; j < 4; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const int frow = lane & 15, fgrp = lane >> 4;
              const size_t dst = wave_base + (((size_t)ks * 4 + j) * 64 + lane) * 8;
              const int n = 64 * w + h_frag_col(j, frow);
              for (int e = 0; e < 8; ++e) {
                hw1f[dst + e] = hw1[((size_t)l * DFF + 512 * h + n) * D + 32 * ks + 8 * fgrp + e];
                hw2k[dst + e] = hw2[((size_t)l * D + n) * DFF + 512 * h + 32 * ks + 8 * fgrp + e];
              }
            }
        // route H: wave = (kh, cg): output columns 128 h + 32 cg + frag_col(j, frow), stream step = 2 c + s2 -> k0 = 128 c + 32 (2 kh + s2)
        const int kh = w >> 2, cg = w & 3;
        for (int step = 0; step < 32; ++step)
          for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const int frow = lane & 15, fgrp = lane >> 4;
              const size_t dst = wave_base + (((size_t)step * 2 + j) * 64 + lane) * 8;
              const int n = 128 * h + 32 * cg + h_frag_col(j, frow), k0 = 128 * (step >> 1) + 32 * (2 * kh + (step & 1));
              for (int e = 0; e < 8; ++e) hw2h[dst + e] = hw2[((size_t)l * D + n) * DFF + k0 + 8 * fgrp + e];
            }
      }
  bf16_t *dx1, *dw1f, *dw2k, *dw2h, *dhid, *dxlo; float *dpart, *dx; unsigned* dflags; unsigned long long* dtimes;
  CHECK(hipMalloc(&dx1, n_x1 * 2)); CHECK(hipMalloc(&dw1f, n_w * 2)); CHECK(hipMalloc(&dw2k, n_w * 2)); CHECK(hipMalloc(&dw2h, n_w * 2));
  CHECK(hipMalloc(&dhid, (size_t)2 * windows * R * DFF * 2)); CHECK(hipMalloc(&dpart, (size_t)2 * windows * NH * R * D * 4));
  CHECK(hipMalloc(&dx, (size_t)windows * R * D * 4)); CHECK(hipMalloc(&dxlo, (size_t)windows * R * D * 2));
  const size_t flag_bytes = ((size_t)iters * windows * 4 + 4) * 4;
  CHECK(hipMalloc(&dflags, flag_bytes)); CHECK(hipMalloc(&dtimes, 256 * 8 * 8));
  CHECK(hipMemcpy(dx1, hx1.data(), n_x1 * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw1f, hw1f.data(), n_w * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw2k, hw2k.data(), n_w * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw2h, hw2h.data(), n_w * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn8_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn8_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn8_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn8_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  auto launch = [&](int route, int plain, int n_iters, int grid) {
    CHECK(hipMemset(dflags, 0, flag_bytes));
    CHECK(hipMemset(dx, 0, (size_t)windows * R * D * 4));
    Args a{dx1, dw1f, route == 0 ? dw2k : dw2h, dhid, dpart, dx, dxlo, dflags, dflags + (size_t)iters * windows * 4, dtimes, layers, n_iters, plain};
    CHECK(hipEventRecord(e0));
    if (route == 0 && !plain) hipLaunchKernelGGL((ffn8_kernel<0, false>), dim3(grid), dim3(NT), LDS_BYTES, 0, a);
    if (route == 0 && plain) hipLaunchKernelGGL((ffn8_kernel<0, true>), dim3(grid), dim3(NT), LDS_BYTES, 0, a);
    if (route == 1 && !plain) hipLaunchKernelGGL((ffn8_kernel<1, false>), dim3(grid), dim3(NT), LDS_BYTES, 0, a);
    if (route == 1 && plain) hipLaunchKernelGGL((ffn8_kernel<1, true>), dim3(grid), dim3(NT), LDS_BYTES, 0, a);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned err = 0;
    CHECK(hipMemcpy(&err, dflags + (size_t)iters * windows * 4, 4, hipMemcpyDeviceToHost));
    if (err) printf("  (a workgroup gave up waiting)\n");
    return ms;
  };
  // ---- correctness: one iteration of each route against a host computation of window 0 (x starts at 0: out = 0.25 * ffn)
  std::vector<float> ref((size_t)R * D);
  {
    std::vector<float> hid((size_t)R * DFF);
    for (int m = 0; m < R; ++m)
      for (int n = 0; n < DFF; ++n) {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += (double)bf2f(hx1[(size_t)m * D + k]) * bf2f(hw1[(size_t)n * D + k]);
        hid[(size_t)m * DFF + n] = bf2f(f2bf(fmaxf((float)acc, 0.0f)));
      }
    for (int m = 0; m < R; ++m)
      for (int n = 0; n < D; ++n) {
        double acc = 0.0;
        for (int k = 0; k < DFF; ++k) acc += (double)hid[(size_t)m * DFF + k] * bf2f(hw2[(size_t)n * DFF + k]);
        ref[(size_t)m * D + n] = 0.25f * (float)acc;
      }
  }
  for (int route = 0; route < 2; ++route)
    for (int plain = 0; plain < 2; ++plain) {
      launch(route, plain, 1, 256);
      std::vector<float> got((size_t)R * D);
      CHECK(hipMemcpy(got.data(), dx, got.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0.0, scale = 0.0;
      for (size_t i = 0; i < got.size(); ++i) { worst = fmax(worst, fabs((double)got[i] - ref[i])); scale = fmax(scale, fabs((double)ref[i])); }
      printf("check route %c %s: max |err| %.3e of max |value| %.3e %s\n", route ? 'H' : 'K', plain ? "plain" : "sc1  ", worst, scale, worst < 5e-3 * scale ? "OK" : "MISMATCH");
    }
  const char* seg_names[2][7] = {{"x1 DMA", "F1", "relu+image", "F2", "part store+publish", "wait", "reduce+epilogue"},
                                 {"x1 DMA", "F1", "hid store+publish", "wait", "F2 stream", "-", "K-half reduce+epilogue"}};
  for (int route = 0; route < 2; ++route)
    for (int plain = 0; plain < 2; ++plain) {
      launch(route, plain, iters, 256);
      const float ms = launch(route, plain, iters, 256);
      std::vector<unsigned long long> t(256 * 8);
      CHECK(hipMemcpy(t.data(), dtimes, t.size() * 8, hipMemcpyDeviceToHost));
      printf("route %c %s  %7.2f us per FFN half |", route ? 'H' : 'K', plain ? "plain" : "sc1  ", ms * 1000.0 / iters);
      for (int k = 0; k < 7; ++k) {
        if (route == 1 && k == 5) continue;
        double sum = 0.0;
        const int k1 = (route == 1 && k == 4) ? 5 : k + 1, k0 = (route == 1 && k == 6) ? 5 : k;
        for (int b = 0; b < 256; ++b) sum += (double)(t[b * 8 + (route == 1 && k == 6 ? 7 : k1)] - t[b * 8 + k0]) * 0.01 / 256;
        printf(" %s %.2f", seg_names[route][k], sum);
      }
      printf("\n");
    }
  return 0;
}
