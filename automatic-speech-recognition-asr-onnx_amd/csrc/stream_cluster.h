// Shared device helpers of the streaming-Paraformer cluster kernels (stream_layers.hip: encoder layers, stream_dec.hip: decoder layers): clusters of four
// workgroups per stream that meet through memory, 8 waves per workgroup, weights streamed from fragment-major copies straight into registers.
#pragma once
#include "kernels.h"

namespace {

constexpr int NH = 4, NW = 8, NT = NW * 64;      // workgroups per cluster (= heads), waves / threads per workgroup

typedef unsigned long long u64;
// Every barrier of this kernel orders LDS traffic only (what crosses workgroups goes through publish / consume, which drain the vector queue themselves):
// a bare s_barrier behind an LDS wait. __syncthreads() would also wait for every outstanding global load -- the weight batches and the L2 warm-up
// requested ahead are meant to stay in flight across barriers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;      // (a plain vector: HIP's uint4 class cannot be read through an address-space pointer)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// pointers read from the layer table arrive as generic ones: say that they are global (a flat load counts on both wait counters and orders against LDS)
#define GAS __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const GAS T* glob(const T* p) { return (const GAS T*)p; }
template <typename T> __device__ __forceinline__ GAS T* glob(T* p) { return (GAS T*)p; }

// Payload and counters of an exchange: agent-scope relaxed accesses (sc1: stores write through this XCD's L2, loads do not trust a stale line of it,
// the counter lives at the memory side). Tried and dropped: "workgroup"-scope accesses (sc0 stores and loads around an L1 invalidate, L2 atomics) for clusters
// that sit on one XCD, to shorten every hop to an L2 round trip -- on gfx950 a waiting workgroup then sees its siblings' counts 3-5 us LATER than through the
// memory side (7 us per exchange instead of 2.5), and under graph replay clusters timed out.
__device__ __forceinline__ void put8(void* p, u64 v) { __hip_atomic_store(reinterpret_cast<u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 get8(const void* p) { return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void publish(unsigned* flag, bool withhold = false) {      // withhold: fault injection (tests), the count never arrives
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave's stores are out
  lds_barrier();
  if (threadIdx.x == 0 && !withhold) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void consume(unsigned* flag, unsigned* err, unsigned need = (unsigned)NH) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    const u64 t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;                  // somebody gave up: the launch is void
        if (wall_clock64() - t0 > 20000000ull) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 0.2 s of a 100 MHz clock
      }
    }
  }
  lds_barrier();
}

// L2 warm-up: the 6.3 MB of a layer are read by all the (<= 32) workgroups of an XCD at about the same time, so every weight request is an L2 MISS that the
// first asker waits out (2.5 us) with the others queued behind it. Each workgroup therefore touches 1 / n of the NEXT phase's region -- one lane per 128-byte
// line, a wave instruction covers 8 KB -- while it waits for its cluster; the fills run under the wait and the phase's own requests hit the L2.
// The value is consumed only behind the next publish (whose vmcnt(0) has drained it anyway).
__device__ __forceinline__ unsigned warm(const unsigned char* base, int n_lines, int wg, int n_wg, int tid) {
  const int per = min((n_lines + n_wg - 1) / n_wg, 2 * (NT - 64));      // (few streams: two touches per thread at most -- the head of the region -- rather than a long detour)
  unsigned t = 0;
  for (int i = tid - 64; i < per; i += NT - 64) {           // (wave 0 polls the counter: nothing of its own queues in front of the poll)
    if (i < 0) break;
    const int line = wg * per + i;
    if (line < n_lines) t = *glob(reinterpret_cast<const unsigned*>(base + (size_t)line * 128));
  }
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// fragments [first, first + N) of this wave's stream -> registers
template <int N>
__device__ __forceinline__ void wload(u32x4 (&w)[N], const unsigned char* wp, int first) {
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = *glob(reinterpret_cast<const u32x4*>(wp + (size_t)(first + i) * 1024));
  __builtin_amdgcn_sched_barrier(0);           // the requests stay HERE: the scheduler would sink every load to its first use
}
// KB k-steps of 32 against NJ column tiles: one A fragment read from LDS per k-step; NACC accumulators per tile break the MFMA dependency chain
template <int NJ, int KB, int NACC>
__device__ __forceinline__ void wmul(const u32x4 (&w)[KB * NJ], const unsigned char* ap, int ks0, f32x4_t (&acc)[NJ * NACC]) {
#pragma unroll
  for (int kk = 0; kk < KB; ++kk) {
    const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ap + (ks0 + kk) * 64);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      acc[j * NACC + (kk % NACC)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, w[kk * NJ + j]), acc[j * NACC + (kk % NACC)], 0, 0, 0);
  }
}
// one GEMM phase of a wave: KS k-steps in batches of KB; batch 0 is already requested into w0 (and batch 1 into w1 when PF2), batch b + 2 is requested
// as soon as batch b is multiplied
template <int NJ, int KS, int KB, int NACC, bool PF2>
__device__ __forceinline__ void gemm_phase(u32x4 (&w0)[KB * NJ], u32x4 (&w1)[KB * NJ], const unsigned char* wp, const unsigned char* ap, f32x4_t (&acc)[NJ * NACC]) {
  constexpr int NB = KS / KB;
  static_assert(NB == 1 || NB % 2 == 0, "batches come in pairs");
  if constexpr (NB == 1) {
    wmul<NJ, KB, NACC>(w0, ap, 0, acc);
  } else {
    if constexpr (!PF2) wload<KB * NJ>(w1, wp, KB * NJ);
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      wmul<NJ, KB, NACC>(w0, ap, b * KB, acc);
      if (b + 2 < NB) wload<KB * NJ>(w0, wp, (b + 2) * KB * NJ);
      wmul<NJ, KB, NACC>(w1, ap, (b + 1) * KB, acc);
      if (b + 3 < NB) wload<KB * NJ>(w1, wp, (b + 3) * KB * NJ);
    }
  }
}

}  // namespace
