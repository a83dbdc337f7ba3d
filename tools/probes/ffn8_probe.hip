// FFN-pair probe for the 8-wave SANM block design (standalone; not part of libasr_mi355x.so).
// One 512-thread workgroup per CU = (window, hidden slab h). The A operand (144 x 512 bf16) is DMA'd WHOLE into LDS, the weights are read
// straight from global memory into MFMA fragments from a fragment-major copy (one contiguous KB per wave instruction), there is no barrier
// inside a GEMM loop:
//   F1  hid[144][512 h ..] = relu(x1[144][512] W1[512 h ..][512]^T)   -> bf16 image in LDS (over the dead x1 image)
//   F2  part[144][512]     = hid[144][512 h ..] W2[:, 512 h ..]^T     -> f32 partial (K-split FFN-2), stored
// Question answered: what one such pair costs per workgroup with all 256 CUs busy, against 62 us for phases C + D of the 12-wave kernel.
//   hipcc --offload-arch=gfx950 -O3 -o ffn8_probe ffn8_probe.hip && ./ffn8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef uint16_t bf16_t;

constexpr int R = 144, RF = 9, D = 512, DFF = 2048, NH = 4;
constexpr int NW = 8, NT = NW * 64;
constexpr int LDS_BYTES = 160 * 1024;
constexpr int PF = 3;                       // W fragment sets in flight (K-steps ahead)

struct Args {
  const bf16_t* x1;        // [layers][windows][144][512]
  const bf16_t* w1f;       // [layers][4 h][8 w][16 ks][4 j][64 lane][8]
  const bf16_t* w2f;       // same shape
  float* part;             // [windows][4 h][144][512]
  unsigned long long* times;
  int layers, iters, mode;     // mode bit 1: no MFMA, 2: no A fragment reads, 4: no W loads (after the first), 8: skip the partial stores
};

__device__ __forceinline__ int frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// C[144][64 (this wave)] += A[144][512] (LDS image, 1 KB rows, 16-byte chunks at c ^ (row & 15)) x Wf (this wave's fragment stream, 16 K-steps x 4 KB)
// Addresses: chunk c = 4 ks + fgrp, position c ^ frow = ((ks >> 2) << 2 | (ks & 3) ^ (frow >> 2)) << 2 | fgrp ^ (frow & 3): four lane bases (one per ks & 3)
// plus immediates; the W stream is a wave-uniform base + lane * 16.
template <int MODE>
__device__ __forceinline__ void gemm_144x64(const unsigned char* img, const unsigned char* wf_wave, f32x4_t (&acc)[RF][4], int lane) {
  const int frow = lane & 15, fgrp = lane >> 4;
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wload = [&](int ks, int j) { return *reinterpret_cast<const bf16x8_t*>(wf_wave + (size_t)(ks * 4 + j) * 1024 + lane16); };
  bf16x8_t wf[PF][4];
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[s][j] = wload(s, j);
  const unsigned char* abase[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) abase[q] = img + frow * 1024 + ((q ^ (frow >> 2)) << 6) + ((fgrp ^ (frow & 3)) << 4);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    bf16x8_t af[RF];
    if (!(MODE & 2) || ks == 0) {
#pragma unroll
      for (int i = 0; i < RF; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(abase[ks & 3] + (ks >> 2) * 256 + i * 16384);
    }
    if (!(MODE & 1)) {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % PF][j], af[i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wf[ks % PF][j]));
#pragma unroll
      for (int i = 0; i < RF; ++i) asm volatile("" ::"v"(af[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks + PF < 16 && !(MODE & 4)) {          // refill the set just consumed: in flight over the next PF - 1 K-steps
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[ks % PF][j] = wload(ks + PF, j);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(NT) void ffn8_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = blockIdx.x >> 3, cl = ((idx >> 2) << 3) + (blockIdx.x & 7), h = idx & 3;      // the four workgroups of a window on one XCD
  const int windows = gridDim.x / 4;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  for (int it = 0; it < a.iters; ++it) {
    const int layer = it % a.layers;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));           // per-iteration address arithmetic (the real kernel runs a phase once)
    const int frow = lane & 15, fgrp = lane >> 4;
    if (tid == 0) t0 = wall_clock64();
    // ---- x1 rows -> LDS image (one row = one wave instruction)
    const unsigned char* x1 = reinterpret_cast<const unsigned char*>(a.x1 + ((size_t)layer * windows + cl) * R * D);
    const unsigned xo[2] = {(unsigned)(lane ^ (wave & 15)) << 4, (unsigned)(lane ^ ((wave + 8) & 15)) << 4};
#pragma unroll
    for (int t = 0; t < R / NW; ++t) {
      const int m = wave + NW * t;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x1 + (size_t)m * 1024 + xo[t & 1]),
                                       (__attribute__((address_space(3))) void*)(smem + m * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) t1 = wall_clock64();
    f32x4_t acc[RF][4];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const size_t wslab = ((size_t)layer * NH + h) * NW + wave;
    gemm_144x64<MODE>(smem, reinterpret_cast<const unsigned char*>(a.w1f + wslab * (16 * 4 * 512)), acc, lane);
    __syncthreads();                       // every wave is done reading the x1 image
    if (tid == 0) t2 = wall_clock64();
    // ---- relu -> bf16 hid image (this wave's 64 columns = chunks 8 w .. 8 w + 7)
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      unsigned char* row = smem + (16 * i + frow) * 1024;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = fmaxf(acc[i][2 * p][r], 0.0f); v[4 + r] = fmaxf(acc[i][2 * p + 1][r], 0.0f); }
        uint4 pk;
        pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(row + (((8 * wave + 4 * p + fgrp) ^ frow) << 4)) = pk;
      }
    }
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (tid == 0) t3 = wall_clock64();
    gemm_144x64<MODE>(smem, reinterpret_cast<const unsigned char*>(a.w2f + wslab * (16 * 4 * 512)), acc, lane);
    if (tid == 0) t4 = wall_clock64();
    if (!(MODE & 8)) {
      float* out = a.part + ((size_t)cl * NH + h) * R * D;
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float* o = out + (size_t)(16 * i + frow) * D + 64 * wave + 32 * p + 8 * fgrp;
          *reinterpret_cast<f32x4_t*>(o) = acc[i][2 * p];
          *reinterpret_cast<f32x4_t*>(o + 4) = acc[i][2 * p + 1];
        }
    } else {
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    }
    __syncthreads();                       // hid image dead before the next iteration's DMA
    if (tid == 0 && a.times && it == a.iters - 1) {
      unsigned long long* t = a.times + (size_t)blockIdx.x * 8;
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3; t[4] = t4; t[5] = wall_clock64();
    }
  }
}

static float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static int h_frag_col(int j, int fr) { return (j >> 1) * 32 + ((fr >> 2) << 3) + ((j & 1) << 2) + (fr & 3); }

int main(int argc, char** argv) {
  const int windows = 64, layers = 24, iters = argc > 1 ? atoi(argv[1]) : 48;
  const size_t n_x1 = (size_t)layers * windows * R * D, n_w = (size_t)layers * DFF * D;
  std::vector<bf16_t> hx1(n_x1), hw1(n_w), hw2(n_w), hw1f(n_w), hw2f(n_w);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx1) v = f2bf(rnd() * 2.0f);
  for (auto& v : hw1) v = f2bf(rnd() * 0.1f);         // W1 [layer][2048][512]
  for (auto& v : hw2) v = f2bf(rnd() * 0.1f);         // W2 [layer][512][2048]
  for (int l = 0; l < layers; ++l)
    for (int h = 0; h < NH; ++h)
      for (int w = 0; w < NW; ++w)
        for (int ks = 0; ks < 16; ++ks)
          for (int j = 0; j < 4; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const int frow = lane & 15, fgrp = lane >> 4;
              const size_t dst = ((((((size_t)l * NH + h) * NW + w) * 16 + ks) * 4 + j) * 64 + lane) * 8;
              const int n = 64 * w + h_frag_col(j, frow);
              for (int e = 0; e < 8; ++e) {
                hw1f[dst + e] = hw1[((size_t)l * DFF + 512 * h + n) * D + 32 * ks + 8 * fgrp + e];
                hw2f[dst + e] = hw2[((size_t)l * D + n) * DFF + 512 * h + 32 * ks + 8 * fgrp + e];
              }
            }
  bf16_t *dx1, *dw1f, *dw2f; float* dpart; unsigned long long* dtimes;
  CHECK(hipMalloc(&dx1, n_x1 * 2)); CHECK(hipMalloc(&dw1f, n_w * 2)); CHECK(hipMalloc(&dw2f, n_w * 2));
  CHECK(hipMalloc(&dpart, (size_t)windows * NH * R * D * 4)); CHECK(hipMalloc(&dtimes, 256 * 8 * 8));
  CHECK(hipMemcpy(dx1, hx1.data(), n_x1 * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw1f, hw1f.data(), n_w * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw2f, hw2f.data(), n_w * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  Args a{dx1, dw1f, dw2f, dpart, dtimes, layers, 1, 0};
  auto launch = [&](int mode, const Args& aa, int grid) {
    switch (mode) {
#define CASE(M) case M: CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn8_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); \
      hipLaunchKernelGGL(ffn8_kernel<M>, dim3(grid), dim3(NT), LDS_BYTES, 0, aa); break;
      CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(8) CASE(12) CASE(6)
#undef CASE
      default: printf("mode %d not instantiated\n", mode); exit(1);
    }
    CHECK(hipGetLastError());
  };
  // ---- correctness: one iteration (layer 0), workgroups of window 0 against a host computation
  launch(0, a, 256);
  CHECK(hipDeviceSynchronize());
  {
    std::vector<float> part((size_t)NH * R * D);
    CHECK(hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (int h = 0; h < NH; ++h) {
      std::vector<float> hid((size_t)R * 512);
      for (int m = 0; m < R; ++m)
        for (int n = 0; n < 512; ++n) {
          double acc = 0.0;
          for (int k = 0; k < D; ++k) acc += (double)bf2f(hx1[(size_t)m * D + k]) * bf2f(hw1[((size_t)512 * h + n) * D + k]);
          hid[(size_t)m * 512 + n] = bf2f(f2bf(fmaxf((float)acc, 0.0f)));
        }
      for (int m = 0; m < R; m += 7)
        for (int n = 0; n < D; n += 5) {
          double acc = 0.0;
          for (int k = 0; k < 512; ++k) acc += (double)hid[(size_t)m * 512 + k] * bf2f(hw2[(size_t)n * DFF + 512 * h + k]);
          const double got = part[((size_t)h * R + m) * D + n];
          worst = fmax(worst, fabs(got - acc)); scale = fmax(scale, fabs(acc));
        }
    }
    printf("check window 0: max |err| %.3e of max |value| %.3e %s\n", worst, scale, worst < 2e-2 * scale ? "OK" : "MISMATCH");
  }
  const int modes[] = {0, 8, 1, 2, 3, 4, 5, 6, 12};
  const char* names[] = {"full", "no partial stores", "no MFMA", "no A reads", "no MFMA, no A reads", "no W loads", "no MFMA, no W loads", "no A reads, no W loads", "no W loads, no stores"};
  for (int grid : {256, 64}) {
    for (int mi = 0; mi < 9; ++mi) {
      Args b = a; b.iters = iters; b.mode = modes[mi];
      launch(modes[mi], b, grid);                   // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      launch(modes[mi], b, grid);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> t(256 * 8);
      CHECK(hipMemcpy(t.data(), dtimes, t.size() * 8, hipMemcpyDeviceToHost));
      double seg[5] = {0, 0, 0, 0, 0};
      for (int b2 = 0; b2 < grid; ++b2)
        for (int k = 0; k < 5; ++k) seg[k] += (double)(t[b2 * 8 + k + 1] - t[b2 * 8 + k]) * 0.01 / grid;
      printf("grid %3d  %-26s %7.2f us per pair | last iteration: x1 DMA %5.2f  F1 %5.2f  relu+image %5.2f  F2 %5.2f  stores %5.2f\n", grid, names[mi], ms * 1000.0 / iters,
             seg[0], seg[1], seg[2], seg[3], seg[4]);
    }
  }
  return 0;
}
