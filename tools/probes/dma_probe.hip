// L2 -> LDS stream-rate probe (standalone; not part of libasr_mi355x.so).
// One 512-thread workgroup per CU streams GEMM-shaped operand stages (rows x 128 B or rows x 256 B pieces of a
// row-major bf16 matrix with a 1 KB / 4 KB row pitch) into LDS with global_load_lds_dwordx4 and does nothing else.
// Question answered: what the LDS-DMA path sustains per CU for each lane -> address pattern and wait discipline.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args {
  const unsigned char* buf;
  int pitch;          // bytes per matrix row
  int rows;           // rows per stage (multiple of 8 x waves)
  int ksteps;         // K-steps per pass over the rows (k0 advances by the piece width)
  int passes;
  int pattern;        // 0 linear 1 KB, 1 8 rows x 128 B, 2 = 1 with slot ^= row, 3 4 rows x 256 B xor (row & 15), 4 = 1 with slot ^= (row & 3) << 1, 5 4 rows x 256 B linear
  int drain;          // 1: vmcnt(0) + barrier per stage; 0: never wait inside the loop; 2: counted (one stage in flight)
  int region_rows;    // rows of the region a workgroup walks (tile rows); regions of the workgroups of one XCD may coincide
  int share;          // workgroups of an XCD use region (wg_in_xcd % share)
  int to_vgpr;        // 1: plain global_load_dwordx4 into registers instead of the LDS-DMA; 2: MIXED -- odd pieces into registers, even pieces by LDS-DMA (do the two paths add up?)
};

__global__ __launch_bounds__(512) void probe(const Args a, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3;
  const int region = xcd * a.share + (in_xcd % a.share);
  const unsigned char* base = a.buf + (size_t)region * a.region_rows * a.pitch;
  int row_in_piece, off_in_row, piece_rows, kwidth;
  switch (a.pattern) {
    case 0: row_in_piece = 0; off_in_row = lane * 16; piece_rows = 1; kwidth = 1024; break;
    case 1: row_in_piece = lane >> 3; off_in_row = (lane & 7) * 16; piece_rows = 8; kwidth = 128; break;
    case 2: row_in_piece = lane >> 3; off_in_row = ((lane & 7) ^ (lane >> 3)) * 16; piece_rows = 8; kwidth = 128; break;
    case 3: row_in_piece = lane >> 4; off_in_row = ((lane & 15) ^ ((lane >> 4) * 5 & 15)) * 16; piece_rows = 4; kwidth = 256; break;
    case 4: row_in_piece = lane >> 3; off_in_row = ((lane & 7) ^ (((lane >> 3) & 3) << 1)) * 16; piece_rows = 8; kwidth = 128; break;
    default: row_in_piece = lane >> 4; off_in_row = (lane & 15) * 16; piece_rows = 4; kwidth = 256; break;
  }
  const int pieces = a.rows / piece_rows;          // wave-instructions per stage (8 per wave are issued)
  constexpr int per_wave = 8;
  unsigned acc = 0;
  uint4 r[per_wave];
  for (int p = 0; p < a.passes; ++p) {
    for (int ks = 0; ks < a.ksteps; ++ks) {
      const int slot = (p * a.ksteps + ks) & 1;
      unsigned char* dst = smem + slot * pieces * 1024;
      if (a.drain == 2) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8) : "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
      for (int t = 0; t < per_wave; ++t) {
        const int piece = wave + 8 * t;
        const int row = (piece * piece_rows + row_in_piece) % a.region_rows;
        const unsigned char* src = base + (size_t)row * a.pitch + (size_t)ks * kwidth + off_in_row;
        if (a.pattern == 0) src = base + ((size_t)(ks * pieces + piece) * 1024) % ((size_t)a.region_rows * a.pitch) + lane * 16;
        if (a.to_vgpr == 1 || (a.to_vgpr == 2 && (t & 1))) {
          r[t] = *reinterpret_cast<const uint4*>(src);
        } else {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
        }
      }
      if (a.to_vgpr == 1) { for (int t = 0; t < per_wave; ++t) acc += r[t].x ^ r[t].w; }
      if (a.to_vgpr == 2) { for (int t = 1; t < per_wave; t += 2) acc += r[t].x ^ r[t].w; }
      if (a.drain == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = acc + smem[17];
}

int main() {
  const size_t bytes = (size_t)256 << 20;
  unsigned char* buf; unsigned* sink;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMalloc(&sink, 4096));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Case { const char* name; int pattern, drain, rows, pitch, share, to_vgpr, region_rows; };
  std::vector<Case> cases;
  const char* pn[6] = {"linear-1KB", "8x128 linear", "8x128 xor-row", "4x256 xor", "8x128 xor-pair", "4x256 linear"};
  for (int vg = 0; vg < 2; ++vg)
    for (int pat = 0; pat < 6; ++pat)
      for (int drain = 0; drain < 3; ++drain) {
        if (vg && drain == 2) continue;
        cases.push_back({pn[pat], pat, drain, 544, 1024, 4, vg, 544});
      }
  // footprint sensitivity: every workgroup its own region (32 regions per XCD = 17 MB per XCD > L2) and a 4 KB pitch (FFN-2 operand)
  cases.push_back({"8x128 xor-row own-region", 2, 1, 544, 1024, 32, 0, 544});
  cases.push_back({"8x128 linear own-region", 1, 1, 544, 1024, 32, 0, 544});
  cases.push_back({"8x128 xor-row pitch4K", 2, 1, 544, 4096, 1, 0, 544});
  cases.push_back({"8x128 linear pitch4K", 1, 1, 544, 4096, 1, 0, 544});
  cases.push_back({"4x256 xor pitch4K", 3, 1, 272, 4096, 1, 0, 272});
  for (int drain = 0; drain < 2; ++drain) {
    cases.push_back({"8x128 xor-row MIXED", 2, drain, 544, 1024, 4, 2, 544});
    cases.push_back({"8x128 linear MIXED", 1, drain, 544, 1024, 4, 2, 544});
  }
  printf("%-28s %5s %5s %6s %9s %10s %9s\n", "pattern", "vgpr", "drain", "rows", "GB/s/CU", "TB/s chip", "B/clk@2.4");
  for (const Case& c : cases) {
    Args a;
    a.buf = buf; a.pitch = c.pitch; a.rows = c.rows; a.pattern = c.pattern; a.drain = c.drain; a.share = c.share; a.to_vgpr = c.to_vgpr;
    a.region_rows = c.region_rows;
    const int kw = (c.pattern == 0) ? 1024 : (c.pattern == 3 || c.pattern == 5) ? 256 : 128;
    a.ksteps = c.pitch / kw; a.passes = 40;
    if (c.pattern == 0) a.ksteps = 8;
    const int piece_rows = c.pattern == 0 ? 1 : (kw == 256 ? 4 : 8);
    if (c.pattern == 0) a.rows = c.rows / 8;            // same bytes per stage as the 128-B patterns
    const size_t lds = (size_t)2 * (a.rows / piece_rows) * 1024;
    if (lds > 160 * 1024) { printf("%s: skip (LDS)\n", c.name); continue; }
    for (int it = 0; it < 2; ++it) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), lds, 0, a, sink);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wg = (double)a.passes * a.ksteps * 64 * 1024.0;
    const double gbs = per_wg / (ms * 1e-3) / 1e9;
    printf("%-28s %5d %5d %6d %9.1f %10.2f %9.1f\n", c.name, c.to_vgpr, c.drain, a.rows, gbs, gbs * 256 / 1e3, gbs / 2.4);
  }
  return 0;
}
