// MFMA GEMMs for gfx950 with fused epilogues (bias / activation / residual adds / transposed
// stores / row arg-max). C[M][N] = A[M][K] * W[N][K]^T.
//
// bf16 kernel: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16; operands staged by global_load_lds_dwordx4 (LDS-DMA, no VGPR round
// trip) into a double-buffered LDS image. The LDS image is lane-linear, so the 16-byte-slot XOR
// swizzle (slot ^= row & 7) is applied on the per-lane GLOBAL source address and again on the
// ds_read_b128 address (cdna_hip_programming.md rule 21 / T2). Workgroup ids are remapped so each
// XCD (private L2) walks a contiguous range of output tiles sharing A rows (T1).
//
// f32 kernel: same tiling on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) for verification mode.
#include "gemm.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int BK16 = 64;                       // bf16 K-step
constexpr int LDS_TILE_BYTES = BM * BK16 * 2;  // 16 KiB per operand tile

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_GELU_TANH: {
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (1.0f + tanhf(u));
    }
    default: return v;
  }
}

// XCD-aware bijective remap (8 XCDs, block b runs on XCD b % 8): XCD x gets a contiguous tile range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

template <typename OutT>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& g, const f32x4_t& acc, int m0, int n) {
  // this lane holds C[m0 + r][n], r = 0..3
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (g.bias) {
    const float b = g.bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += b;
  }
  if (g.add_t) {
    const float4 t = *reinterpret_cast<const float4*>(g.add_t + (size_t)n * g.ld_add_t + m0);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  if (g.add) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += g.add[(size_t)(m0 + r) * g.ld_add + n];
  }
  if (g.act != ACT_NONE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
  }
  if (g.out_f32) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m0 + r < g.M) g.out_f32[(size_t)(m0 + r) * g.ld_out_f32 + n] = v[r];
  }
  if (n < g.n_split) {
    if (g.out_lo) {
      OutT* o = reinterpret_cast<OutT*>(g.out_lo);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m0 + r < g.M) Elem<OutT>::store(o + (size_t)(m0 + r) * g.ld_out_lo + n, v[r]);
    }
  } else if (g.out_t) {
    OutT* o = reinterpret_cast<OutT*>(g.out_t) + (size_t)(n - g.n_split) * g.ld_out_t + m0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m0 + r < g.M) Elem<OutT>::store(o + r, v[r]);
  }
}

// per-row arg-max over this wave's 64 columns; acc[i][j] fragments, lane holds col (lane&15)+16j
__device__ __forceinline__ void epilogue_argmax(const GemmArgs& g, f32x4_t (&acc)[4][4], int m_wave, int n_wave, int lane) {
  const int slab = n_wave >> 6;
  const int n_slabs = g.N >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float best = -INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n_wave + j * 16 + (lane & 15);
        float v = acc[i][j][r] + (g.bias ? g.bias[n] : 0.0f);
        if (n >= g.n_valid) v = -INFINITY;
        if (v > best) { best = v; bidx = n; }   // j ascending => lowest index kept on ties
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
      }
      const int m = m_wave + i * 16 + (lane >> 4) * 4 + r;
      if ((lane & 15) == 0 && m < g.M) {
        g.amax_val[(size_t)m * n_slabs + slab] = best;
        g.amax_idx[(size_t)m * n_slabs + slab] = bidx;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ bf16
__global__ __launch_bounds__(256, 2) void gemm_bf16_128x128x64(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A 16K | W 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;

  // staging: each wave-instruction moves 8 rows x 128 B (8 slots of 16 B) = 1 KiB, lane-linear in LDS
  const int srow = lane >> 3;                         // 0..7 within the 8-row group
  const int sslot = (lane & 7) ^ srow;                // source slot: un-swizzle on the global side
  const bf16_t* a_src = reinterpret_cast<const bf16_t*>(g.A) + (size_t)(tile_m * BM + wave * 8 + srow) * g.lda + sslot * 8;
  const bf16_t* w_src = reinterpret_cast<const bf16_t*>(g.W) + (size_t)(tile_n * BN + wave * 8 + srow) * g.ldw + sslot * 8;
  const size_t a_pass = (size_t)32 * g.lda, w_pass = (size_t)32 * g.ldw;

  auto stage = [&](int buf, int k0) {
    unsigned char* base = smem + buf * (2 * LDS_TILE_BYTES) + wave * 1024;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + p * a_pass + k0),
                                       (__attribute__((address_space(3))) void*)(base + p * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + p * w_pass + k0),
                                       (__attribute__((address_space(3))) void*)(base + LDS_TILE_BYTES + p * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK16;
  const int frow = lane & 15, fgrp = lane >> 4;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // tile kt landed (vmcnt(0) is part of the barrier with LDS-DMA in flight); buf (kt+1)&1 is free
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK16);
    const unsigned char* As = smem + (kt & 1) * (2 * LDS_TILE_BYTES);
    const unsigned char* Ws = As + LDS_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[4], wf[4];
      const int c = kk * 4 + fgrp;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8_t*>(As + r * 128 + ((c ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + frow;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + r * 128 + ((c ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
    }
  }

  const int m_wave = tile_m * BM + wm * 64, n_wave = tile_n * BN + wn * 64;
  if (g.amax_val) {
    epilogue_argmax(g, acc, m_wave, n_wave, lane);
    if (!g.out_f32) return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epilogue_tile<bf16_t>(g, acc[i][j], m_wave + i * 16 + fgrp * 4, n_wave + j * 16 + frow);
}

// ------------------------------------------------------------------------------------ f32
constexpr int BK32 = 16;
constexpr int LDF = BK32 + 1;    // padded LDS row (floats): conflict-free fragment reads

__global__ __launch_bounds__(256, 2) void gemm_f32_128x128x16(const GemmArgs g) {
  __shared__ float As[BM * LDF];
  __shared__ float Ws[BN * LDF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const float* A = reinterpret_cast<const float*>(g.A);
  const float* W = reinterpret_cast<const float*>(g.W);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fgrp = lane >> 4;
  // 128 rows x 16 floats = 512 float4 per operand; thread t loads float4 #t and #t+256
  const int lrow = tid >> 2, lcol = (tid & 3) * 4;
  for (int k0 = 0; k0 < g.K; k0 += BK32) {
    float4 a0 = *reinterpret_cast<const float4*>(A + (size_t)(tile_m * BM + lrow) * g.lda + k0 + lcol);
    float4 a1 = *reinterpret_cast<const float4*>(A + (size_t)(tile_m * BM + lrow + 64) * g.lda + k0 + lcol);
    float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)(tile_n * BN + lrow) * g.ldw + k0 + lcol);
    float4 w1 = *reinterpret_cast<const float4*>(W + (size_t)(tile_n * BN + lrow + 64) * g.ldw + k0 + lcol);
    __syncthreads();
    float* pa = As + lrow * LDF + lcol;
    pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w;
    pa += 64 * LDF;
    pa[0] = a1.x; pa[1] = a1.y; pa[2] = a1.z; pa[3] = a1.w;
    float* pw = Ws + lrow * LDF + lcol;
    pw[0] = w0.x; pw[1] = w0.y; pw[2] = w0.z; pw[3] = w0.w;
    pw += 64 * LDF;
    pw[0] = w1.x; pw[1] = w1.y; pw[2] = w1.z; pw[3] = w1.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK32 / 4; ++kk) {
      float af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = As[(wm * 64 + i * 16 + frow) * LDF + kk * 4 + fgrp];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = Ws[(wn * 64 + j * 16 + frow) * LDF + kk * 4 + fgrp];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wf[j], acc[i][j], 0, 0, 0);
    }
  }
  const int m_wave = tile_m * BM + wm * 64, n_wave = tile_n * BN + wn * 64;
  if (g.amax_val) {
    epilogue_argmax(g, acc, m_wave, n_wave, lane);
    if (!g.out_f32) return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epilogue_tile<float>(g, acc[i][j], m_wave + i * 16 + fgrp * 4, n_wave + j * 16 + frow);
}

__global__ void argmax_reduce_kernel(const float* __restrict__ val, const int32_t* __restrict__ idx, int M, int n_slabs,
                                     int32_t* __restrict__ ids) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int s = lane; s < n_slabs; s += 64) {
    const float v = val[(size_t)m * n_slabs + s];
    const int i = idx[(size_t)m * n_slabs + s];
    if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (lane == 0) ids[m] = bidx;
}

void check_args(const GemmArgs& g, int kstep, int elt) {
  ASR_REQUIRE(g.A && g.W, "gemm: null operand");
  ASR_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  ASR_REQUIRE(g.N % BN == 0, "gemm: N=%d must be a multiple of %d", g.N, BN);
  ASR_REQUIRE(g.K % kstep == 0, "gemm: K=%d must be a multiple of %d", g.K, kstep);
  ASR_REQUIRE((g.lda * elt) % 16 == 0 && (g.ldw * elt) % 16 == 0, "gemm: leading dims must be 16-byte multiples");
  if (g.add_t) ASR_REQUIRE(g.ld_add_t % 4 == 0, "gemm: ld_add_t must be a multiple of 4");
}

}  // namespace

void launch_gemm_bf16(const GemmArgs& g, hipStream_t s) {
  check_args(g, BK16, 2);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_128x128x64),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * LDS_TILE_BYTES));
    attr_set = true;
  }
  const int grid = ((g.M + BM - 1) / BM) * (g.N / BN);
  hipLaunchKernelGGL(gemm_bf16_128x128x64, dim3(grid), dim3(256), 4 * LDS_TILE_BYTES, s, g);
  HIP_CHECK(hipGetLastError());
}

void launch_gemm_f32(const GemmArgs& g, hipStream_t s) {
  check_args(g, BK32, 4);
  const int grid = ((g.M + BM - 1) / BM) * (g.N / BN);
  hipLaunchKernelGGL(gemm_f32_128x128x16, dim3(grid), dim3(256), 0, s, g);
  HIP_CHECK(hipGetLastError());
}

void launch_argmax_reduce(const float* val, const int32_t* idx, int M, int n_slabs, int32_t* ids, hipStream_t s) {
  hipLaunchKernelGGL(argmax_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, s, val, idx, M, n_slabs, ids);
  HIP_CHECK(hipGetLastError());
}
