// FP8 matrix-pipe GEMM for the opt-in low-bit mode (ASR_PRECISION_FP8MM): C[M][N] = (A8[M][K] W8[N][K]^T) * a_scale * w_scale[n] with e4m3 (OCP) operands
// on v_mfma_scale_f32_16x16x128_f8f6f4 -- the block-scaled instruction at unit block scales (e8m0 127), which runs at twice the bf16 rate; the power-of-
// two scales (one per weight row from quantize_rows_fp8, one per activation tensor) are applied in the epilogue. The reference's counterpart is its
// int8 / 4-bit MatMulNBits graphs (Optimize_ONNX_Common.py:55-60, Whisper/Optimize_ONNX.py:81-96).
//
// The kernel is the persistent ping-pong GEMM of gemm_pp.hip with three changes: (1) a K-step is 128 k = the same 128 BYTES per operand row, so units,
// ring, waits and barriers are untouched; (2) a lane's MFMA operand is 32 consecutive bytes of its row (k block lane >> 4) instead of two 16-byte
// fragments of different k sub-steps -- the staging source permutes the 16-byte slots of a row (src_slot) so that those 32 bytes are the SAME two
// conflict-free ds_read_b128 as before; (3) one 16x16x128 MFMA per (row fragment, column fragment) and phase: 8 per matrix segment (32 cycles each)
// for twice the flops of the 16 bf16 MFMAs. Epilogues: scale + bias + erf-GELU -> e4m3 bytes (saturating at 448; the next GEMM's operand), or
// scale + bias + f32 residual rows -> f32.
#include <algorithm>
#include "gemm_dev.h"

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

namespace {

constexpr int PP_T = 256;
constexpr int PP_UNIT = 128 * 128;
constexpr int PP_BUF = 4 * PP_UNIT;
#define PP_GLDS(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int ACT, int OUT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_ppp(const Fp8GemmArgs g, const int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int n_sat = 0;                                         // byte output: elements of this lane that met the e4m3 clamp
  const int tiles_n = g.N / PP_T, tiles_m = total_tiles / tiles_n;
  auto tile_of = [&](int vb, int& tm, int& tn) {
    const int tile = xcd_remap(vb, total_tiles);
    tm = tile / tiles_n; tn = tile % tiles_n;
    if (g.group_m > 1) {
      const int gsz = g.group_m * tiles_n;
      const int grp = tile / gsz, first = grp * g.group_m, local = tile - grp * gsz;
      const int rows_in = min(g.group_m, tiles_m - first);
      tm = first + local % rows_in;
      tn = local / rows_in;
    }
  };

  const int srow = lane >> 3;
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.A);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.W);
  // LDS position p of a row holds LOGICAL slot j = p ^ key, and logical slot j holds the row's bytes 32 (j & 3) + 16 (j >> 2) .. + 15: a lane's 32-byte
  // k block (bytes 32 fgrp ..) is then logical slots fgrp and fgrp + 4 -- the two conflict-free fragment reads of the bf16 kernel, unchanged
  auto src_slot = [](int j) { return ((j & 3) << 1) | (j >> 2); };
  const int a_slot = src_slot((lane & 7) ^ srow) << 4;
  const int w_slot = src_slot((lane & 7) ^ w_swz(wave * 8 + srow)) << 4;
  uint32_t a_src[2][2], w_src[2][2];
  auto set_src = [&](int tm, int tn) {                 // staging sources of the tile the stream is in
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        a_src[h][p] = (uint32_t)min(tm * PP_T + p * 128 + h * 64 + wave * 8 + srow, g.M - 1) * (uint32_t)g.lda + a_slot;
        w_src[h][p] = (uint32_t)(tn * PP_T + (2 * p + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + srow) * (uint32_t)g.ldw + w_slot;
      }
  };
  unsigned char* const lds_w = smem + wave * 1024;
  auto issue = [&](int u, int kt) {
    unsigned char* dst = lds_w + (kt & 1) * PP_BUF + u * PP_UNIT;
    const uint32_t k0 = (uint32_t)kt * 128u;
    if (u == 0) { PP_GLDS(Wb + (w_src[0][0] + k0), dst); PP_GLDS(Wb + (w_src[0][1] + k0), dst + 8192); }
    else if (u == 1) { PP_GLDS(Ab + (a_src[0][0] + k0), dst); PP_GLDS(Ab + (a_src[0][1] + k0), dst + 8192); }
    else if (u == 2) { PP_GLDS(Wb + (w_src[1][0] + k0), dst); PP_GLDS(Wb + (w_src[1][1] + k0), dst + 8192); }
    else { PP_GLDS(Ab + (a_src[1][0] + k0), dst); PP_GLDS(Ab + (a_src[1][1] + k0), dst + 8192); }
  };

  const int frow = lane & 15, fgrp = lane >> 4;
  int a_rd[2], w_rd[2];
  {
    const int ra = wr * 64 + frow, rw = wc * 32 + ((frow >> 2) << 3) + (frow & 3);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fgrp;
      a_rd[kk] = ra * 128 + ((c ^ (ra & 7)) << 4);
      w_rd[kk] = rw * 128 + ((c ^ w_swz(rw)) << 4);
    }
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  i32x8_t af[4], wf0[2], wf1[2];                     // a lane's 32-byte k block per fragment: the MFMA operand as it is (two ds_read_b128 fill its halves)
  const int nk = g.K / 128;

  typedef __attribute__((ext_vector_type(4))) int i32x4_t;
  auto rd_a = [&](const unsigned char* buf, int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i].lo = *reinterpret_cast<const i32x4_t*>(buf + (1 + 2 * h) * PP_UNIT + a_rd[0] + i * 2048);
      af[i].hi = *reinterpret_cast<const i32x4_t*>(buf + (1 + 2 * h) * PP_UNIT + a_rd[1] + i * 2048);
    }
  };
  auto rd_w = [&](const unsigned char* buf, int h, i32x8_t (&wf)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      wf[jj].lo = *reinterpret_cast<const i32x4_t*>(buf + 2 * h * PP_UNIT + w_rd[0] + jj * 512);
      wf[jj].hi = *reinterpret_cast<const i32x4_t*>(buf + 2 * h * PP_UNIT + w_rd[1] + jj * 512);
    }
  };
  auto mma = [&](int ha, int hb, const i32x8_t (&wf)[2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        acc[ha * 4 + i][hb * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[jj], af[i], acc[ha * 4 + i][hb * 2 + jj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto mem_end = [&]() {
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K-step = four phases: phases 0, 1 issue units 2, 3 of K-step kA, phases 2, 3 units 0, 1 of K-step kB; `turn`: the staging sources move to the
  // next tile between them (second-to-last K-step of a tile)
  int ntm = 0, ntn = 0;
  auto kstep = [&](const unsigned char* buf, int kA, int kB, bool turn) {
    rd_w(buf, 0, wf0); rd_a(buf, 0);
    issue(2, kA);
    mem_end();
    mma(0, 0, wf0);
    rd_w(buf, 1, wf1);
    issue(3, kA);
    mem_end();
    mma(0, 1, wf1);
    rd_a(buf, 1);
    if (turn) set_src(ntm, ntn);
    issue(0, kB);
    mem_end();
    mma(1, 1, wf1);
    issue(1, kB);
    mem_end();
    mma(1, 0, wf0);
  };
  // the tile's 256 bias values and 256 weight-row scales ride the operand stream into spare LDS (two slots, alternating per tile)
  int tpar = 0;
  auto stage_bias = [&](int tn_) {
    if (wave == 0) PP_GLDS(reinterpret_cast<const unsigned char*>(g.bias) + (size_t)tn_ * PP_T * 4 + lane * 16, smem + 2 * PP_BUF + tpar * 2048);
    if (wave == 1) PP_GLDS(reinterpret_cast<const unsigned char*>(g.w_scale) + (size_t)tn_ * PP_T * 4 + lane * 16, smem + 2 * PP_BUF + tpar * 2048 + 1024);
  };
  int vb = blockIdx.x, tm, tn;
  tile_of(vb, tm, tn);
  set_src(tm, tn);
  stage_bias(tn);
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0); issue(0, 1); issue(1, 1);
  wait_vmcnt<8>();
  __builtin_amdgcn_s_barrier();

  for (;;) {
    if (wr == 1) __builtin_amdgcn_s_barrier();           // the second group runs one segment behind the first
    __builtin_amdgcn_sched_barrier(0);
    const int nvb = vb + (int)gridDim.x;
    const bool has_next = nvb < total_tiles;
    ntm = tm; ntn = tn;                                  // last tile: the stream re-reads this tile's first K-steps (never consumed)
    if (has_next) tile_of(nvb, ntm, ntn);
    if (vb != (int)blockIdx.x) stage_bias(tn);
    for (int kt = 0; kt < nk; kt += 2) {                 // the last pair finishes this tile's stream and starts the next tile's
      const bool last = kt + 2 >= nk;
      kstep(smem, kt + 1, last ? 0 : kt + 2, last);
      kstep(smem + PP_BUF, last ? 0 : kt + 2, last ? 1 : kt + 3, false);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();           // pairs with the second group's last segment: both groups store together
    __builtin_amdgcn_sched_barrier(0);
    const int m_wave = tm * PP_T + wr * 128, n_wave = tn * PP_T + wc * 64;
    {
      const float* bias_l = reinterpret_cast<const float*>(smem + 2 * PP_BUF + tpar * 2048) - tn * PP_T;
      const float* scale_l = bias_l + 256;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = n_wave + p * 32 + fgrp * 8;
        float s8[8], b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s8[e] = scale_l[n + e] * g.a_scale; b8[e] = bias_l[n + e]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = m_wave + i * 16 + frow;
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * p][r]; v[4 + r] = acc[i][2 * p + 1][r]; }
          acc[i][2 * p] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][2 * p + 1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], s8[e], b8[e]);
          if constexpr (OUT == 1) {                                   // + f32 residual rows -> f32
            if (m < g.M) {
              const float* q = g.add + (size_t)m * g.ld_add + n;
              const float4 r0 = *reinterpret_cast<const float4*>(q), r1 = *reinterpret_cast<const float4*>(q + 4);
              float* o = g.out_f32 + (size_t)m * g.ld_out_f32 + n;
              *reinterpret_cast<float4*>(o) = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
              *reinterpret_cast<float4*>(o + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
            }
          } else {                                                    // activation -> e4m3 bytes (saturating), the next GEMM's operand
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              if constexpr (ACT == ACT_GELU_ERF) gelu_erf_fast2(v[e], v[e + 1]); else { v[e] = apply_act_ct<ACT>(v[e]); v[e + 1] = apply_act_ct<ACT>(v[e + 1]); }
              const float t0 = v[e] * g.out_inv_scale, t1 = v[e + 1] * g.out_inv_scale;
              v[e] = fminf(fmaxf(t0, -448.0f), 448.0f);
              v[e + 1] = fminf(fmaxf(t1, -448.0f), 448.0f);
              if (m < g.M) n_sat += (v[e] != t0) + (v[e + 1] != t1);          // (a NaN compares unequal too)
            }
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
            if (m < g.M) *reinterpret_cast<uint2*>(g.out8 + (size_t)m * g.ld_out8 + n) = make_uint2((unsigned)w0, (unsigned)w1);
          }
        }
      }
    }
    if (!has_next) break;
    vb = nvb; tm = ntm; tn = ntn; tpar ^= 1;
  }
  if constexpr (OUT == 0) {
    if (g.sat_count && n_sat) atomicAdd(g.sat_count, (unsigned long long)n_sat);          // (rare: a saturating operand is a mis-scaled one)
  }
  wait_vmcnt<0>();                                       // the units nobody reads must land before the LDS is handed to another workgroup
}


template <int ACT, int OUT>
void launch_fp8_inst(const Fp8GemmArgs& g, hipStream_t s) {
  constexpr int lds = 2 * PP_BUF + 4096;
  static PerDeviceOnce attr_once;
  static int n_cu[32] = {};
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_ppp<ACT, OUT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_CHECK(hipDeviceGetAttribute(&n_cu[dev & 31], hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int tiles_m = (g.M + PP_T - 1) / PP_T, tiles = tiles_m * (g.N / PP_T);
  Fp8GemmArgs gg = g;
  gg.group_m = tiles_m >= 16 ? 8 : 0;
  const int grid = std::min(tiles, n_cu[dev & 31] > 0 ? n_cu[dev & 31] : 256);
  hipLaunchKernelGGL((gemm_fp8_ppp<ACT, OUT>), dim3(grid), dim3(512), lds, s, gg, tiles);
  HIP_CHECK(hipGetLastError());
}

}  // namespace

void launch_gemm_fp8(const Fp8GemmArgs& g, hipStream_t s) {
  ASR_REQUIRE(g.A && g.W && g.w_scale && g.bias && g.M >= 1 && g.N % PP_T == 0 && g.K % 256 == 0 && g.lda % 16 == 0 && g.ldw % 16 == 0,
              "gemm_fp8: unsupported shape (M = %d, N = %d, K = %d)", g.M, g.N, g.K);
  ASR_REQUIRE((size_t)g.M * g.lda < ((size_t)1 << 32) && (size_t)g.N * g.ldw < ((size_t)1 << 32), "gemm_fp8: operands beyond 4 GiB");
  if (g.out8) {
    ASR_REQUIRE(!g.out_f32 && !g.add && g.ld_out8 % 8 == 0, "gemm_fp8: byte output takes no residual term");
    if (g.act == ACT_GELU_ERF) launch_fp8_inst<ACT_GELU_ERF, 0>(g, s);
    else if (g.act == ACT_GELU_TANH) launch_fp8_inst<ACT_GELU_TANH, 0>(g, s);
    else if (g.act == ACT_RELU) launch_fp8_inst<ACT_RELU, 0>(g, s);
    else launch_fp8_inst<ACT_NONE, 0>(g, s);
  } else {
    ASR_REQUIRE(g.out_f32 && g.add && g.act == ACT_NONE, "gemm_fp8: the f32 output is bias + residual, no activation");
    launch_fp8_inst<ACT_NONE, 1>(g, s);
  }
}
