// MFMA GEMMs for gfx950 with fused epilogues (bias / activation / residual adds / transposed store /
// row arg-max). C[M][N] = A[M][K] * W[N][K]^T.
//
// bf16 kernel: 128 x BN x 64 tile (BN = 128 or 64), 4 waves (2 x 2), v_mfma_f32_16x16x32_bf16.
//  * Operands are staged by global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a STAGES-deep LDS
//    ring. The LDS image of a DMA is lane-linear, so the 16-byte-slot XOR swizzle (slot ^= row & 7) that
//    makes the ds_read_b128 fragment reads conflict-free is applied on the per-lane GLOBAL source address
//    and again on the read address (cdna_hip_programming.md rule 21 / T2).
//  * One raw s_barrier per K-step; the DMA queue is never drained in the loop: `s_waitcnt vmcnt(G*ahead)`
//    retires only the tile about to be read (T3+T4; __syncthreads() would drain the queue).
//  * Workgroup ids are remapped so each XCD (private L2) walks a contiguous range of tiles (T1).
//  * Orientation. The product is computed as mfma(W-fragment, A-fragment): the C fragment then holds, per
//    lane, FOUR CONSECUTIVE COLUMNS of one output row, so bias / residual loads and the row-major bf16 / f32
//    stores are direct 8- / 16-byte accesses from registers -- no LDS transpose, no extra barrier. The
//    un-swapped orientation (four consecutive ROWS per lane) is used only for the transposed store (V^T).
//  * Epilogues are compile-time specialised (EPI / ACT) so they are straight-line code.
//
// f32 kernel: same tile on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) for verification mode.
#include <cstring>
#include <type_traits>
#include <string>
#include "gemm.h"
#include "gemm_dev.h"

// ---- tuning / ablation switches (environment), re-read by gemm_reload_env() at every session creation so that a test can flip them in-process
struct GemmEnv {
  bool t144 = true, t144w = true, t288w = true, big = true, pp = true, splitk = true, deep = true, skinny144 = false;
  bool amax_pp = true;                   // ASR_GEMM_AMAX_PP=0: the arg-max head on the 288 x 256 tiles instead of the persistent ping-pong kernel (CTC head at 64 x 8 s: 288 -> 217 us)
  int skinny_splitk = -1, tall_min = 16, skinny_max_plain = 32;
  int decode_nt = 0, decode_ks = 0;      // ASR_DECODE_NT / ASR_DECODE_KS: force the decode GEMM's column granule / split count (0 = the cost model)
  bool decode_attn_wave = true;          // ASR_DECODE_ATTN_WAVE=0: single-token self-attention on the general kernel
  int decode_rb = 0;                     // ASR_DECODE_RB: 0 = the plan decides, 1 = never split 33 .. 64 rows into two row blocks, 2 = always where it fits
  int skinny_nt = 2;                     // ASR_SKINNY_NT=1: one 16-column granule per workgroup of the skinny GEMM even where the output is wider than the chip
  bool decode_attn_online = true;        // ASR_DECODE_ATTN_ONLINE=0: single-token cross-attention on the general (two-pass) kernel
  int n_cus = 0;                         // CUs of the current device (the decode GEMM's one-round grid bound)
  bool loaded = false;
};
static GemmEnv g_env;
static thread_local const char* g_last_kernel = "";
// launches per kernel family since the last reset (host-side counters: a hipGraph replay does not re-count): lets a test assert
// WHICH tiling a session's batch geometry dispatched to
struct KernelCount { const char* name; long n; };
static KernelCount g_counts[16] = {};
static void note_kernel(const char* tag) {
  g_last_kernel = tag;
  for (KernelCount& c : g_counts) {
    if (!c.name) { c.name = tag; c.n = 1; return; }
    if (c.name == tag || !strcmp(c.name, tag)) { ++c.n; return; }
  }
}
void gemm_kernel_counts_reset() { for (KernelCount& c : g_counts) c = KernelCount{nullptr, 0}; }
int gemm_kernel_counts(char* buf, int cap) {         // "name=count;name=count;..."
  std::string out;
  for (const KernelCount& c : g_counts) if (c.name) out += std::string(c.name) + "=" + std::to_string(c.n) + ";";
  if (buf && cap > 0) snprintf(buf, cap, "%s", out.c_str());
  return (int)out.size();
}
static bool env_flag(const char* name, bool dflt) { const char* e = getenv(name); return e ? e[0] != '0' : dflt; }
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
void gemm_reload_env() {
  GemmEnv e;
  e.t144 = env_flag("ASR_GEMM_T144", true); e.t144w = env_flag("ASR_GEMM_T144W", true); e.t288w = env_flag("ASR_GEMM_T288W", true);
  e.amax_pp = env_flag("ASR_GEMM_AMAX_PP", true);
  e.big = env_flag("ASR_GEMM_BIG", true); e.pp = env_flag("ASR_GEMM_PP", true); e.splitk = env_flag("ASR_GEMM_SPLITK", true); e.deep = env_flag("ASR_GEMM_DEEP", true);
  e.skinny144 = getenv("ASR_SKINNY_M144") && getenv("ASR_SKINNY_M144")[0] == '1';
  e.skinny_splitk = env_int("ASR_SKINNY_SPLITK", -1); e.tall_min = env_int("ASR_GEMM_TALL_MIN", 16);
  e.skinny_max_plain = env_int("ASR_SKINNY_MAX_M", 32); e.skinny_nt = env_int("ASR_SKINNY_NT", 2); e.decode_rb = env_int("ASR_DECODE_RB", 0);
  e.decode_nt = env_int("ASR_DECODE_NT", 0); e.decode_ks = env_int("ASR_DECODE_KS", 0); e.decode_attn_wave = env_flag("ASR_DECODE_ATTN_WAVE", true); e.decode_attn_online = env_flag("ASR_DECODE_ATTN_ONLINE", true);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&e.n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || e.n_cus <= 0) e.n_cus = 256;
  e.loaded = true;
  g_env = e;
}
static const GemmEnv& genv() { if (!g_env.loaded) gemm_reload_env(); return g_env; }
const char* gemm_last_kernel() { return g_last_kernel; }
bool gemm_skinny144_enabled() { return genv().skinny144; }
int gemm_env_decode_nt() { return genv().decode_nt; }
int gemm_env_decode_ks() { return genv().decode_ks; }
bool gemm_env_decode_attn_wave() { return genv().decode_attn_wave; }
bool gemm_env_decode_attn_online() { return genv().decode_attn_online; }
int gemm_env_decode_rb() { return genv().decode_rb; }
int gemm_env_cus() { return genv().n_cus; }

namespace {

constexpr int BM = 128;
constexpr int BK16 = 64;                       // bf16 K-step


// ------------------------------------------------------------------------------------ bf16, ring-pipelined
template <int BN_, int STAGES, int ACT, int EPI, bool SWAP>
__global__ __launch_bounds__(256, 2) void gemm_bf16_pipe(const GemmArgs g0) {
  GemmArgs g = g0;
  if (g0.m_dev) g.M = min(g0.M, *g0.m_dev);
  if (g0.k_splits > 1) {                          // this workgroup's K range; partial products land in its slab of the workspace
    const int kc = g0.K / g0.k_splits, ks = blockIdx.y;
    g.A = reinterpret_cast<const bf16_t*>(g0.A) + (size_t)ks * kc;
    g.W = reinterpret_cast<const bf16_t*>(g0.W) + (size_t)ks * kc;
    g.K = kc;
    g.out_f32 = g0.out_f32 + (size_t)ks * g0.M * g0.ld_out_f32;
  }
  constexpr int WN = BN_ / 2;                     // columns per wave
  constexpr int NJ = WN / 16;                     // column fragments per wave
  constexpr int STAGE_BYTES = BM * 128 + BN_ * 128;
  constexpr int W_PASSES = BN_ / 32;              // LDS-DMA instructions per thread for the W tile
  constexpr int G = 4 + W_PASSES;                 // LDS-DMA instructions per thread per stage
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / BN_;
  // with a device-side row count only the leading row tiles do work: keep them interleaved over the XCDs instead of contiguous
  const int tile = g0.m_dev ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  if (g.group_m > 1) {       // wide N: consecutive workgroups share a W column tile over group_m row tiles (W is fetched once per group)
    const int tiles_m = gridDim.x / tiles_n, gsz = g.group_m * tiles_n;
    const int grp = tile / gsz, first = grp * g.group_m, local = tile - grp * gsz;
    const int rows_in = min(g.group_m, tiles_m - first);
    tile_m = first + local % rows_in;
    tile_n = local / rows_in;
  }
  if (tile_m * BM >= g.M) return;                  // (device-side row count) nothing to do for this row tile

  // staging: one wave-instruction moves 8 rows x 128 B (8 slots of 16 B) = 1 KiB, lane-linear in LDS
  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ srow;            // un-swizzle on the global side
  const bf16_t* a_src = reinterpret_cast<const bf16_t*>(g.A) + (size_t)(tile_m * BM + wave * 8 + srow) * g.lda + sslot * 8;
  const int wslot = (lane & 7) ^ w_swz(wave * 8 + srow);      // rows p*32 + wave*8 + srow: p*32 leaves the key bits alone
  const bf16_t* w_src = reinterpret_cast<const bf16_t*>(g.W) + (size_t)(tile_n * BN_ + wave * 8 + srow) * g.ldw + wslot * 8;
  const size_t a_pass = (size_t)32 * g.lda, w_pass = (size_t)32 * g.ldw;

  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + p * a_pass + k0),
                                       (__attribute__((address_space(3))) void*)(base + p * 4096), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + p * w_pass + k0),
                                       (__attribute__((address_space(3))) void*)(base + BM * 128 + p * 4096), 16, 0, 0);
  };

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK16;
  const int frow = lane & 15, fgrp = lane >> 4;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s * BK16);

  // per-lane fragment offsets inside a stage (swizzled), hoisted out of the K loop
  int a_off[2][4], w_off[2][NJ];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c = kk * 4 + fgrp;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + frow; a_off[kk][i] = r * 128 + ((c ^ (r & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = wn * WN + frag_col(j, frow); w_off[kk][j] = BM * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(nk, kt + STAGES - 1) - (kt + 1);      // tiles allowed to stay in flight
    if (STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * G>();
    else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // tile kt landed for every wave; every wave is done reading slot (kt-1) % S
    if (kt + STAGES - 1 < nk && !(g.dbg & 1)) stage((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK16);
    if (g.dbg & 2) continue;
    const unsigned char* St = smem + ((g.dbg & 1) ? 0 : (kt % STAGES)) * STAGE_BYTES;
    bf16x8_t af[2][4], wf[2][NJ];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kk][i] = *reinterpret_cast<const bf16x8_t*>(St + a_off[kk][i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[kk][j] = *reinterpret_cast<const bf16x8_t*>(St + w_off[kk][j]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], wf[kk][j], acc[i][j], 0, 0, 0);
        }
  }

  const int m_wave = tile_m * BM + wm * 64, n_wave = tile_n * BN_ + wn * WN;
  if ((g.dbg & 4) && acc[0][0][0] != 12345.678f) return;
  if constexpr (SWAP) epilogue_rows<bf16_t, ACT, EPI, NJ>(g, acc, m_wave, n_wave, lane);
  else epilogue_transposed<bf16_t, NJ>(g, acc, m_wave, n_wave, lane);
}

// ------------------------------------------------------------------------------------ bf16, 256 x 256 tile, 8 waves (large M)
// For the shapes of the Whisper / Qwen3-ASR stacks (M >= 8 k rows, N and K >= 1 k) the 128 x 64 tiles are bound by the L2 -> LDS
// operand stream: every workgroup re-reads (128 + 64) x K operands for 128 x 64 outputs = 43 flop / byte, i.e. 3.9 GB through the
// LDS-DMA paths for Whisper's fc1 (231 us at the ~17 TB/s the stream saturates at; measured 226 us). A 256 x 256 tile moves
// (256 + 256) x K for 256 x 256 outputs = 128 flop / byte. 8 waves as 4 x 2, each owning 64 x 128 outputs (32 C fragments = 128
// accumulator registers; per K step 24 fragment reads feed 64 MFMAs: 96 B / clk / CU of LDS reads at full MFMA rate); two 64 KiB
// stages (the whole 160 KiB LDS is one workgroup's: one workgroup per CU, two waves per SIMD), same LDS-DMA staging, slot swizzle
// and swapped orientation as the ring kernel above. Row indices are clamped to M - 1 on the global side, so the A operand needs
// no padding rows beyond M.
constexpr int BIG = 256;

template <int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_big(const GemmArgs g) {
  constexpr int NJ = 8;                           // 128 columns per wave
  constexpr int STAGE_BYTES = 2 * BIG * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / BIG;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;

  const int srow = lane >> 3;
  const int sslot = (lane & 7) ^ srow;
  const bf16_t* a_src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
    a_src[p] = reinterpret_cast<const bf16_t*>(g.A) + (size_t)min(tile_m * BIG + p * 64 + wave * 8 + srow, g.M - 1) * g.lda + sslot * 8;
  const int wslot = (lane & 7) ^ w_swz(wave * 8 + srow);      // rows p*64 + wave*8 + srow: p*64 leaves the key bits alone
  const bf16_t* w_src = reinterpret_cast<const bf16_t*>(g.W) + (size_t)(tile_n * BIG + wave * 8 + srow) * g.ldw + wslot * 8;
  const size_t w_pass = (size_t)64 * g.ldw;

  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[p] + k0),
                                       (__attribute__((address_space(3))) void*)(base + p * 8192), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + p * w_pass + k0),
                                       (__attribute__((address_space(3))) void*)(base + BIG * 128 + p * 8192), 16, 0, 0);
  };

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK16;
  const int frow = lane & 15, fgrp = lane >> 4;
  stage(0, 0);
  int a_off[2][4], w_off[2][NJ];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c = kk * 4 + fgrp;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + frow; a_off[kk][i] = r * 128 + ((c ^ (r & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = wn * 128 + frag_col(j, frow); w_off[kk][j] = BIG * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }
  }
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // tile kt landed for every wave; every wave is done reading the other slot
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK16);
    const unsigned char* St = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[4], wf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_off[kk][i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(St + w_off[kk][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  epilogue_rows<bf16_t, ACT, EPI, NJ>(g, acc, tile_m * BIG + wm * 64, tile_n * BIG + wn * 128, lane);
}

// ------------------------------------------------------------------------------------ bf16, 144 x 128 tile, 8 waves
// Row tiles of 144 = one 8 s window (137 rows padded to 144): M = 64 x 144 gives 64 row tiles, so an N = 512 GEMM is
// exactly 256 workgroups (one per CU) and N = 2048 exactly 1024 (two rounds of two co-resident workgroups) -- the
// 128-row tiling leaves 72 row tiles and a 12 % last round. 8 waves = 2 K-halves x 4 column groups of 32: a wave owns ALL
// 9 row fragments of its 32 columns (the A fragment is shared by 2 W fragments: 11 LDS reads per 18 MFMA) for ONE
// 32-wide half of every 64-wide K-step; the two halves are summed through LDS once per tile (the ring is dead by
// then), after which the K-half-0 waves finish row fragments 0..4 and the K-half-1 waves 5..8 with the register
// epilogue. STAGES = 4 (one workgroup per CU, counted vmcnt keeps two stages in flight) or 2 (two workgroups per CU).
constexpr int TM = 144, TN = 128, TMI = TM / 16;
constexpr int T_STAGE = (TM + TN) * 128;
constexpr int T_AI = TM / 8, T_NI = (TM + TN) / 8;     // LDS-DMA wave-instructions per stage: A part / total
constexpr int T_RED = TM * TN * 4;

template <int STAGES, int ACT, int EPI>
__global__ __launch_bounds__(512, (STAGES == 2 ? 4 : 2)) void gemm_bf16_t144(const GemmArgs g0) {
  GemmArgs g = g0;
  if (g0.m_dev) g.M = min(g0.M, *g0.m_dev);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, cg = wave & 3;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int tiles_n = g.N / TN;
  const int tile = g0.m_dev ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  if (tile_m * TM >= g.M) return;
  const int a_rows = (g0.M + 127) & ~127;                 // rows of A that may be read (padded allocation)

  const int srow = lane >> 3;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* src[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int ii = min(wave + 8 * t, T_NI - 1);
    if (ii < T_AI) {
      const int r = min(tile_m * TM + ii * 8 + srow, a_rows - 1);
      src[t] = Ab + (size_t)r * g.lda + (((lane & 7) ^ srow) << 3);
    } else {
      const int wr = (ii - T_AI) * 8 + srow;               // row inside the W tile
      src[t] = Wb + (size_t)(tile_n * TN + wr) * g.ldw + (((lane & 7) ^ w_swz(wr)) << 3);
    }
  }
  const bool five = wave + 32 < T_NI;                     // waves 0, 1 carry a fifth instruction per stage
  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * T_STAGE + wave * 1024;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                       (__attribute__((address_space(3))) void*)(base + t * 8192), 16, 0, 0);
    if (five)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[4] + k0),
                                       (__attribute__((address_space(3))) void*)(base + 4 * 8192), 16, 0, 0);
  };

  f32x4_t acc[TMI][2];
#pragma unroll
  for (int i = 0; i < TMI; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const int nk = g.K / BK16;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s * BK16);

  constexpr bool LN = EPI >= 0 && (EPI & E_LN) != 0;
  constexpr int T_MAIN = (STAGES * T_STAGE > T_RED) ? STAGES * T_STAGE : T_RED;
  float2* st_part = reinterpret_cast<float2*>(smem + T_MAIN);           // [2][144] partial (sum, sum of squares)
  float2* st_fin = st_part + 2 * TM;                                     // [144] (mean, rstd)
  const int st_row = tid % TM, st_half = tid / TM;                       // threads < 288 own (row, four 16-byte K slots)
  float st_s = 0.0f, st_ss = 0.0f;

  if constexpr (LN) {                                     // producer-side statistics: finalise (mean, rstd) while the first stages fly
    if (g.ln_stats_in && tid < TM) {
      const float2* sp = g.ln_stats_in + (size_t)min(tile_m * TM + tid, g.M - 1) * g.ln_slots;
      const float2 ss = sum_row_partials(sp, g.ln_slots);
      const float inv_d = 1.0f / (float)g.ln_dim;
      const float mean = ss.x * inv_d;
      const float var = fmaxf(ss.y * inv_d - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + g.ln_eps));
    }
  }
  const int c = kg * 4 + fgrp;                            // this wave's 16-byte K-chunk inside the 128-byte stage row
  const int a_off = frow * 128 + ((c ^ (frow & 7)) << 4);
  int w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = cg * 32 + frag_col(j, frow); w_off[j] = TM * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }

  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(nk, kt + STAGES - 1) - (kt + 1);
    if (STAGES >= 4 && ahead >= 2) { if (five) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
    else if (STAGES >= 3 && ahead >= 1) { if (five) wait_vmcnt<5>(); else wait_vmcnt<4>(); }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk && !(g.dbg & 1)) stage((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK16);
    if (g.dbg & 2) continue;
    const unsigned char* St = smem + (kt % STAGES) * T_STAGE;
    if constexpr (LN) {
      if (!g.ln_stats_in && tid < 2 * TM) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 raw = *reinterpret_cast<const uint4*>(St + st_row * 128 + (((4 * st_half + q) ^ (st_row & 7)) << 4));
          const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(wds[e] << 16), hi = __uint_as_float(wds[e] & 0xffff0000u);
            st_s += lo + hi;
            st_ss = fmaf(lo, lo, fmaf(hi, hi, st_ss));
          }
        }
      }
    }
    const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(St + w_off[0]);
    const bf16x8_t w1 = *reinterpret_cast<const bf16x8_t*>(St + w_off[1]);
    bf16x8_t af[TMI];
#pragma unroll
    for (int i = 0; i < TMI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_off + i * 2048);
#pragma unroll
    for (int i = 0; i < TMI; ++i) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af[i], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af[i], acc[i][1], 0, 0, 0);
    }
  }
  if ((g.dbg & 4) && acc[0][0][0] != 12345.678f) return;

  // ---- sum the two K-halves: half 1 hands row fragments 0..4 to half 0, half 0 hands 5..8 to half 1
  if constexpr (LN) { if (!g.ln_stats_in && tid < 2 * TM) st_part[st_half * TM + st_row] = make_float2(st_s, st_ss); }
  __syncthreads();                                        // ring dead
  if constexpr (LN) {
    if (!g.ln_stats_in && tid < TM) {
      const float2 p0 = st_part[tid], p1 = st_part[TM + tid];
      const float inv_d = 1.0f / (float)g.ln_dim;
      const float mean = (p0.x + p1.x) * inv_d;
      const float var = fmaxf((p0.y + p1.y) * inv_d - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + g.ln_eps));
    }
  }
  float4* red = reinterpret_cast<float4*>(smem);
  constexpr int LO = 5, HI = TMI - LO;
  if (kg == 1) {
#pragma unroll
    for (int i = 0; i < LO; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        red[((cg * LO + i) * 2 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  } else {
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        red[4 * LO * 2 * 64 + ((cg * HI + i) * 2 + j) * 64 + lane] =
            make_float4(acc[LO + i][j][0], acc[LO + i][j][1], acc[LO + i][j][2], acc[LO + i][j][3]);
  }
  __syncthreads();
  const int n_wave = tile_n * TN + cg * 32;
  if (kg == 0) {
    f32x4_t fin[LO][2];
#pragma unroll
    for (int i = 0; i < LO; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 o = red[((cg * LO + i) * 2 + j) * 64 + lane];
        fin[i][j] = f32x4_t{acc[i][j][0] + o.x, acc[i][j][1] + o.y, acc[i][j][2] + o.z, acc[i][j][3] + o.w};
      }
    epilogue_rows<bf16_t, ACT, EPI, 2, LO>(g, fin, tile_m * TM, n_wave, lane, st_fin);
  } else {
    f32x4_t fin[HI][2];
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 o = red[4 * LO * 2 * 64 + ((cg * HI + i) * 2 + j) * 64 + lane];
        fin[i][j] = f32x4_t{o.x + acc[LO + i][j][0], o.y + acc[LO + i][j][1], o.z + acc[LO + i][j][2], o.w + acc[LO + i][j][3]};
      }
    epilogue_rows<bf16_t, ACT, EPI, 2, HI>(g, fin, tile_m * TM + LO * 16, n_wave, lane, st_fin + LO * 16);
  }
}

// ------------------------------------------------------------------------------------ bf16, 144 x 256 tile, 8 waves
// The 144 x 128 tiles move (144 + 128) operand rows through the LDS-DMA path per 144 x 128 outputs = 68 flop / byte (ffn1 of the SANM
// block: 285 MB of L2 -> LDS traffic per launch, ~8 TB/s chip-wide, MFMA pipe 19 % busy -- profiles/r01_mfma_util.json). 144 x 256 tiles
// move (144 + 256) rows per twice the outputs = 92 flop / byte and read 13 LDS fragments per 36 MFMA instead of 11 per 18. Same wave
// layout (2 K-halves x 4 column groups, now 64 columns each: 144 accumulator registers), three 50 KB stages, one workgroup per CU.
// Used where N / 256 column tiles still fill the chip (ffn1: 64 windows x 8 = 512 tiles = two full rounds); LayerNorm-folded epilogue
// with producer-side statistics only. Measured (SenseVoice B = 64): timed alone the launch is no faster (35.7 vs 34.3 us: one
// workgroup per CU exposes the tile prologue / epilogue), back to back inside the step it is -- 8.99 vs 9.20 ms per step, A/B twice --
// the quarter less L2 traffic is what the neighbouring launches gain. ASR_GEMM_T144W=0 turns it off.
constexpr int TW = 256;
constexpr int TW_STAGE = (TM + TW) * 128;
constexpr int TW_NI = (TM + TW) / 8;                    // 50 LDS-DMA wave-instructions per stage: waves 0, 1 issue 7, the others 6
constexpr int TW_RED = TM * TW * 4;
constexpr int TW_STAGES = 3;

template <int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t144w(const GemmArgs g0) {
  GemmArgs g = g0;
  if (g0.m_dev) g.M = min(g0.M, *g0.m_dev);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, cg = wave & 3;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int tiles_n = g.N / TW;
  const int tile = g0.m_dev ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  if (tile_m * TM >= g.M) return;
  const int a_rows = (g0.M + 127) & ~127;

  const int srow = lane >> 3;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* src[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int ii = min(wave + 8 * t, TW_NI - 1);
    if (ii < T_AI) {
      const int r = min(tile_m * TM + ii * 8 + srow, a_rows - 1);
      src[t] = Ab + (size_t)r * g.lda + (((lane & 7) ^ srow) << 3);
    } else {
      const int wr = (ii - T_AI) * 8 + srow;
      src[t] = Wb + (size_t)(tile_n * TW + wr) * g.ldw + (((lane & 7) ^ w_swz(wr)) << 3);
    }
  }
  const bool seven = wave + 48 < TW_NI;
  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * TW_STAGE + wave * 1024;
#pragma unroll
    for (int t = 0; t < 6; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                       (__attribute__((address_space(3))) void*)(base + t * 8192), 16, 0, 0);
    if (seven)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[6] + k0),
                                       (__attribute__((address_space(3))) void*)(base + 6 * 8192), 16, 0, 0);
  };

  f32x4_t acc[TMI][4];
#pragma unroll
  for (int i = 0; i < TMI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nk = g.K / BK16;
#pragma unroll
  for (int s = 0; s < TW_STAGES - 1; ++s)
    if (s < nk) stage(s, s * BK16);

  constexpr int T_MAIN = (TW_STAGES * TW_STAGE > TW_RED) ? TW_STAGES * TW_STAGE : TW_RED;
  float2* st_fin = reinterpret_cast<float2*>(smem + T_MAIN);             // [144] (mean, rstd) from the producer's row statistics
  if (tid < TM) {
    const float2* sp = g.ln_stats_in + (size_t)min(tile_m * TM + tid, g.M - 1) * g.ln_slots;
    const float2 ss = sum_row_partials(sp, g.ln_slots);
    const float inv_d = 1.0f / (float)g.ln_dim;
    const float mean = ss.x * inv_d;
    const float var = fmaxf(ss.y * inv_d - mean * mean, 0.0f);
    st_fin[tid] = make_float2(mean, rsqrtf(var + g.ln_eps));
  }
  const int c = kg * 4 + fgrp;
  const int a_off = frow * 128 + ((c ^ (frow & 7)) << 4);
  int w_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int r = cg * 64 + frag_col(j, frow); w_off[j] = TM * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }

  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(nk, kt + TW_STAGES - 1) - (kt + 1);
    if (ahead >= 1) { if (seven) wait_vmcnt<7>(); else wait_vmcnt<6>(); }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + TW_STAGES - 1 < nk) stage((kt + TW_STAGES - 1) % TW_STAGES, (kt + TW_STAGES - 1) * BK16);
    const unsigned char* St = smem + (kt % TW_STAGES) * TW_STAGE;
    bf16x8_t wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(St + w_off[j]);
    bf16x8_t af[TMI];
#pragma unroll
    for (int i = 0; i < TMI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_off + i * 2048);
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  }

  // ---- sum the two K-halves: half 1 hands row fragments 0..4 to half 0, half 0 hands 5..8 to half 1
  __syncthreads();                                        // ring dead
  float4* red = reinterpret_cast<float4*>(smem);
  constexpr int LO = 5, HI = TMI - LO;
  if (kg == 1) {
#pragma unroll
    for (int i = 0; i < LO; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        red[((cg * LO + i) * 4 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  } else {
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        red[4 * LO * 4 * 64 + ((cg * HI + i) * 4 + j) * 64 + lane] =
            make_float4(acc[LO + i][j][0], acc[LO + i][j][1], acc[LO + i][j][2], acc[LO + i][j][3]);
  }
  __syncthreads();
  const int n_wave = tile_n * TW + cg * 64;
  if (kg == 0) {
    f32x4_t fin[LO][4];
#pragma unroll
    for (int i = 0; i < LO; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 o = red[((cg * LO + i) * 4 + j) * 64 + lane];
        fin[i][j] = f32x4_t{acc[i][j][0] + o.x, acc[i][j][1] + o.y, acc[i][j][2] + o.z, acc[i][j][3] + o.w};
      }
    epilogue_rows<bf16_t, ACT, EPI, 4, LO>(g, fin, tile_m * TM, n_wave, lane, st_fin);
  } else {
    f32x4_t fin[HI][4];
#pragma unroll
    for (int i = 0; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 o = red[4 * LO * 4 * 64 + ((cg * HI + i) * 4 + j) * 64 + lane];
        fin[i][j] = f32x4_t{o.x + acc[LO + i][j][0], o.y + acc[LO + i][j][1], o.z + acc[LO + i][j][2], o.w + acc[LO + i][j][3]};
      }
    epilogue_rows<bf16_t, ACT, EPI, 4, HI>(g, fin, tile_m * TM + LO * 16, n_wave, lane, st_fin + LO * 16);
  }
}

// ------------------------------------------------------------------------------------ bf16, 288 x 256 tile, 8 waves
// Two 8 s windows x 256 columns per workgroup: (288 + 256) operand rows per 288 x 256 outputs = 135 flop / byte through the LDS-DMA
// path (the 144 x 256 tile: 92). 8 waves = 2 row groups (one window each) x 4 column groups of 64: every wave owns 9 x 4 C fragments
// (144 accumulator registers) over the WHOLE K-step, so no cross-wave reduction; two 68 KB stages, one workgroup per CU. For the
// LayerNorm-folded FFN-1 when ceil(M / 288) x N / 256 makes whole rounds of 256 workgroups (B = 64: 32 x 8 = 256).
constexpr int TM2 = 288, TM2I = 9;
constexpr int T2_STAGE = (TM2 + TW) * 128;
constexpr int T2_AI = TM2 / 8, T2_NI = (TM2 + TW) / 8;   // 36 A pieces of 68 LDS-DMA wave-instructions per stage: waves 0..3 issue 9, 4..7 issue 8

template <int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t288w(const GemmArgs g0) {
  GemmArgs g = g0;
  if (g0.m_dev) g.M = min(g0.M, *g0.m_dev);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 2, cg = wave & 3;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int tiles_n = g.N / TW;
  const int tile = g0.m_dev ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  if (tile_m * TM2 >= g.M) return;
  const int a_rows = (g0.M + 127) & ~127;

  const int srow = lane >> 3;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* src[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ii = min(wave + 8 * t, T2_NI - 1);
    if (ii < T2_AI) {
      const int r = min(tile_m * TM2 + ii * 8 + srow, a_rows - 1);
      src[t] = Ab + (size_t)r * g.lda + (((lane & 7) ^ srow) << 3);
    } else {
      const int wr = (ii - T2_AI) * 8 + srow;
      src[t] = Wb + (size_t)(tile_n * TW + wr) * g.ldw + (((lane & 7) ^ w_swz(wr)) << 3);
    }
  }
  const bool nine = wave + 64 < T2_NI;
  auto stage = [&](int slot, int k0) {
    unsigned char* base = smem + slot * T2_STAGE + wave * 1024;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                       (__attribute__((address_space(3))) void*)(base + t * 8192), 16, 0, 0);
    if (nine)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[8] + k0),
                                       (__attribute__((address_space(3))) void*)(base + 8 * 8192), 16, 0, 0);
  };

  f32x4_t acc[TM2I][4];
#pragma unroll
  for (int i = 0; i < TM2I; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nk = g.K / BK16;
  stage(0, 0);

  float2* st_fin = reinterpret_cast<float2*>(smem + 2 * T2_STAGE);       // [288] (mean, rstd) from the producer's row statistics
  if constexpr ((EPI & E_LN) != 0) {
    if (tid < TM2) {
      const float2* sp = g.ln_stats_in + (size_t)min(tile_m * TM2 + tid, g.M - 1) * g.ln_slots;
      const float2 ss = sum_row_partials(sp, g.ln_slots);
      const float inv_d = 1.0f / (float)g.ln_dim;
      const float mean = ss.x * inv_d;
      const float var = fmaxf(ss.y * inv_d - mean * mean, 0.0f);
      st_fin[tid] = make_float2(mean, rsqrtf(var + g.ln_eps));
    }
  }
  int a_off[2], w_off[2][4];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c = kk * 4 + fgrp;                          // 16-byte K-chunk of this lane inside the 128-byte stage row
    a_off[kk] = (rg * 144 + frow) * 128 + ((c ^ (frow & 7)) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = cg * 64 + frag_col(j, frow); w_off[kk][j] = TM2 * 128 + r * 128 + ((c ^ w_swz(r)) << 4); }
  }

  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK16);
    const unsigned char* St = smem + (kt & 1) * T2_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(St + w_off[kk][j]);
      bf16x8_t af[TM2I];
#pragma unroll
      for (int i = 0; i < TM2I; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(St + a_off[kk] + i * 2048);
#pragma unroll
      for (int i = 0; i < TM2I; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                                        // st_fin visible (written before the loop; the loop's barriers already order it)
  epilogue_rows<bf16_t, ACT, EPI, 4, TM2I>(g, acc, tile_m * TM2 + rg * 144, tile_n * TW + cg * 64, lane, st_fin + rg * 144);
}

// ------------------------------------------------------------------------------------ bf16, skinny M (decode)
// out[M <= 64][N] = A[M][K] W[N][K]^T: pure weight streaming. One workgroup = 16 output columns x all rows; its 8 waves
// split K eight ways, each streaming its slice of the 16 weight rows straight from HBM into MFMA A-fragments
// (16 bytes per lane, no LDS staging: the weights are read exactly once chip-wide), activations come from L2.
// Partial sums are reduced across the waves through LDS; the swapped orientation leaves 4 consecutive columns per lane
// for the bias / residual / store epilogue. With g.ln_x the A rows are LayerNorm(ln_x) computed in the prologue.
constexpr int SK_WAVES = 8;

typedef __attribute__((ext_vector_type(2))) unsigned int sk_u32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 sk_bf16x2_hw_t;
// 8 e4m3 bytes -> one bf16x8 MFMA fragment (v_cvt_scalef32_pk_bf16_fp8, scale 1: exact)
__device__ __forceinline__ bf16x8_t sk_fp8x8_to_bf16x8(sk_u32x2_t r) {
  union { bf16x8_t v; sk_bf16x2_hw_t p[4]; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, false); u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, true);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, false); u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, true);
  return u.v;
}

struct sk_w4_t { unsigned r; unsigned e; };       // MXFP4: a lane's 8 nibbles of one K-step + the block's e8m0 scale byte
__device__ __forceinline__ bf16x8_t sk_fp4x8_to_bf16x8(sk_w4_t w) {
  const float sc = __uint_as_float(w.e << 23);
  union { bf16x8_t v; sk_bf16x2_hw_t p[4]; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w.r, sc, 0); u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w.r, sc, 1);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w.r, sc, 2); u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w.r, sc, 3);
  return u.v;
}

template <int MT, bool LN, int WQ = 0, int NT = 1>     // NT: 16-column granules per workgroup (2: an activation fragment read from L2 feeds two weight granules -- outputs wider than the chip, see launch_skinny); LN: the LayerNorm prologue (own instance: its registers / LDS allow one workgroup per CU only); WQ: 1 = e4m3 weight bytes (GemmArgs::W8), 2 = MXFP4 (GemmArgs::W4)
__global__ __launch_bounds__(512, !LN && NT == 1 && MT >= 2 && MT <= 4 ? 4 : 2) void gemm_bf16_skinny(const GemmArgs g) {   // <= 128 VGPRs: two workgroups share a CU
  constexpr bool W8 = WQ == 1, W4 = WQ == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  static_assert(NT == 1 || !LN, "the LayerNorm instance keeps one granule per workgroup");
  const int n0 = blockIdx.x * 16 * NT;
  const int KS = g.sk_splits > 1 ? g.sk_splits : 1, ks = blockIdx.y;
  const int kslice = g.K / (SK_WAVES * KS);               // multiple of 32 (host-checked)
  // Every workgroup reads the SAME activation rows: if all of them walked K in the same order, each L2 channel would be hit by all
  // CUs of its XCD at the same moment while the others idle. Rotating the wave -> K-slice assignment by the workgroup index spreads
  // the simultaneous requests over the slices (the cross-wave sum below is order-stable per workgroup, so results stay reproducible).
  const int kw = (wave + (int)blockIdx.x) & (SK_WAVES - 1);
  const int k_begin = (ks * SK_WAVES + kw) * kslice;
  using wfrag_t = typename std::conditional<W8, sk_u32x2_t, typename std::conditional<W4, sk_w4_t, bf16x8_t>::type>::type;       // a lane's 8 weights of one K-step: 8 bytes, 4 bytes + a scale, or 16
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(g.W) + (size_t)(n0 + frow) * g.ldw + k_begin + fgrp * 8;
  const unsigned char* wp8 = g.W8 + (size_t)(n0 + frow) * g.ldw8 + k_begin + fgrp * 8;
  const unsigned char* wp4 = g.W4 + (size_t)(n0 + frow) * (g.K >> 1) + ((k_begin + fgrp * 8) >> 1);
  const unsigned char* sp4 = g.w_scale4 + (size_t)(n0 + frow) * (g.K >> 5) + (k_begin >> 5);
  auto w_load = [&](int k, int j) -> wfrag_t {          // granule j of this workgroup: weight rows n0 + 16 j ..
    if constexpr (W4) return sk_w4_t{__builtin_nontemporal_load(reinterpret_cast<const unsigned*>(wp4 + (size_t)j * 16 * (g.K >> 1) + (k >> 1))), (unsigned)sp4[(size_t)j * 16 * (g.K >> 5) + (k >> 5)]};
    else
    if constexpr (W8) return __builtin_nontemporal_load(reinterpret_cast<const sk_u32x2_t*>(wp8 + (size_t)j * 16 * g.ldw8 + k));
    else return __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)j * 16 * g.ldw + k));
  };

  constexpr int U = MT >= 4 ? 4 : 8;                    // K-steps per trip: U x 16-byte weight loads in flight per lane (4 row tiles: fewer, to stay under 128 VGPRs => two workgroups per CU)
  // the weight stream does not depend on the activations: start it before the LayerNorm prologue
  wfrag_t wf0[NT][U];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u * 32 < kslice) wf0[j][u] = w_load(u * 32, j);

  // the residual term of the epilogue (wave w finishes row tiles w, w + 8) is requested now: its L2 round trip hides behind the stream
  float4 addv[NT][(MT + SK_WAVES - 1) / SK_WAVES];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < (MT + SK_WAVES - 1) / SK_WAVES; ++t) {
      const int m = (wave + t * SK_WAVES) * 16 + frow;
      addv[j][t] = (g.add && wave + t * SK_WAVES < MT && m < g.M) ? *reinterpret_cast<const float4*>(g.add + (size_t)m * g.ld_add + n0 + j * 16 + fgrp * 4)
                                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }

  // ---- A source: bf16 rows, or LayerNorm(ln_x) built here into LDS as bf16 [MT*16][K]
  const bf16_t* A;
  int lda;
  if constexpr (LN) {                                    // (<= 32 rows: beyond that every workgroup redoing all rows costs more than a launch)
    {
      bf16_t* An = reinterpret_cast<bf16_t*>(smem + SK_WAVES * MT * NT * 1024);
      // wave w normalises rows w, w+8, ...: ALL of its rows are fetched in one batch of float4 loads (one L2 round trip),
      // then mean / variance / output come from registers.
      constexpr int RPW = MT * 16 / SK_WAVES;        // rows per wave
      constexpr int KV = 5;                          // float4 per lane per row: K <= 1280
      float4 v[RPW][KV];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int m = wave + rr * SK_WAVES;
#pragma unroll
        for (int q = 0; q < KV; ++q) {
          const int k = q * 256 + lane * 4;
          v[rr][q] = (m < g.M && k < g.K) ? *reinterpret_cast<const float4*>(g.ln_x + (size_t)m * g.ld_ln_x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int m = wave + rr * SK_WAVES;
        float s1 = 0.f;
#pragma unroll
        for (int q = 0; q < KV; ++q) s1 += (v[rr][q].x + v[rr][q].y) + (v[rr][q].z + v[rr][q].w);
        const float mean = wave_sum(s1) / (float)g.K;
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < KV; ++q) {
          if (q * 256 + lane * 4 < g.K) {
            const float a = v[rr][q].x - mean, b = v[rr][q].y - mean, c = v[rr][q].z - mean, d = v[rr][q].w - mean;
            s2 += (a * a + b * b) + (c * c + d * d);
          }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)g.K + g.ln_eps);
        bf16_t* o = An + (size_t)m * g.K;
#pragma unroll
        for (int q = 0; q < KV; ++q) {
          const int k = q * 256 + lane * 4;
          if (k < g.K) {
            float y0 = (v[rr][q].x - mean) * rstd, y1 = (v[rr][q].y - mean) * rstd, y2 = (v[rr][q].z - mean) * rstd, y3 = (v[rr][q].w - mean) * rstd;
            if (g.ln_gamma) {
              const float4 ga = *reinterpret_cast<const float4*>(g.ln_gamma + k), be = *reinterpret_cast<const float4*>(g.ln_beta + k);
              y0 = y0 * ga.x + be.x; y1 = y1 * ga.y + be.y; y2 = y2 * ga.z + be.z; y3 = y3 * ga.w + be.w;
            }
            if (m >= g.M) { y0 = y1 = y2 = y3 = 0.f; }
            uint2 w2;
            w2.x = pack_bf16x2(y0, y1);
            w2.y = pack_bf16x2(y2, y3);
            *reinterpret_cast<uint2*>(o + k) = w2;
          }
        }
      }
      __syncthreads();
      A = An;
      lda = g.K;
    }
  } else {
    A = reinterpret_cast<const bf16_t*>(g.A);
    lda = g.lda;
  }
  const bf16_t* ap = A + (size_t)frow * lda + k_begin + fgrp * 8;

  f32x4_t acc[NT][MT];
  f32x4_t gram[MT];                                      // a_rms: A A^T of the row tile on the (idle) MFMA pipe; its diagonal is sum(x^2)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    gram[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool a_rms = g.a_rms_eps > 0.0f;
  // Two register sets of U weight fragments: the loads of trip t + 1 are issued BEFORE the MFMAs of trip t, so a long K slice (fc2:
  // K = 5120 -> 20 K-steps per wave = 3 trips) pays one HBM round trip, not one per trip (19.2 -> ~10 us per launch at 32 rows).
  auto load_set = [&](wfrag_t (&wf)[NT][U], int k) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k + u * 32 < kslice) wf[j][u] = w_load(k + u * 32, j);
  };
  auto mma_set = [&](const wfrag_t (&wf)[NT][U], int k) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + u * 32 < kslice) {
        bf16x8_t wv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (W8) wv[j] = sk_fp8x8_to_bf16x8(wf[j][u]); else if constexpr (W4) wv[j] = sk_fp4x8_to_bf16x8(wf[j][u]); else wv[j] = wf[j][u];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          // rows past M are never stored: their lanes request nothing from memory (at one row the 16-row tile pulled 16 x the bytes through this CU's 64 B / clk path;
          // csrc/decode_gemm.hip, profiles/r05_decode_gemm_clock.txt). The LayerNorm instance reads its rows from LDS, where rows past M are zero already.
          bf16x8_t af = {0, 0, 0, 0, 0, 0, 0, 0};
          if (LN || i * 16 + frow < g.M) af = *reinterpret_cast<const bf16x8_t*>(ap + (size_t)i * 16 * lda + k + u * 32);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[j], af, acc[j][i], 0, 0, 0);   // D[n = 4 fgrp + r][m = frow]
          if (a_rms) gram[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, af, gram[i], 0, 0, 0);
        }
      }
    }
  };
  wfrag_t wf1[NT][U];
  for (int k = 0; k < kslice; k += 64 * U) {
    if (k + 32 * U < kslice) load_set(wf1, k + 32 * U);
    mma_set(wf0, k);
    if (k + 64 * U < kslice) load_set(wf0, k + 64 * U);
    if (k + 32 * U < kslice) mma_set(wf1, k + 32 * U);
  }
  // ---- cross-wave reduction
  float4* red = reinterpret_cast<float4*>(smem);                                     // [granule][wave][MT][64]
  float (*ss_red)[MT * 16] = reinterpret_cast<float (*)[MT * 16]>(smem + SK_WAVES * MT * NT * 1024);    // a_rms only (excludes ln_x)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) red[((j * SK_WAVES + wave) * MT + i) * 64 + lane] = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
  if (a_rms) {
#pragma unroll
    for (int i = 0; i < MT; ++i)                          // D[4 fgrp + r][frow]: the diagonal entry of row frow sits in lane group frow / 4
      if (fgrp == (frow >> 2)) ss_red[wave][i * 16 + frow] = gram[i][frow & 3];
  }
  __syncthreads();
  const int rows16 = MT * 16;
  float4 sums[NT][(MT + SK_WAVES - 1) / SK_WAVES];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < (MT + SK_WAVES - 1) / SK_WAVES; ++t) {                        // wave w finishes row tiles w, w + 8
      const int i = wave + t * SK_WAVES;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < MT) {
        sum = red[(j * SK_WAVES * MT + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < SK_WAVES; ++w) {
          const float4 q = red[((j * SK_WAVES + w) * MT + i) * 64 + lane];
          sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
        }
      }
      sums[j][t] = sum;
    }
  if constexpr (NT == 1) if (KS > 1) {                      // hand the partial sums over; the last workgroup of this column granule finishes.
    // Cross-XCD visibility without a release fence (on this part __threadfence() writes back / invalidates a whole L2): the
    // partials and the ticket are relaxed AGENT-scope atomics (sc1: written through to / read from the memory side), and every
    // thread waits for its own stores to complete before the workgroup takes its ticket.
#pragma unroll
    for (int t = 0; t < (MT + SK_WAVES - 1) / SK_WAVES; ++t) {
      const int i = wave + t * SK_WAVES;
      if (i < MT) {        // one 16-byte write-through store per lane (four 4-byte sc1 stores are four fabric writes: ~6x the time per byte)
        float* dst = g.sk_ws + ((size_t)ks * rows16 + i * 16 + frow) * g.N + n0 + fgrp * 4;
        const f32x4_t v = {sums[0][t].x, sums[0][t].y, sums[0][t].z, sums[0][t].w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // every wave is done with `red`: its first word carries the verdict
    int* sk_last = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(g.sk_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *sk_last = ticket == KS - 1;
      if (ticket == KS - 1) __hip_atomic_store(g.sk_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*sk_last) return;
#pragma unroll
    for (int t = 0; t < (MT + SK_WAVES - 1) / SK_WAVES; ++t) {
      const int i = wave + t * SK_WAVES;
      if (i < MT) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        // fixed order => bit-reproducible; 16-byte sc1 loads (L1 bypassed: the partials were stored write-through). Up to eight splits are requested together
        // and waited for once (the wait names the registers, so no add moves above it)
        for (int s0 = 0; s0 < KS; s0 += 8) {
          f32x4_t q[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float* src = g.sk_ws + ((size_t)min(s0 + j, KS - 1) * rows16 + i * 16 + frow) * g.N + n0 + fgrp * 4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(q[j]) : "v"(src) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (s0 + j < KS) { sum.x += q[j][0]; sum.y += q[j][1]; sum.z += q[j][2]; sum.w += q[j][3]; }
        }
        sums[0][t] = sum;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
  for (int t = 0; t < (MT + SK_WAVES - 1) / SK_WAVES; ++t) {
    const int i = wave + t * SK_WAVES;
    if (i >= MT) continue;
    float4 sum = sums[j][t];
    const int m = i * 16 + frow, n = n0 + j * 16 + fgrp * 4;
    if (m >= g.M) continue;
    if constexpr (W8) { const float4 sc = *reinterpret_cast<const float4*>(g.w_scale + n); sum.x *= sc.x; sum.y *= sc.y; sum.z *= sc.z; sum.w *= sc.w; }   // (powers of two: exact)
    if (a_rms) {                                          // fixed summation order over the waves
      float t = ss_red[0][m];
#pragma unroll
      for (int w = 1; w < SK_WAVES; ++w) t += ss_red[w][m];
      const float r = rsqrtf(t / (float)g.K + g.a_rms_eps);
      sum.x *= r; sum.y *= r; sum.z *= r; sum.w *= r;
    }
    if (g.act == ACT_SWIGLU) {                            // columns (gate_j, up_j, gate_j+1, up_j+1)
      const float y0 = sum.x / (1.0f + __expf(-sum.x)) * sum.y, y1 = sum.z / (1.0f + __expf(-sum.z)) * sum.w;
      *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(g.out_lo) + (size_t)m * g.ld_out_lo + (n >> 1)) = pack_bf16x2(y0, y1);
      continue;
    }
    if (g.bias) { const float4 b = *reinterpret_cast<const float4*>(g.bias + n); sum.x += b.x; sum.y += b.y; sum.z += b.z; sum.w += b.w; }
    if (g.add) { const float4 q = addv[j][t]; sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w; }
    if (g.act != ACT_NONE) { sum.x = apply_act_rt(sum.x, g.act); sum.y = apply_act_rt(sum.y, g.act); sum.z = apply_act_rt(sum.z, g.act); sum.w = apply_act_rt(sum.w, g.act); }
    if (g.add2) {
      const float4 q = *reinterpret_cast<const float4*>(g.add2 + (size_t)(g.add2_rows ? g.add2_rows[m] : m) * g.ld_add2 + n);
      sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
    }
    if (g.out_f32) store4<float>(g.out_f32 + (size_t)m * g.ld_out_f32 + n, sum.x, sum.y, sum.z, sum.w);
    if (g.out_lo) store4<bf16_t>(reinterpret_cast<bf16_t*>(g.out_lo) + (size_t)m * g.ld_out_lo + n, sum.x, sum.y, sum.z, sum.w);
  }
}

// split-K across workgroups: enough splits that (N / 16) x splits workgroups cover about half the chip, K slices stay MFMA-sized
static int skinny_splits(const GemmArgs& g, int rows16) {
  // measured on MI355X (Whisper-large-v3 decode, SenseVoice single window): the hand-over costs more than the extra workgroups
  // gain (3.84 -> 4.27 ms / token), so the split is opt-in (ASR_SKINNY_SPLITK=1) and kept for shapes where N / 16 is tiny
  // ... except for 33..64 rows (Qwen3-ASR decode, batch 64: 2.68 -> 2.45 ms / token), where every workgroup also re-reads 4 row tiles
  // of activations: there the split is on by default for the narrow outputs (o_proj / down_proj, N / 16 = 64 workgroups otherwise)
  const int on = genv().skinny_splitk;
  const int granules = g.N / 16;
  // ... and for a long K behind few column granules (Whisper fc2: 80 workgroups x 164 KB of weights + 328 KB of activation rows each): a
  // CU streams ~40 GB/s, so 80 CUs alone take 13 us; four K slices put every CU to work (19.2 -> ~10 us with the 16-byte hand-over)
  const bool long_k = g.K >= 3072 && granules <= 128;
  if (on == 0 || (on < 0 && rows16 <= 32 && !long_k) || !g.sk_ws || !g.sk_cnt || g.a_rms_eps > 0.0f) return 1;
  int best = 1;
  for (int s : {2, 3, 4, 5, 8}) {
    if (g.K % (SK_WAVES * 32 * s) != 0) continue;
    if ((size_t)s * rows16 * g.N * 4 > g.sk_ws_bytes) continue;
    if (granules * best >= (long_k ? 256 : 160)) break;
    best = s;
  }
  return best;
}

template <int MT>
void launch_skinny(const GemmArgs& g, hipStream_t s) {
  ASR_REQUIRE(!(g.ln_x && g.a_rms_eps > 0.0f), "gemm(skinny): ln_x and a_rms_eps are exclusive");
  const size_t lds = (size_t)SK_WAVES * MT * 1024 + (g.ln_x ? (size_t)MT * 16 * g.K * 2 : 0) + (g.a_rms_eps > 0.0f ? (size_t)SK_WAVES * MT * 64 : 0);
  ASR_REQUIRE(lds <= 160 * 1024, "gemm(skinny): LayerNorm prologue needs %zu bytes of LDS", lds);
  static size_t attr = 0;
  if (lds > attr) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_skinny<MT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    if constexpr (MT <= 2)
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_skinny<MT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    attr = 160 * 1024;
  }
  GemmArgs gg = g;
  gg.sk_splits = skinny_splits(g, MT * 16);
  // outputs wider than the chip at three or four row tiles (Qwen3-ASR gate|up at 64 sequences: 384 granules, 128 KB of activation rows per granule from L2): a CU that
  // holds two workgroups pulls 2 x (rows + weights) through its fill path; two granules per workgroup read the rows once -- 12.8 -> ~9 us per launch (ASR_SKINNY_NT=1: off)
  if constexpr (MT >= 3 && MT <= 4) {
    const bool nt2 = genv().skinny_nt >= 2;
    if (nt2 && !g.ln_x && gg.sk_splits == 1 && g.N % 32 == 0 && g.N / 16 > 256) {
      const size_t lds2 = (size_t)SK_WAVES * MT * 2 * 1024 + (g.a_rms_eps > 0.0f ? (size_t)SK_WAVES * MT * 64 : 0);
      static PerDeviceOnce once2;
      if (once2.first()) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_skinny<MT, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_skinny<MT, false, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_skinny<MT, false, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
      }
      const dim3 grid2(g.N / 32, 1);
      if (g.W4) {
        ASR_REQUIRE(g.w_scale4 && g.K % 32 == 0, "gemm(skinny): MXFP4 weights need their block scales and K a multiple of 32");
        note_kernel("skinny_w4");
        hipLaunchKernelGGL((gemm_bf16_skinny<MT, false, 2, 2>), grid2, dim3(64 * SK_WAVES), lds2, s, gg);
      } else if (g.W8) {
        ASR_REQUIRE(g.w_scale && g.ldw8 % 8 == 0, "gemm(skinny): byte weights need their scales and 8-byte aligned rows");
        note_kernel("skinny_w8");
        hipLaunchKernelGGL((gemm_bf16_skinny<MT, false, 1, 2>), grid2, dim3(64 * SK_WAVES), lds2, s, gg);
      } else {
        note_kernel("skinny");
        hipLaunchKernelGGL((gemm_bf16_skinny<MT, false, 0, 2>), grid2, dim3(64 * SK_WAVES), lds2, s, gg);
      }
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if constexpr (MT <= 2) {
    if (g.ln_x) {
      note_kernel("skinny_ln");
      hipLaunchKernelGGL((gemm_bf16_skinny<MT, true>), dim3(g.N / 16, gg.sk_splits), dim3(64 * SK_WAVES), lds, s, gg);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if constexpr (MT <= 4) {
    if (g.W4) {
      ASR_REQUIRE(g.w_scale4 && !g.ln_x && g.K % 32 == 0, "gemm(skinny): MXFP4 weights need their block scales, K a multiple of 32 and no LayerNorm prologue");
      note_kernel("skinny_w4");
      hipLaunchKernelGGL((gemm_bf16_skinny<MT, false, 2>), dim3(g.N / 16, gg.sk_splits), dim3(64 * SK_WAVES), lds, s, gg);
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (g.W8) {
      ASR_REQUIRE(g.w_scale && !g.ln_x && g.ldw8 % 8 == 0, "gemm(skinny): byte weights need their scales, 8-byte aligned rows and no LayerNorm prologue");
      note_kernel("skinny_w8");
      hipLaunchKernelGGL((gemm_bf16_skinny<MT, false, 1>), dim3(g.N / 16, gg.sk_splits), dim3(64 * SK_WAVES), lds, s, gg);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  note_kernel("skinny");
  hipLaunchKernelGGL((gemm_bf16_skinny<MT, false>), dim3(g.N / 16, gg.sk_splits), dim3(64 * SK_WAVES), lds, s, gg);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------ f32 (verification mode)
constexpr int BK32 = 16;
constexpr int LDF = BK32 + 1;    // padded LDS row (floats): conflict-free fragment reads

template <bool SWAP>
__global__ __launch_bounds__(256, 2) void gemm_f32_128x128x16(const GemmArgs g) {
  __shared__ float As[BM * LDF];
  __shared__ float Ws[128 * LDF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / 128;
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
  // split-K (gridDim.y slices, pass 1 of launch_gemm_f32): this workgroup's K range; its raw partial tile goes to slice blockIdx.y of the workspace
  const int k_len = g.K / (int)gridDim.y, k_first = (int)blockIdx.y * k_len;
  for (int k0 = k_first; k0 < k_first + k_len; k0 += BK32) {
    const float4 a0 = *reinterpret_cast<const float4*>(A + (size_t)(tile_m * BM + lrow) * g.lda + k0 + lcol);
    const float4 a1 = *reinterpret_cast<const float4*>(A + (size_t)(tile_m * BM + lrow + 64) * g.lda + k0 + lcol);
    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)(tile_n * 128 + lrow) * g.ldw + k0 + lcol);
    const float4 w1 = *reinterpret_cast<const float4*>(W + (size_t)(tile_n * 128 + lrow + 64) * g.ldw + k0 + lcol);
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
      for (int j = 0; j < 4; ++j) wf[j] = Ws[(wn * 64 + frag_col(j, frow)) * LDF + kk * 4 + fgrp];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  const int m_wave = tile_m * BM + wm * 64, n_wave = tile_n * 128 + wn * 64;
  if constexpr (SWAP) {
    if (gridDim.y > 1) { GemmArgs gs = g; gs.out_f32 = g.out_f32 + (size_t)blockIdx.y * g.M * g.N; epilogue_rows<float, -1, -1, 4>(gs, acc, m_wave, n_wave, lane); }
    else epilogue_rows<float, -1, -1, 4>(g, acc, m_wave, n_wave, lane);
  } else epilogue_transposed<float, 4>(g, acc, m_wave, n_wave, lane);
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
  ASR_REQUIRE(g.N % 128 == 0, "gemm: N=%d must be a multiple of 128", g.N);
  ASR_REQUIRE(g.K % kstep == 0, "gemm: K=%d must be a multiple of %d", g.K, kstep);
  ASR_REQUIRE((g.lda * elt) % 16 == 0 && (g.ldw * elt) % 16 == 0, "gemm: leading dims must be 16-byte multiples");
  if (g.add) ASR_REQUIRE(g.ld_add % 4 == 0, "gemm: ld_add must be a multiple of 4");
  ASR_REQUIRE(g.N % 32 == 0, "gemm: N must be a multiple of 32");
  if (g.add2) ASR_REQUIRE(g.ld_add2 % 4 == 0, "gemm: ld_add2 must be a multiple of 4");
  if (g.out_f32) ASR_REQUIRE(g.ld_out_f32 % 4 == 0, "gemm: ld_out_f32 must be a multiple of 4");
  if (g.out_lo) ASR_REQUIRE(g.ld_out_lo % 8 == 0 && g.lo_group % 8 == 0, "gemm: ld_out_lo / lo_group must be multiples of 8");
  if (g.act == ACT_SWIGLU)
    ASR_REQUIRE(g.out_lo && !(g.add || g.add2 || g.out_f32 || g.amax_val || g.bias || g.lo_group || g.out_t), "gemm: SwiGLU stores to out_lo only");
  if (g.out_t) {
    ASR_REQUIRE(!(g.add || g.add2 || g.out_f32 || g.out_lo || g.amax_val || g.act != ACT_NONE),
                "gemm: the transposed store excludes row-major epilogue terms");
    ASR_REQUIRE(g.ld_out_t % 4 == 0, "gemm: ld_out_t must be a multiple of 4");
  }
}

// second launch of a split-K GEMM: out = epilogue(sum over splits, in split order); 4 columns per thread
template <typename LoT>      // element type behind out_lo: bf16 in bf16 sessions, float in verification mode
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g, const float* __restrict__ ws, int splits) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = g.N >> 2;
  if (idx >= (size_t)g.M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx - (size_t)m * n4) * 4;
  // every partial of this thread is requested before the first add (the summation order stays split order): as a load-add loop the compiler chained the
  // L2 round trips, 5.5 us for 8 splits of a single window's FFN-2
  float4 v = *reinterpret_cast<const float4*>(ws + (size_t)m * g.N + n);
  for (int sp0 = 1; sp0 < splits; sp0 += 8) {
    float4 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const float4*>(ws + ((size_t)min(sp0 + j, splits - 1) * g.M + m) * g.N + n);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (sp0 + j < splits) { v.x += q[j].x; v.y += q[j].y; v.z += q[j].z; v.w += q[j].w; }
  }
  if (g.bias) { const float4 b = *reinterpret_cast<const float4*>(g.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (g.add) { const float4 q = *reinterpret_cast<const float4*>(g.add + (size_t)m * g.ld_add + n); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
  if (g.act != ACT_NONE) { v.x = apply_act_rt(v.x, g.act); v.y = apply_act_rt(v.y, g.act); v.z = apply_act_rt(v.z, g.act); v.w = apply_act_rt(v.w, g.act); }
  if (g.add2) { const float4 q = *reinterpret_cast<const float4*>(g.add2 + (size_t)m * g.ld_add2 + n); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
  if (g.out_f32) store4<float>(g.out_f32 + (size_t)m * g.ld_out_f32 + n, v.x, v.y, v.z, v.w);
  if (g.out_lo) store4<LoT>(reinterpret_cast<LoT*>(g.out_lo) + (size_t)m * g.ld_out_lo + n, v.x, v.y, v.z, v.w);
  if (g.rms_out) {                               // N == 1024: this workgroup holds exactly row m; sum(x^2) over its 256 threads in a fixed order
    __shared__ float rs[4];
    float ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) rs[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float r = rsqrtf(((rs[0] + rs[1]) + (rs[2] + rs[3])) / (float)g.N + g.rms_eps);
    store4<LoT>(reinterpret_cast<LoT*>(g.rms_out) + (size_t)m * g.ld_rms_out + n, v.x * r, v.y * r, v.z * r, v.w * r);
  }
  if (g.st_out) {                                // (sum, sum of squares) of the bf16-rounded values per 32-column group = 8 consecutive threads
    const uint32_t p0 = pack_bf16x2(v.x, v.y), p1 = pack_bf16x2(v.z, v.w);
    const float a = __uint_as_float(p0 << 16), b = __uint_as_float(p0 & 0xffff0000u), c = __uint_as_float(p1 << 16), d = __uint_as_float(p1 & 0xffff0000u);
    float s1 = (a + b) + (c + d), s2 = fmaf(a, a, fmaf(b, b, fmaf(c, c, d * d)));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((threadIdx.x & 7) == 0) g.st_out[(size_t)m * (g.N >> 5) + (n >> 5)] = make_float2(s1, s2);
  }
}

// split-K for the tiled kernel: a grid of at most a few dozen tiles walking a long K is one latency-bound loop per workgroup
// (M = 144, N = 512, K = 2048: 16 workgroups x 32 K-steps = 30 us); S-way split => S x the workgroups, 1 / S the steps
static int tiled_splits(const GemmArgs& g) {
  const bool off = !genv().splitk;
  if (off || !g.sk_ws || g.out_t || g.amax_val || g.lo_group || g.add2_rows || g.ln_colsum || g.m_dev || g.act == ACT_SWIGLU || g.N % 32) return 1;
  const int tiles = ((g.M + BM - 1) / BM) * (g.N / 64);
  if (tiles > 128 || g.K < 1024) return 1;
  int best = 1;
  for (int sp : {2, 4, 8}) {
    if (g.K % (sp * BK16) != 0 || g.K / sp < 256) break;
    if ((size_t)sp * g.M * g.N * 4 > g.sk_ws_bytes) break;
    best = sp;
    if (tiles * sp >= 128) break;
  }
  return best;
}

template <int BN_, int STAGES, int ACT, int EPI, bool SWAP>
void launch_pipe_inst(const GemmArgs& g, hipStream_t s) {
  constexpr int lds = STAGES * (BM * 128 + BN_ * 128);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pipe<BN_, STAGES, ACT, EPI, SWAP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int grid = ((g.M + BM - 1) / BM) * (g.N / BN_);
  GemmArgs gg = g;
  if ((size_t)g.N * g.K * 2 > ((size_t)3 << 20) && g.M > 2 * BM) gg.group_m = 8;     // weights beyond an XCD's L2: group the row tiles
  note_kernel(g.k_splits > 1 ? "pipe_splitk" : "pipe");
  hipLaunchKernelGGL((gemm_bf16_pipe<BN_, STAGES, ACT, EPI, SWAP>), dim3(grid, g.k_splits > 1 ? g.k_splits : 1), dim3(256), lds, s, gg);
  HIP_CHECK(hipGetLastError());
}

template <int ACT, int EPI>
void launch_big_inst(const GemmArgs& g, hipStream_t s) {
  constexpr int lds = 2 * 2 * BIG * 128;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_big<ACT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int grid = ((g.M + BIG - 1) / BIG) * (g.N / BIG);
  note_kernel("big");
  hipLaunchKernelGGL((gemm_bf16_big<ACT, EPI>), dim3(grid), dim3(512), lds, s, g);
  HIP_CHECK(hipGetLastError());
}

// the 256 x 256 tiling pays when it fills whole rounds of the chip (one workgroup per CU) with little padding
bool big_fits(const GemmArgs& g) {
  const bool off = !genv().big;
  if (off || g.out_t || g.amax_val || g.lo_group || g.add2_rows || g.ln_colsum || g.st_out || g.m_dev || g.N % BIG || g.K % BK16 || g.M < 2048 || g.K < 512) return false;
  const int tiles_m = (g.M + BIG - 1) / BIG, tiles = tiles_m * (g.N / BIG);
  const int rounds = (tiles + 255) / 256;
  return (double)tiles / (rounds * 256.0) >= 0.85 && (double)g.M / (tiles_m * BIG) >= 0.9;
}

// the ping-pong 256 x 256 kernel (gemm_pp.hip): one workgroup per CU, so it pays when its tiles fill whole rounds of the chip
bool pp_fits(const GemmArgs& g) {
  if (!genv().pp || !gemm_pp_supported(g) || g.M < 1024 || g.K < 256) return false;
  const int tiles_m = (g.M + 255) / 256, tiles = tiles_m * (g.N / 256);
  const int rounds = (tiles + 255) / 256;
  return (double)tiles / (rounds * 256.0) >= 0.8 && (double)g.M / (tiles_m * 256) >= 0.9;
}

bool launch_big(const GemmArgs& g, hipStream_t s) {
  const int epi = (g.add ? E_ADD : 0) | (g.add2 ? E_ADD2 : 0) | (g.out_f32 ? E_F32 : 0) | (g.out_lo ? E_LO : 0) | (g.bias ? E_BIAS : 0);
#define ASR_BIG_CASE(ACT_, EPI_) \
  if (g.act == (ACT_) && epi == (EPI_)) { launch_big_inst<ACT_, EPI_>(g, s); return true; }
  ASR_BIG_CASE(ACT_NONE, E_BIAS | E_LO)                      // q|k projections
  ASR_BIG_CASE(ACT_GELU_ERF, E_BIAS | E_LO)                  // Whisper fc1
  ASR_BIG_CASE(ACT_GELU_TANH, E_BIAS | E_LO)                 // Qwen3-ASR encoder fc1 / conv stem
  ASR_BIG_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32)             // out-proj / fc2 + residual
  ASR_BIG_CASE(ACT_NONE, E_F32)                              // Qwen3 decoder q|k|v
  ASR_BIG_CASE(ACT_SWIGLU, E_LO)                             // Qwen3 decoder gate|up
  ASR_BIG_CASE(ACT_NONE, E_ADD | E_F32)                      // Qwen3 decoder o_proj / down_proj
  ASR_BIG_CASE(ACT_NONE, E_BIAS | E_F32)                     // logits
  ASR_BIG_CASE(ACT_RELU, E_BIAS | E_F32)
#undef ASR_BIG_CASE
  return false;
}

// Specialised epilogues used on the hot path; anything else runs the generic (runtime-checked) instantiation.
template <int BN_, int STAGES>
void launch_pipe(const GemmArgs& g, hipStream_t s) {
  if (g.out_t) { launch_pipe_inst<BN_, STAGES, ACT_NONE, 0, false>(g, s); return; }
  const int epi = (g.add ? E_ADD : 0) | (g.add2 ? E_ADD2 : 0) | (g.out_f32 ? E_F32 : 0) | (g.out_lo ? E_LO : 0) |
                  (g.amax_val ? E_AMAX : 0) | (g.bias ? E_BIAS : 0) | (g.st_out ? E_ST : 0);
#define ASR_GEMM_CASE(ACT_, EPI_) \
  if (g.act == (ACT_) && epi == (EPI_)) { launch_pipe_inst<BN_, STAGES, ACT_, EPI_, true>(g, s); return; }
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_LO)                    // q|k projection, cross-KV, plain projections
  ASR_GEMM_CASE(ACT_RELU, E_BIAS | E_LO)                    // SANM FFN-1
  ASR_GEMM_CASE(ACT_GELU_ERF, E_BIAS | E_LO)                // Whisper fc1
  ASR_GEMM_CASE(ACT_GELU_TANH, E_BIAS | E_LO)
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32)           // SANM out-proj: + FSMN memory + residual
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_F32)                    // SANM out-proj of the first block (no residual)
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32)           // FFN-2 / out-proj with bias + residual
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32 | E_LO)
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_F32 | E_LO)
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32 | E_LO)
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32 | E_LO | E_ST)
  ASR_GEMM_CASE(ACT_NONE, E_ADD | E_F32 | E_LO | E_ST)
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32 | E_LO | E_ST)
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_AMAX)                  // CTC / LM head arg-max
  ASR_GEMM_CASE(ACT_NONE, E_BIAS | E_F32)                   // LM head logits
  ASR_GEMM_CASE(ACT_NONE, E_F32)
  ASR_GEMM_CASE(ACT_SWIGLU, E_LO)
  ASR_GEMM_CASE(ACT_RELU, E_BIAS | E_F32)                   // Paraformer decoder FFN-1 (f32 out feeds the inner LayerNorm)
  ASR_GEMM_CASE(ACT_GELU_ERF, E_BIAS | E_ADD2 | E_F32)      // Whisper conv2: gelu(conv) + positions
  ASR_GEMM_CASE(ACT_GELU_TANH, E_BIAS | E_ADD2 | E_F32)
#undef ASR_GEMM_CASE
  ASR_REQUIRE(!g.st_out, "gemm: no statistics-producing instance for this epilogue");
  launch_pipe_inst<BN_, STAGES, -1, -1, true>(g, s);
}

bool g_t144_generic = false;

template <int STAGES, int ACT, int EPI>
void launch_t144_inst(const GemmArgs& g, hipStream_t s) {
  constexpr int ring = STAGES * T_STAGE;
  constexpr int lds = (ring > T_RED ? ring : T_RED) + 3 * TM * 8;      // + LayerNorm statistics
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_t144<STAGES, ACT, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  const int grid = ((g.M + TM - 1) / TM) * (g.N / TN);
  note_kernel("t144");
  hipLaunchKernelGGL((gemm_bf16_t144<STAGES, ACT, EPI>), dim3(grid), dim3(512), lds, s, g);
  HIP_CHECK(hipGetLastError());
}

template <int STAGES>
bool launch_t144(const GemmArgs& g, hipStream_t s) {
  const int epi = (g.add ? E_ADD : 0) | (g.add2 ? E_ADD2 : 0) | (g.out_f32 ? E_F32 : 0) | (g.out_lo ? E_LO : 0) | (g.bias ? E_BIAS : 0) |
                  (g.ln_colsum ? E_LN : 0) | (g.st_out ? E_ST : 0);
#define ASR_T144_CASE(ACT_, EPI_) \
  if (g.act == (ACT_) && epi == (EPI_)) { launch_t144_inst<STAGES, ACT_, EPI_>(g, s); return true; }
  ASR_T144_CASE(ACT_NONE, E_BIAS | E_LO)
  ASR_T144_CASE(ACT_NONE, E_F32)                             // decoder q|k|v (f32 for the per-head RMSNorm / RoPE)
  ASR_T144_CASE(ACT_SWIGLU, E_LO)                            // decoder gate|up with the SwiGLU epilogue
  ASR_T144_CASE(ACT_GELU_TANH, E_BIAS | E_LO)
  ASR_T144_CASE(ACT_GELU_ERF, E_BIAS | E_LO)                 // Whisper fc1
  ASR_T144_CASE(ACT_RELU, E_BIAS | E_LO)
  ASR_T144_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32)
  ASR_T144_CASE(ACT_NONE, E_ADD | E_F32)
  ASR_T144_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32)
  ASR_T144_CASE(ACT_RELU, E_BIAS | E_LO | E_LN)              // FFN-1 on the raw residual rows, LayerNorm inside
  ASR_T144_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32 | E_LO)     // residual stream written in f32 and as the next bf16 operand
  ASR_T144_CASE(ACT_NONE, E_ADD | E_F32 | E_LO)
  ASR_T144_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32 | E_LO)
  ASR_T144_CASE(ACT_NONE, E_ADD | E_ADD2 | E_F32 | E_LO | E_ST)
  ASR_T144_CASE(ACT_NONE, E_ADD | E_F32 | E_LO | E_ST)
  ASR_T144_CASE(ACT_NONE, E_BIAS | E_ADD | E_F32 | E_LO | E_ST)
#undef ASR_T144_CASE
  if (g.ln_colsum || g.st_out) return false;
  if (g_t144_generic) { launch_t144_inst<STAGES, -1, -1>(g, s); return true; }   // runtime-checked epilogue (op hooks / tests)
  return false;
}

// 144 x 256 tiles: the LayerNorm-folded FFN-1 instance, when the wide tiles still give every CU two full rounds' worth of work
bool launch_t144w(const GemmArgs& g, hipStream_t s) {
  const bool on = genv().t144w;
  if (!on || !g.ln_colsum || !g.ln_stats_in || g.st_out || g.N % TW || g.act != ACT_RELU) return false;
  const int epi = (g.add ? E_ADD : 0) | (g.add2 ? E_ADD2 : 0) | (g.out_f32 ? E_F32 : 0) | (g.out_lo ? E_LO : 0) | (g.bias ? E_BIAS : 0) | E_LN;
  if (epi != (E_BIAS | E_LO | E_LN)) return false;
  const int tiles = ((g.M + TM - 1) / TM) * (g.N / TW);
  if (tiles < 448 || (tiles % 256 > 0 && tiles % 256 < 192)) return false;     // whole rounds of one workgroup per CU (or nearly)
  constexpr int lds = (TW_STAGES * TW_STAGE > TW_RED ? TW_STAGES * TW_STAGE : TW_RED) + TM * 8;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_t144w<ACT_RELU, E_BIAS | E_LO | E_LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  note_kernel("t144w");
  hipLaunchKernelGGL((gemm_bf16_t144w<ACT_RELU, E_BIAS | E_LO | E_LN>), dim3(tiles), dim3(512), lds, s, g);
  HIP_CHECK(hipGetLastError());
  return true;
}

// 288 x 256 tiles: the same instance when pairs of windows x 256 columns make whole rounds
bool launch_t288w(const GemmArgs& g, hipStream_t s) {
  const bool on = genv().t288w;
  if (!on || !g.ln_colsum || !g.ln_stats_in || g.st_out || g.N % TW || g.act != ACT_RELU) return false;
  const int epi = (g.add ? E_ADD : 0) | (g.add2 ? E_ADD2 : 0) | (g.out_f32 ? E_F32 : 0) | (g.out_lo ? E_LO : 0) | (g.bias ? E_BIAS : 0) | E_LN;
  if (epi != (E_BIAS | E_LO | E_LN)) return false;
  const int tiles = ((g.M + TM2 - 1) / TM2) * (g.N / TW);
  if (tiles < 200 || (tiles % 256 != 0 && tiles % 256 < 200)) return false;      // whole rounds of one workgroup per CU (or nearly)
  constexpr int lds = 2 * T2_STAGE + TM2 * 8;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_t288w<ACT_RELU, E_BIAS | E_LO | E_LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  note_kernel("t288w");
  hipLaunchKernelGGL((gemm_bf16_t288w<ACT_RELU, E_BIAS | E_LO | E_LN>), dim3(tiles), dim3(512), lds, s, g);
  HIP_CHECK(hipGetLastError());
  return true;
}

// ... and for the vocabulary head with the arg-max epilogue (CTC / decoder vocab GEMM over a whole batch: thousands of tiles)
bool launch_t288w_amax(const GemmArgs& g, hipStream_t s) {
  const bool on = genv().t288w;
  if (!on || !g.amax_val || !g.bias || g.out_f32 || g.out_lo || g.out_t || g.add || g.add2 || g.ln_colsum || g.st_out || g.lo_group || g.act != ACT_NONE ||
      g.N % TW || g.K % BK16 || g.M < 8 * TM2)
    return false;
  const int tiles = ((g.M + TM2 - 1) / TM2) * (g.N / TW);
  if (tiles < 1024) return false;
  constexpr int lds = 2 * T2_STAGE + TM2 * 8;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_t288w<ACT_NONE, E_BIAS | E_AMAX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  note_kernel("t288w_amax");
  hipLaunchKernelGGL((gemm_bf16_t288w<ACT_NONE, E_BIAS | E_AMAX>), dim3(tiles), dim3(512), lds, s, g);
  HIP_CHECK(hipGetLastError());
  return true;
}

// 144-row tiles: usable when the epilogue is one the kernel implements and the row count is close to a multiple of 144
bool t144_geom_ok(const GemmArgs& g) {
  if (g.out_t || g.amax_val || g.lo_group || g.add2_rows || g.N % TN || g.K % BK16 || g.M < 128) return false;
  const int tiles_m = (g.M + TM - 1) / TM;
  return (double)tiles_m * TM <= 1.06 * g.M;                              // padding waste (one 137-row window: 144 / 137)
}
int t144_stages(const GemmArgs& g) { return ((g.M + TM - 1) / TM) * (g.N / TN) <= 256 ? 4 : 2; }
// ... and pay off when the tile count fills the chip more evenly than the 128-row tiling does
bool t144_fits(const GemmArgs& g, int* stages) {
  if (!t144_geom_ok(g) || g.M < 8 * TM) return false;
  const int tiles = ((g.M + TM - 1) / TM) * (g.N / TN);
  const int old_tiles = ((g.M + BM - 1) / BM) * (g.N / 64);
  const double old_cost = (double)((old_tiles + 511) / 512) * BM * 64;   // rounds x tile area (two co-resident workgroups per CU)
  *stages = t144_stages(g);
  if (tiles <= 256) return (double)TM * TN * 0.75 < old_cost;            // one round, one workgroup per CU
  return (double)((tiles + 511) / 512) * TM * TN * 0.85 < old_cost;
}

int g_gemm_variant = -1;   // -1 = heuristic; 5 / 6 = 144-row tiles with a 4- / 2-stage ring

}  // namespace

void gemm_set_variant(int v) { g_gemm_variant = v; }

static bool t144_enabled() { return genv().t144; }

bool gemm_ln_fusable(const GemmArgs& g) {
  return g_gemm_variant < 0 && t144_enabled() && g.M > 64 && g.ln_dim > 0 && g.K == (g.ln_dim + 63) / 64 * 64 && t144_geom_ok(g);
}

// mirrors the routing of launch_gemm_bf16 (conservatively: "false" only costs the caller a stand-alone RMSNorm launch)
bool gemm_reduce_can_norm(const GemmArgs& g) {
  if (g.N != 1024 || g_gemm_variant >= 0 || g.ln_x || g.a_rms_eps != 0.0f || g.ln_colsum || g.amax_val || g.out_t || g.lo_group) return false;
  const bool needs_skinny = !g.sk_ws || g.W8 != nullptr || g.W4 != nullptr;
  if (g.M <= 64 && (needs_skinny || g.M <= genv().skinny_max_plain) && g.K % (32 * SK_WAVES) == 0) return false;
  if (genv().skinny144 && g.M <= 144 && g.sk_ws && !g.st_out && g.K % (32 * SK_WAVES) == 0 && (g.lda * 2) % 16 == 0) return false;
  int st = 0;
  if (pp_fits(g) || big_fits(g) || (t144_enabled() && t144_fits(g, &st))) return false;
  return tiled_splits(g) > 1;
}

void launch_gemm_bf16(const GemmArgs& g, hipStream_t s) {
  if (g.rms_out) ASR_REQUIRE(gemm_reduce_can_norm(g), "gemm: rms_out is written by the split-K reduce launch only (check gemm_reduce_can_norm first; M = %d, N = %d, K = %d)", g.M, g.N, g.K);
  if (g.ln_x) ASR_REQUIRE(g.M <= 32 && !g.A, "gemm: the fused LayerNorm prologue exists on the skinny path for M <= 32 only");
  // 33..64 rows against a vocabulary-sized N: the 128 x 128 tiles re-read the activations 8 x less often than 16-column granules do
  // (lm_head 64 x 151936 x 1024: 76 us vs 240 us)
  const int tall_min = genv().tall_min;
  const bool tall = g.M > tall_min && g.N >= 16384 && !g.ln_x && g.a_rms_eps == 0.0f && g.act != ACT_SWIGLU;
  const int skinny_max_plain = genv().skinny_max_plain;   // rows up to which PLAIN GEMMs (no prologue) stream weights; above, the tiled split-K pass shares the activation rows across 64 columns (Whisper B = 64: 4.62 -> 3.89 ms per token)
  const bool needs_skinny = g.ln_x || g.a_rms_eps != 0.0f || !g.sk_ws || g.W8 != nullptr || g.W4 != nullptr;
  if (g.M <= 64 && (needs_skinny || g.M <= skinny_max_plain || g.N % 128 != 0) && !tall && !g.out_t && !g.amax_val && g.lo_group == 0 && g.K % (32 * SK_WAVES) == 0 && g_gemm_variant < 0) {
    ASR_REQUIRE((g.A || g.ln_x) && g.W && g.N % 16 == 0, "gemm(skinny): bad operands");
    ASR_REQUIRE(g.ln_x || (g.lda * 2) % 16 == 0, "gemm(skinny): lda must be a 16-byte multiple");
    if (g.ln_x) ASR_REQUIRE(g.K % 4 == 0 && g.K <= 1280 && g.ld_ln_x % 4 == 0, "gemm(skinny): fused LayerNorm needs K <= 1280, float4-aligned rows");
    if (g.M <= 16) launch_skinny<1>(g, s);
    else if (g.M <= 32) launch_skinny<2>(g, s);
    else launch_skinny<4>(g, s);
    return;
  }
  // one 8 s window (<= 144 rows): a tiled GEMM would put 4..16 workgroups on the chip; stream the weights with the skinny kernel instead
  const bool skinny144 = genv().skinny144;
  if (skinny144 && g.M <= 144 && g.sk_ws && !g.ln_x && !g.ln_colsum && !g.st_out && !g.out_t && !g.amax_val && g.lo_group == 0 && g.N % 16 == 0 &&
      g.K % (32 * SK_WAVES) == 0 && (g.lda * 2) % 16 == 0 && g_gemm_variant < 0) {
    launch_skinny<9>(g, s);
    return;
  }
  ASR_REQUIRE(g.a_rms_eps == 0.0f && !g.ln_x, "gemm: the RMSNorm / LayerNorm prologues exist on the weight-streaming (skinny) path only (M = %d, N = %d, K = %d)", g.M, g.N, g.K);
  check_args(g, BK16, 2);
  int v = g_gemm_variant;
  if (v < 0) {
    int st = 0;
    if (g.ln_colsum) {
      ASR_REQUIRE(gemm_ln_fusable(g), "gemm: the fused LayerNorm needs the 144-row-tile kernel (check gemm_ln_fusable first)");
      if (launch_t288w(g, s)) return;
      if (launch_t144w(g, s)) return;
      ASR_REQUIRE(t144_stages(g) == 4 ? launch_t144<4>(g, s) : launch_t144<2>(g, s), "gemm: no LayerNorm-fused instance for this epilogue");
      return;
    }
    if (g.amax_val && genv().amax_pp && pp_fits(g) && launch_gemm_pp(g, s)) { note_kernel("pp_amax"); return; }
    if (launch_t288w_amax(g, s)) return;
    if (!g.amax_val && pp_fits(g) && launch_gemm_pp(g, s)) { note_kernel("pp"); return; }
    if (big_fits(g) && launch_big(g, s)) return;
    if (t144_enabled() && t144_fits(g, &st) && (st == 4 ? launch_t144<4>(g, s) : launch_t144<2>(g, s))) return;
    if (const int sp = tiled_splits(g); sp > 1) {
      ASR_REQUIRE(!g.rms_out || g.N == 1024, "gemm: rms_out needs N == 1024 (one row per reduce workgroup)");
      GemmArgs p = g;                                    // pass 1: raw f32 partials, no epilogue terms
      p.rms_out = nullptr;
      p.bias = nullptr; p.add = nullptr; p.add2 = nullptr; p.act = ACT_NONE; p.out_lo = nullptr;
      p.out_f32 = g.sk_ws; p.ld_out_f32 = g.N; p.k_splits = sp; p.st_out = nullptr;
      const bool deep_sk = genv().deep;
      if (deep_sk && g.K / sp >= 256) launch_pipe<64, 3>(p, s); else launch_pipe<64, 2>(p, s);   // lone workgroups per CU: a third stage covers the DMA latency
      const size_t n = (size_t)g.M * (g.N / 4);
      hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, (const float*)g.sk_ws, sp);
      HIP_CHECK(hipGetLastError());
      return;
    }
    v = tall ? 2 : 4;
    // token-compacted rows (device-side M far below the launch bound): a handful of live workgroups, each alone on its CU, walking a
    // long K -- a third ring stage covers the DMA round trip the missing co-resident workgroup would have covered
    const bool deep = genv().deep;
    if (deep && g.m_dev && g.K >= 1024 && !tall && !g.amax_val) v = 3;
    if (deep && !tall && !g.amax_val && ((g.M + BM - 1) / BM) * (g.N / 64) <= 256 && g.K >= 256) v = 3;   // one round, at most one workgroup per CU
  }
  if (v >= 8 && v <= 8 + 128) {                // 8: persistent ping-pong kernel; 9..135: its tuning instances; 136: one workgroup per tile
    ASR_REQUIRE(launch_gemm_pp(g, s, v == 136 ? -1 : v - 8), "gemm: variant 8 (ping-pong 256 x 256 tiles) has no instance for this shape / epilogue");
    note_kernel("pp");
    return;
  }
  if (v == 7) {
    ASR_REQUIRE(g.N % BIG == 0 && launch_big(g, s), "gemm: variant 7 (256 x 256 tiles) has no instance for this shape / epilogue");
    return;
  }
  if (v == 5 || v == 6) {
    ASR_REQUIRE(!(g.out_t || g.amax_val || g.lo_group || g.add2_rows) && g.N % TN == 0, "gemm: variant %d (144-row tiles) does not support this epilogue", v);
    g_t144_generic = true;
    const bool ok = v == 5 ? launch_t144<4>(g, s) : launch_t144<2>(g, s);
    g_t144_generic = false;
    ASR_REQUIRE(ok, "gemm: variant %d has no instance for this epilogue", v);
    return;
  }
  if (g.amax_val && (v == 3 || v == 4)) v = 2;     // arg-max partials are per 64-column slab (BN = 128)
  switch (v) {
    case 1: launch_pipe<128, 3>(g, s); break;
    case 2: launch_pipe<128, 2>(g, s); break;
    case 3: launch_pipe<64, 3>(g, s); break;
    case 4: launch_pipe<64, 2>(g, s); break;
    default: ASR_THROW(ASR_ERR_INVALID, "gemm: unknown variant %d", v);
  }
}

// verification mode, small grids (a single window: 8 .. 32 tiles walking K = 512 .. 2048 in 16-column steps, one barrier pair each): K split
// across workgroups, partial tiles summed in split order by splitk_reduce_kernel, which also runs the epilogue. Only for callers that hand
// in a workspace (sk_ws); the summation order then depends on the split count, i.e. on the batch geometry -- within the mode's 1e-3 bar.
static int f32_splits(const GemmArgs& g) {
  if (!g.sk_ws || g.out_t || g.amax_val || g.lo_group || g.add2_rows || g.ln_colsum || g.m_dev || g.act == ACT_SWIGLU || g.ln_x || g.a_rms_eps != 0.0f || g.st_out) return 1;
  const int tiles = ((g.M + BM - 1) / BM) * (g.N / 128);
  if (tiles > 64 || g.K < 512) return 1;
  int best = 1;
  for (int sp : {2, 4, 8, 16}) {
    if (g.K % (sp * BK32) != 0 || g.K / sp < 64) break;
    if ((size_t)sp * g.M * g.N * 4 > g.sk_ws_bytes) break;
    best = sp;
    if (tiles * sp >= 160) break;
  }
  return best;
}

void launch_gemm_f32(const GemmArgs& g, hipStream_t s) {
  check_args(g, BK32, 4);
  const int grid = ((g.M + BM - 1) / BM) * (g.N / 128);
  if (const int sp = f32_splits(g); sp > 1) {
    GemmArgs p = g;                                    // pass 1: raw f32 partials, no epilogue terms
    p.bias = nullptr; p.add = nullptr; p.add2 = nullptr; p.act = ACT_NONE; p.out_lo = nullptr;
    p.out_f32 = g.sk_ws; p.ld_out_f32 = g.N;
    hipLaunchKernelGGL(gemm_f32_128x128x16<true>, dim3(grid, sp), dim3(256), 0, s, p);
    const size_t n = (size_t)g.M * (g.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, (const float*)g.sk_ws, sp);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (g.out_t) hipLaunchKernelGGL(gemm_f32_128x128x16<false>, dim3(grid), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_f32_128x128x16<true>, dim3(grid), dim3(256), 0, s, g);
  HIP_CHECK(hipGetLastError());
}

void launch_argmax_reduce(const float* val, const int32_t* idx, int M, int n_slabs, int32_t* ids, hipStream_t s) {
  hipLaunchKernelGGL(argmax_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, s, val, idx, M, n_slabs, ids);
  HIP_CHECK(hipGetLastError());
}
