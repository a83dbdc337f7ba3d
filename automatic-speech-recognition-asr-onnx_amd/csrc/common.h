// Shared device/host helpers for the gfx950 (CDNA4, wave64) ASR engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define ASR_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __host__ __forceinline__ bf16_t f32_to_bf16(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two f32 -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_v;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_v;
  const f32x2_v v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.0f) & 0xffffu); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum of `n` (sum, sum-of-squares) partials of one row; the loads are issued in batches of 8 slots so that their latencies overlap
__device__ __forceinline__ float2 sum_row_partials(const float2* sp, int n) {
  float sx = 0.0f, sy = 0.0f;
  int q = 0;
  for (; q + 8 <= n; q += 8) {
    float4 t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = *reinterpret_cast<const float4*>(sp + q + 2 * e);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sx += t[e].x; sy += t[e].y; sx += t[e].z; sy += t[e].w; }
  }
  for (; q < n; ++q) { const float2 t = sp[q]; sx += t.x; sy += t.y; }
  return make_float2(sx, sy);
}

// One-time per-DEVICE setup (hipFuncSetAttribute / hipMemcpyToSymbol act on the current device): first() is true the first time it is asked on
// the current device. A process that opens sessions on several GPUs thus sets up every large-LDS kernel on each of them.
struct PerDeviceOnce {
  bool done[32] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    d &= 31;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

// ---- host-side error plumbing (thread-local message, integer status) -------------------
enum {
  ASR_OK = 0,
  ASR_ERR_INVALID = 1,     // bad argument / shape / dtype
  ASR_ERR_HIP = 2,         // HIP runtime failure
  ASR_ERR_NOT_FOUND = 3,   // tensor missing from the arena manifest
  ASR_ERR_UNSUPPORTED = 4,
  ASR_ERR_NO_DEVICE = 5,
};

void asr_set_error(const std::string& msg);

struct AsrError {
  int code;
  std::string msg;
};

#define ASR_THROW(code_, ...)                                   \
  do {                                                          \
    char _b[512];                                               \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                      \
    throw AsrError{(code_), std::string(_b)};                   \
  } while (0)

#define HIP_CHECK(expr)                                                                         \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      ASR_THROW(ASR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define ASR_REQUIRE(cond, ...)                        \
  do {                                                \
    if (!(cond)) ASR_THROW(ASR_ERR_INVALID, __VA_ARGS__); \
  } while (0)
