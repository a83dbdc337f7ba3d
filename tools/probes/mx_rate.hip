// GPU probe: issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit scales) against v_mfma_f32_16x16x32_bf16: cycles per instruction on one SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int KIND> __global__ void k(float* out, long long* cyc, int n, int fmt) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  i32x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x3a3a3a3a + i; }
  s16x8 ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = 0x3f80; hb[i] = 0x3f00; }
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
      else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp4 x fp4
    }
  }
  const long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* o; long long* c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 8);
  const int n = 2000;
  for (int kind = 0; kind < 3; ++kind) {
    for (int waves = 1; waves <= 2; ++waves) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, o, c, n, 0);
      else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, o, c, n, 0);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * waves), 0, 0, o, c, n, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long cyc; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
      const double flop = (kind == 0 ? 16384.0 : 65536.0) * 8 * n * 256 * 4 * waves;
      printf("%s, %d wave(s) per SIMD: %.1f us, %.0f TF/s, s_memtime %.1f ticks per MFMA per wave\n", kind == 0 ? "bf16 16x16x32 " : kind == 1 ? "e4m3 16x16x128" : "fp4  16x16x128", waves, ms * 1e3, flop / ms / 1e9, (double)cyc / (8.0 * n));
    }
  }
  return 0;
}
