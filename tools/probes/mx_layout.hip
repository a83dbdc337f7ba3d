// GPU probe: operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (cbsz = blgp = 0) -- checked against a
// host dot product of the decoded bytes. Hypothesis: lane l holds 32 consecutive k (bytes 32 (l >> 4) .. + 31) of row / column (l & 15); the scale
// operand's selected byte is an e8m0 factor for that lane's 32-block; D[m = 4 (l >> 4) + r][n = l & 15] as for the bf16 forms.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb) {        // A [16][128], B [16][128] (rows = n), C [16][16]
  const int l = threadIdx.x;
  i32x8 a = *reinterpret_cast<const i32x8*>(A + (l & 15) * 128 + (l >> 4) * 32);
  i32x8 b = *reinterpret_cast<const i32x8*>(B + (l & 15) * 128 + (l >> 4) * 32);
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  if (e == 15 && m == 7) x = NAN;
  return s ? -x : x;
}
int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  uint32_t x = 1234;
  auto rnd = [&] { x = x * 1664525u + 1013904223u; uint8_t v = (uint8_t)(x >> 13); if ((v & 0x7f) == 0x7f) v ^= 1; return v; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int test = 0; test < 2; ++test) {
    const int sa = test == 0 ? 0x7f7f7f7f : 0x7f7f7f80, sb = 0x7f7f7f7f;        // test 1: A scale byte 0 = 2^1
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double worst = 0, worst_t = 0, worst_abs = 0; 
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      double ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3(A[m * 128 + kk]) * e4m3(B[n * 128 + kk]);
      double reft = 0; for (int kk = 0; kk < 128; ++kk) reft += (double)e4m3(A[n * 128 + kk]) * e4m3(B[m * 128 + kk]);
      const double sc = test == 0 ? 1.0 : 2.0;
      double mag = 0; for (int kk = 0; kk < 128; ++kk) mag += fabs((double)e4m3(A[m * 128 + kk]) * e4m3(B[n * 128 + kk]));
      worst = fmax(worst, fabs(C[m * 16 + n] - sc * ref) / (1e-6 + fabs(ref)));
      worst_abs = fmax(worst_abs, fabs(C[m * 16 + n] - sc * ref) / (sc * mag));
      if (test == 0 && m < 2 && n < 3) printf("  D[%d][%d] = %.6g  ref %.6g  sum|terms| %.6g\n", m, n, C[m * 16 + n], ref, mag);
      worst_t = fmax(worst_t, fabs(C[m * 16 + n] - sc * reft) / (1e-6 + fabs(reft)));
    }
    printf("test %d (A scale %s): max rel err vs D[m][n] = sum_k A[m][k] B[n][k]: %.3g ; vs the transposed reading: %.3g ; max |err| / sum |terms| %.3g\n", test, test ? "2^1 in byte 0" : "1", worst, worst_t, worst_abs);
  }
  return 0;
}
