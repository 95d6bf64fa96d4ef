// Operand map and conversions of the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4,
// e4m3 operands, unit scales), checked with exact small integers, and the rounding of
// v_cvt_pk_fp8_f32 against a host restatement of OCP e4m3 (round to nearest even, saturating at
// 448).  Build: hipcc --offload-arch=gfx950 -O2 -o build/exp/mfma_f8_probe experiments/mfma_f8_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// lane l = (i = l & 15, q = l >> 4), byte e of its 32: A[i][32 q + e], B[32 q + e][i]
__global__ void mm(const uint8_t* A, const uint8_t* B, float* D) {
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  i32x8 a, b;
  uint8_t* ab = reinterpret_cast<uint8_t*>(&a);
  uint8_t* bb = reinterpret_cast<uint8_t*>(&b);
  for (int e = 0; e < 32; ++e) {
    ab[e] = A[i * 128 + 32 * q + e];
    bb[e] = B[(32 * q + e) * 16 + i];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];   // row 4q + r, column i
}

__global__ void cvt(const float* x, uint8_t* o, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * t + 1 < n) {
    const unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * t], x[2 * t + 1], 0u, false);
    o[2 * t] = r & 0xff;
    o[2 * t + 1] = (r >> 8) & 0xff;
  }
}

static float e4m3_to_f32(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = std::ldexp((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = std::ldexp(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
static uint8_t f32_to_e4m3(float x) {  // nearest even, saturating
  uint8_t best = 0;
  float bd = INFINITY;
  const float ax = std::fmin(std::fabs(x), 448.f);
  for (int v = 0; v < 127; ++v) {
    const float d = std::fabs(e4m3_to_f32((uint8_t)v) - ax);
    if (d < bd || (d == bd && (v & 1) == 0)) { bd = d; best = (uint8_t)v; }
  }
  return best | (std::signbit(x) ? 0x80 : 0);
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(128 * 16);
  srand(3);
  for (auto& v : A) v = f32_to_e4m3((float)(rand() % 9 - 4));
  for (auto& v : B) v = f32_to_e4m3((float)(rand() % 9 - 4));
  uint8_t *dA, *dB;
  float* dD;
  hipMalloc(&dA, A.size());
  hipMalloc(&dB, B.size());
  hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0.f;
      for (int k = 0; k < 128; ++k) s += e4m3_to_f32(A[i * 128 + k]) * e4m3_to_f32(B[k * 16 + j]);
      bad += s != D[i * 16 + j];
    }
  printf("16x16x128 e4m3 MFMA with unit scales, lane (i, q) byte e = k 32q+e: %d of 256 wrong\n", bad);

  const int n = 1 << 16;
  std::vector<float> x(n);
  for (int i = 0; i < n; ++i) {
    const float u = (float)rand() / RAND_MAX;
    x[i] = (i % 2 ? -1.f : 1.f) * std::exp(u * 16.f - 9.f);   // 1e-4 .. 1e3: subnormals to beyond 448
  }
  x[0] = 448.f; x[1] = 449.f; x[2] = 464.f; x[3] = 480.f; x[4] = 1e6f; x[5] = 0.f; x[6] = 0.0009765625f;
  float* dx;
  uint8_t* dox;
  hipMalloc(&dx, n * 4);
  hipMalloc(&dox, n);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dox, n);
  std::vector<uint8_t> o(n);
  hipMemcpy(o.data(), dox, n, hipMemcpyDeviceToHost);
  int diff = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    const uint8_t w = f32_to_e4m3(x[i]);
    if (w != o[i]) {
      ++diff;
      if (shown++ < 8) printf("  x=%g: device 0x%02x (%g), host 0x%02x (%g)\n", x[i], o[i], e4m3_to_f32(o[i]), w, e4m3_to_f32(w));
    }
  }
  printf("v_cvt_pk_fp8_f32 against nearest-even saturating e4m3: %d of %d differ\n", diff, n);
  return 0;
}
