// Does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N) shorten a
// dependent launch?  A chain of kernels in a hipGraph, each loading through a pointer it got as an
// argument (so the first global load waits for the argument): pointer arguments, and the same
// pointers inside a by-value struct (never preloaded).  Build twice, with and without the flag:
//   hipcc --offload-arch=gfx950 -O3 experiments/kernarg_preload.hip -o build/exp/kp_off
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 experiments/kernarg_preload.hip -o build/exp/kp_on
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { const float* in; float* out; int n; float scale; long pad[12]; };
__global__ void k_ptrs(const float* in, float* out, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * scale + 1.0f;
}
__global__ void k_struct(Args a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.out[i] = a.in[i] * a.scale + 1.0f;
}
template <typename F> static float run(F launch, hipStream_t s, int n) {
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best * 1e3f / n;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  const int n = 256 * 256;
  float *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
  const int launches = 512;
  for (int blocks : {8, 256}) {
    const int m = blocks * 256;
    float t0 = run([&](int i) { hipLaunchKernelGGL(k_ptrs, dim3(blocks), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, m, 0.5f); }, s, launches);
    float t1 = run([&](int i) { Args a{(i & 1) ? y : x, (i & 1) ? x : y, m, 0.5f, {}};
                                 hipLaunchKernelGGL(k_struct, dim3(blocks), dim3(256), 0, s, a); }, s, launches);
    printf("%3d blocks: pointer arguments %.3f us per dependent launch, the same in a struct %.3f us\n", blocks, t0, t1);
  }
  return 0;
}
