// Cost of a dependent kernel launch inside a hipGraph for kernels that do (almost) nothing:
// no arguments, a 256-byte argument struct, and a 256-byte struct with one global round trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { long v[32]; };
__global__ void k_empty() {}
__global__ void k_args(Big b, int* out) { if (b.v[31] == 12345 && threadIdx.x == 0) out[0] = 1; }
__global__ void k_load(Big b, int* out) { if (threadIdx.x == 0) out[1] = out[0] + (int)b.v[3]; }
template <typename F> static float run(F launch, hipStream_t s, int n) {
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  for (int i = 0; i < n; ++i) launch();
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best * 1e3f / n;
}
extern "C" int launch_floor_run() {
  hipStream_t s; hipStreamCreate(&s);
  int* out; hipMalloc(&out, 64); hipMemset(out, 0, 64);
  Big b{}; const int n = 256;
  for (int blocks : {1, 256, 512}) {
    for (int threads : {64, 512}) {
      float t0 = run([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, s); }, s, n);
      float t1 = run([&] { hipLaunchKernelGGL(k_args, dim3(blocks), dim3(threads), 0, s, b, out); }, s, n);
      float t2 = run([&] { hipLaunchKernelGGL(k_load, dim3(blocks), dim3(threads), 0, s, b, out); }, s, n);
      float t3 = run([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 65536, s); }, s, n);
      printf("blocks %3d x %3d threads: empty %.2f us, 256-B args %.2f us, args+load/store %.2f us, empty+64KB LDS %.2f us\n",
             blocks, threads, t0, t1, t2, t3);
    }
  }
  return 0;
}
