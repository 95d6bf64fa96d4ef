// Cost of a device-wide barrier inside one resident kernel (one workgroup per CU, or two),
// to compare with the cost of a dependent kernel launch inside a hipGraph.  Every spin is
// bounded: a workgroup that waits too long raises a flag and all workgroups leave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Sync { unsigned count; unsigned failed; };

__device__ inline bool grid_barrier(Sync* s, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_fetch_add(&s->count, 1u, __ATOMIC_RELEASE);  // agent scope by default in HIP
    unsigned spins = 0;
    while (__atomic_load_n(&s->count, __ATOMIC_ACQUIRE) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000u || __atomic_load_n(&s->failed, __ATOMIC_RELAXED)) {
        __atomic_store_n(&s->failed, 1u, __ATOMIC_RELAXED);
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  return ok;
}

// rounds barriers; with exchange != 0 every workgroup also publishes one float per thread and
// reads another workgroup's after the barrier (the data hop a fused layer would make).
__global__ void k_barriers(Sync* s, float* buf, float* out, int rounds, int exchange) {
  const unsigned G = gridDim.x;
  const unsigned T = blockDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (exchange) {
      float* mine = buf + (size_t)(r & 1) * G * T + (size_t)blockIdx.x * T;
      mine[threadIdx.x] = (float)(r + 1) + (float)blockIdx.x * 1e-3f;
    }
    if (!grid_barrier(s, (unsigned)(r + 1) * G)) break;
    if (exchange) {
      unsigned other = (blockIdx.x + 97u) % G;
      const float* theirs = buf + (size_t)(r & 1) * G * T + (size_t)other * T;
      acc += theirs[threadIdx.x] - (float)other * 1e-3f;
    }
  }
  if (exchange) out[(size_t)blockIdx.x * T + threadIdx.x] = acc;
}

extern "C" int grid_barrier_run() {
  hipStream_t st; hipStreamCreate(&st);
  Sync* s; hipMalloc(&s, sizeof(Sync));
  const int rounds = 1000;
  for (int blocks : {64, 256, 512}) {
    for (int threads : {256, 512}) {
      if (blocks == 512 && threads == 512) continue;  // keep every workgroup resident
      for (int exchange : {0, 1}) {
        float *buf, *out;
        hipMalloc(&buf, sizeof(float) * 2 * blocks * threads);
        hipMalloc(&out, sizeof(float) * blocks * threads);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9f;
        unsigned failed = 0;
        for (int rep = 0; rep < 3 && !failed; ++rep) {
          hipMemsetAsync(s, 0, sizeof(Sync), st);
          hipEventRecord(a, st);
          hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(threads), 0, st, s, buf, out, rounds, exchange);
          hipEventRecord(b, st);
          hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
          Sync h; hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost);
          failed = h.failed;
        }
        int bad = 0;
        if (exchange && !failed) {
          float* h = (float*)malloc(sizeof(float) * blocks * threads);
          hipMemcpy(h, out, sizeof(float) * blocks * threads, hipMemcpyDeviceToHost);
          const float want = 0.5f * rounds * (rounds + 1);
          for (int i = 0; i < blocks * threads; ++i) if (h[i] < want * 0.999f || h[i] > want * 1.001f) ++bad;
          free(h);
        }
        printf("blocks %3d x %3d threads, exchange %d: %.2f us per barrier%s, stale reads %d\n", blocks, threads,
               exchange, best * 1e3f / rounds, failed ? " (BAILED OUT)" : "", bad);
        hipFree(buf); hipFree(out);
        if (failed) return 1;
      }
    }
  }
  return 0;
}
