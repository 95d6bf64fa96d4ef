// Bytes in flight per CU against (a) the rate a weight stream reaches and (b) the latency of a
// small dependent load issued by ANOTHER wave of the same CU while the stream runs: the two
// numbers a resident multi-phase kernel is designed around (how deep the weight requests may
// run ahead before an exchange between workgroups starts to wait behind them).
//
// One workgroup per CU: 8 streaming waves read a private, contiguous region of a large buffer
// with DEPTH wave loads of 1 KiB in flight each (non-temporal, like the linear kernels'
// weight loads); a ninth wave issues one 8-byte write-through-coherent (sc1) load of a small
// buffer every ~1.5 us and times it with the wall clock (100 MHz).  Host: events around the
// launch give the stream's rate; the probe wave leaves its mean and maximum.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/inflight_latency experiments/inflight_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Probe { unsigned long long sum, max, n; };

template <int DEPTH>
__global__ __launch_bounds__(576) void stream_kernel(const u32x4* __restrict__ buf, size_t per_wg16,
                                                     const unsigned long long* probe_buf,
                                                     Probe* probes, unsigned* sink, unsigned* done) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int finished;
  if (tid == 0) finished = 0;
  __syncthreads();
  if (wave < 8) {
    // wave w walks chunks w, w + 8, ... of 64 lanes x 16 B inside the workgroup's region
    const u32x4* p = buf + (size_t)blockIdx.x * per_wg16 + (size_t)wave * 64 + lane;
    const size_t n_chunks = per_wg16 / (64 * 8);
    u32x4 r[DEPTH];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = __builtin_nontemporal_load(p + (size_t)d * 512);
    size_t c = DEPTH;
    for (; c + DEPTH <= n_chunks; c += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc ^= r[d];                                        // waits for the oldest load only
        r[d] = __builtin_nontemporal_load(p + (c + d) * 512);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= r[d];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;   // keeps the loads alive
    if (lane == 0) atomicAdd(&finished, 1);
  } else {
    unsigned long long sum = 0, mx = 0, n = 0;
    unsigned idx = blockIdx.x * 97u + 13u;
    while (__hip_atomic_load(&finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8) {
      const unsigned long long t0 = wall_clock64();
      const unsigned long long v =
          __hip_atomic_load(probe_buf + (idx & 32767u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the value has arrived
      const unsigned long long dt = wall_clock64() - t0;
      idx = idx * 1664525u + 1013904223u + (unsigned)v;
      sum += dt; n += 1; if (dt > mx) mx = dt;
      for (int i = 0; i < 12; ++i) __builtin_amdgcn_s_sleep(127);   // ~1.5 us between probes
    }
    if (lane == 0) probes[blockIdx.x] = Probe{sum, mx, n};
  }
  if (tid == 0) atomicAdd(done, 1u);
}

template <int DEPTH>
void run(const u32x4* buf, size_t bytes, int wgs, const unsigned long long* pb, Probe* probes,
         unsigned* sink, unsigned* done, hipStream_t st) {
  const size_t per_wg16 = bytes / 16 / wgs / 512 * 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  std::vector<Probe> h(wgs);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(576), 0, st, buf, per_wg16, pb, probes, sink, done);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) { best = ms; CK(hipMemcpy(h.data(), probes, sizeof(Probe) * wgs, hipMemcpyDeviceToHost)); }
  }
  double sum = 0, n = 0, mx = 0;
  for (auto& p : h) { sum += (double)p.sum; n += (double)p.n; if ((double)p.max > mx) mx = (double)p.max; }
  const double moved = (double)per_wg16 * 16 * wgs;
  printf("%3d KiB in flight per CU (%2d loads per wave): stream %.2f TB/s (%.1f us), probe load mean %.2f us, "
         "max %.2f us (%0.f probes)\n", 8 * DEPTH, DEPTH, moved / (best * 1e-3) / 1e12, best * 1e3,
         n ? sum / n / 100.0 : 0.0, mx / 100.0, n);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int wgs = prop.multiProcessorCount;
  const size_t bytes = (size_t)1 << 30;
  u32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
  unsigned long long* pb; CK(hipMalloc(&pb, 32768 * 8)); CK(hipMemset(pb, 0, 32768 * 8));
  Probe* probes; CK(hipMalloc(&probes, sizeof(Probe) * wgs));
  unsigned *sink, *done; CK(hipMalloc(&sink, 4)); CK(hipMalloc(&done, 4)); CK(hipMemset(done, 0, 4));
  printf("%d CUs, 1 GiB streamed per launch, one workgroup per CU\n", wgs);
  run<1>(buf, bytes, wgs, pb, probes, sink, done, st);
  run<2>(buf, bytes, wgs, pb, probes, sink, done, st);
  run<4>(buf, bytes, wgs, pb, probes, sink, done, st);
  run<8>(buf, bytes, wgs, pb, probes, sink, done, st);
  run<16>(buf, bytes, wgs, pb, probes, sink, done, st);
  run<32>(buf, bytes, wgs, pb, probes, sink, done, st);
  return 0;
}
