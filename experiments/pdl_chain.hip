// Does overlapping a dependent kernel's start with its predecessor's end pay on this part?
//
// A chain of N stages; stage i reads the vector stage i-1 wrote and streams its own weights.
//   mode A  one stream, plain loads / stores: every stage waits for the queue barrier.
//   mode B  stages alternate between two streams, so stage i+1 is dispatched while stage i
//           runs (it follows stage i-1 in its own queue); it requests its first weights, then
//           one lane polls stage i's arrival counter; the vector is handed over with
//           write-through stores and L1-bypassing loads.
// Both are captured into a hipGraph and replayed; the result vectors must agree.
//
//   hipcc --offload-arch=gfx950 -O3 -o pdl_chain experiments/pdl_chain.hip && ./pdl_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef unsigned int u32;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int kWGs = 256, kThreads = 512, kVec = 256;

struct StageArgs {
  const u32x4* w;        // this stage's weights: [kWGs][loads][kThreads] 16-byte pieces
  int loads;             // 16-byte loads per thread
  const u32* in;         // [kVec]
  u32* out;              // [kVec]
  const u32* prev_ctr;   // arrival counter of the previous stage (mode B), or null
  u32* my_ctr;
  u32* err;              // set when a bounded spin gives up
  int flagged;           // mode B
  int index;             // stage number
};

__global__ __launch_bounds__(kThreads) void stage_kernel(StageArgs a) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  // first weights go out before anything else
  u32x4 acc = {0u, 0u, 0u, 0u};
  const u32x4* wp = a.w + ((size_t)wg * a.loads) * kThreads + tid;
  u32x4 buf[8];
  const int first = a.loads < 8 ? a.loads : 8;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < first) buf[j] = __builtin_nontemporal_load(wp + (size_t)j * kThreads);
  __shared__ u32 s_in[kVec];
  __shared__ u32 s_red[kThreads / 64];
  if (a.flagged) {
    if (a.prev_ctr && tid == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(a.prev_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)kWGs) {
        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (wall_clock64() - t0 > 2000000ull) {  // 20 ms at 100 MHz: give up, never hang
          __hip_atomic_store(a.err, (u32)(a.index * 1000 + wg + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (tid < kVec) s_in[tid] = __hip_atomic_load(a.in + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (tid < kVec) s_in[tid] = a.in[tid];
  }
  for (int j0 = 0; j0 < a.loads; j0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j0 + j < a.loads) {
        const u32x4 v = buf[j];
        if (j0 + 8 + j < a.loads) buf[j] = __builtin_nontemporal_load(wp + (size_t)(j0 + 8 + j) * kThreads);
        acc ^= v;
      }
    }
  }
  __syncthreads();
  u32 x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  x = x * 2654435761u + s_in[(tid + wg) % kVec];
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = x;
  __syncthreads();
  if (tid == 0) {
    u32 s = 0;
    for (int i = 0; i < kThreads / 64; ++i) s += s_red[i];
    if (a.flagged) {
      __hip_atomic_store(a.out + wg, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(a.my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      a.out[wg] = s;
    }
  }
}

int main(int argc, char** argv) {
  const int N = 64;
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t fork_ev, join_ev, t0, t1;
  CHECK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  const char* names[4] = {"one stream, graph", "fork/join in ONE graph", "two streams, eager",
                          "two streams, one linear graph each"};
  for (int kb_per_wg : {0, 8, 32, 64, 128, 256}) {
    const int loads = kb_per_wg * 1024 / 16 / kThreads;
    const size_t w_elems = (size_t)kWGs * (loads > 0 ? loads : 1) * kThreads;
    std::vector<u32x4*> w(N);
    for (int i = 0; i < N; ++i) {
      CHECK(hipMalloc(&w[i], w_elems * 16));
      CHECK(hipMemset(w[i], 0x11 + i, w_elems * 16));
    }
    u32 *vecs, *ctrs, *err;
    CHECK(hipMalloc(&vecs, (size_t)(N + 1) * kVec * 4));
    CHECK(hipMalloc(&ctrs, 4096));
    err = ctrs + 512;
    std::vector<u32> h_in(kVec);
    for (int i = 0; i < kVec; ++i) h_in[i] = i * 7 + 1;
    u32 result[4][kVec];
    float us[4];
    const double mb = (double)kWGs * loads * kThreads * 16 / 1e6;
    for (int mode = 0; mode < 4; ++mode) {
      CHECK(hipMemset(vecs, 0, (size_t)(N + 1) * kVec * 4));
      CHECK(hipMemcpy(vecs, h_in.data(), kVec * 4, hipMemcpyHostToDevice));
      CHECK(hipMemset(ctrs, 0, 4096));
      CHECK(hipDeviceSynchronize());
      auto stage = [&](int i, hipStream_t s) {
        StageArgs a{w[i], loads, vecs + (size_t)i * kVec, vecs + (size_t)(i + 1) * kVec,
                    (mode >= 1 && i > 0) ? ctrs + (i - 1) : nullptr, ctrs + i, err, mode >= 1, i};
        hipLaunchKernelGGL(stage_kernel, dim3(kWGs), dim3(kThreads), 0, s, a);
      };
      hipGraph_t g[2] = {nullptr, nullptr};
      hipGraphExec_t ex[2] = {nullptr, nullptr};
      if (mode == 0 || mode == 1) {
        CHECK(hipStreamBeginCapture(sa, hipStreamCaptureModeRelaxed));
        if (mode == 1) {
          CHECK(hipMemsetAsync(ctrs, 0, 4096, sa));
          CHECK(hipEventRecord(fork_ev, sa));
          CHECK(hipStreamWaitEvent(sb, fork_ev, 0));
        }
        for (int i = 0; i < N; ++i) stage(i, (mode == 1 && (i & 1)) ? sb : sa);
        if (mode == 1) {
          CHECK(hipEventRecord(join_ev, sb));
          CHECK(hipStreamWaitEvent(sa, join_ev, 0));
        }
        CHECK(hipStreamEndCapture(sa, &g[0]));
        CHECK(hipGraphInstantiate(&ex[0], g[0], nullptr, nullptr, 0));
      } else if (mode == 3) {
        for (int c = 0; c < 2; ++c) {
          hipStream_t s = c ? sb : sa;
          CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
          for (int i = c; i < N; i += 2) stage(i, s);
          CHECK(hipStreamEndCapture(s, &g[c]));
          CHECK(hipGraphInstantiate(&ex[c], g[c], nullptr, nullptr, 0));
        }
      }
      float best = 1e30f;
      for (int r = 0; r < 7; ++r) {
        CHECK(hipEventRecord(t0, sa));
        if (mode <= 1) {
          CHECK(hipGraphLaunch(ex[0], sa));
        } else {
          CHECK(hipMemsetAsync(ctrs, 0, 4096, sa));
          CHECK(hipEventRecord(fork_ev, sa));
          CHECK(hipStreamWaitEvent(sb, fork_ev, 0));
          if (mode == 2) {
            for (int i = 0; i < N; ++i) stage(i, (i & 1) ? sb : sa);
          } else {
            CHECK(hipGraphLaunch(ex[0], sa));
            CHECK(hipGraphLaunch(ex[1], sb));
          }
          CHECK(hipEventRecord(join_ev, sb));
          CHECK(hipStreamWaitEvent(sa, join_ev, 0));
        }
        CHECK(hipEventRecord(t1, sa));
        CHECK(hipEventSynchronize(t1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        if (ms < best) best = ms;
      }
      us[mode] = best * 1e3f / N;
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(result[mode], vecs + (size_t)N * kVec, kVec * 4, hipMemcpyDeviceToHost));
      u32 h_err = 0;
      CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
      bool same = true;
      for (int i = 0; i < kVec; ++i) same &= result[0][i] == result[mode][i];
      printf("%4d KB/WG (%6.1f MB/stage) %-36s %7.2f us/stage (%5.2f TB/s)%s%s\n", kb_per_wg, mb, names[mode],
             us[mode], mb / us[mode], same ? "" : "  RESULTS DIFFER", h_err ? "  A SPIN GAVE UP" : "");
      if (h_err) printf("      first give-up: stage*1000+wg+1 = %u\n", h_err);
      fflush(stdout);
      for (int c = 0; c < 2; ++c)
        if (ex[c]) {
          CHECK(hipGraphExecDestroy(ex[c]));
          CHECK(hipGraphDestroy(g[c]));
        }
    }
    for (int i = 0; i < N; ++i) CHECK(hipFree(w[i]));
    CHECK(hipFree(vecs));
    CHECK(hipFree(ctrs));
  }
  return 0;
}
