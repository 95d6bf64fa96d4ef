// A chain of dependent weight-streaming phases (the linear layers of a small decoder layer at one
// token row) run two ways: one kernel launch per phase inside a hipGraph, and ONE resident kernel
// whose workgroups meet at a flag barrier between phases and request the next phase's first
// weights before they wait.  Same arithmetic, same work split; the question is what a phase
// boundary costs each way.
//
// Cross-workgroup data (the activations, the barrier flags) only moves through agent-scope
// relaxed atomics (sc1 loads / stores: coherent across the XCDs' L2s), so no cache write-back
// or invalidate is needed.  Every wait is bounded.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kThreads = 512, kWaves = 8, kSeg = 8, kMaxPhases = 5, kMaxK = 8192;

struct Phase {
  const bf16_t* W;      // [layers][n_tiles][ksteps][64][8]
  size_t layer_elems;   // elements per layer
  int n_tiles, ksteps;
};
struct Args {
  Phase ph[kMaxPhases];
  int n_phases;
  float* xb[2];         // activations, f32, alternating by phase parity
  unsigned* flags;      // one per workgroup
  unsigned* failed;
  unsigned long long* xt[2];  // LL exchange: {f32 value, u32 tag} words, alternating by phase parity
  int prefetch_across;  // request the next phase's weights before the barrier
  int idle_phase;       // index of a phase in which only 40 workgroups have (latency-bound) work, or -1
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all flags >= epoch?  wave 0 polls, 4 flags per lane (grid <= 256) or 8 (<= 512)
__device__ bool barrier_wait(const Args& a, unsigned epoch, int* ok_s) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const unsigned G = gridDim.x;
    bool ok = true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      bool mine = true;
      for (unsigned i = tid; i < G; i += 64) mine &= ld_flag(a.flags + i) >= epoch;
      if (__all(mine)) break;
      if (wall_clock64() - t0 > 200000000ull || ld_flag(a.failed)) {  // 2 s at 100 MHz
        ok = false;
        break;
      }
    }
    if (tid == 0) {
      if (!ok) __hip_atomic_store(a.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_s = ok;
    }
  }
  __syncthreads();
  return *ok_s != 0;
}

__device__ __forceinline__ unsigned long long ll_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | __float_as_uint(v);
}

// Waits until word idx of buf carries `tag`, returns its value (bounded; *ok cleared on time-out).
__device__ __forceinline__ float ll_read(const unsigned long long* buf, int idx, unsigned tag, bool* ok) {
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    const unsigned long long w = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(w >> 32) == tag) return __uint_as_float((unsigned)w);
    if (wall_clock64() - t0 > 200000000ull) {
      *ok = false;
      return 0.f;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void chain_kernel(Args a, int g_lo, int g_hi) {
  constexpr bool PERSIST = MODE != 0;
  constexpr bool LL = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                       // [2][kMaxK + 8]: row 0 = x, row 1 = 0
  float* red = reinterpret_cast<float*>(smem + 2 * (kMaxK + 8) * 2);  // [2][waves][256]
  int* ok_s = reinterpret_cast<int*>(red + 2 * kWaves * 256);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LDX = kMaxK + 8;
  for (int c = tid * 8; c < LDX; c += kThreads * 8)
    *reinterpret_cast<bf16x8*>(xs + LDX + c) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  const int xc = lane & 15, xq = lane >> 4;
  const bf16_t* xrow = xs + (size_t)(xc < 1 ? 0 : 1) * LDX + 8 * xq;

  bf16x8 nxt[kSeg];
  auto load_seg = [&](int g, int tile, int seg) {
    const Phase& p = a.ph[g % a.n_phases];
    const bf16_t* w = p.W + (size_t)(g / a.n_phases) * p.layer_elems +
                      ((size_t)tile * p.ksteps + (size_t)seg * kSeg) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < kSeg; ++j)
      nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w + (size_t)j * 512));
  };
  auto first_load = [&](int g) {
    const Phase& p = a.ph[g % a.n_phases];
    const int n_seg = p.ksteps / kSeg;
    const int units = (g % a.n_phases) == a.idle_phase ? 0 : p.n_tiles;
    if ((int)blockIdx.x < units && wave < n_seg) load_seg(g, blockIdx.x, wave);
  };
  first_load(g_lo);
  int parity = 0;
  for (int g = g_lo; g < g_hi; ++g) {
    const Phase& p = a.ph[g % a.n_phases];
    const int K = p.ksteps * 32, n_seg = p.ksteps / kSeg;
    const bool idle = (g % a.n_phases) == a.idle_phase;
    if (MODE == 1 && g > g_lo) {
      if (!barrier_wait(a, (unsigned)(g - g_lo), ok_s)) return;
    }
    bool ll_ok = true;
    const unsigned long long* tin = a.xt[g & 1];
    unsigned long long* tout = a.xt[(g + 1) & 1];
    const unsigned tag_in = (unsigned)g, tag_out = (unsigned)(g + 1);
    const float* xin = a.xb[g & 1];
    float* xout = a.xb[(g + 1) & 1];
    if (idle) {
      // 40 workgroups walk a chain of 3 dependent round trips (what a small attention pass
      // is); everyone else only waits.  Output: x passes through unchanged, doubled.
      if (blockIdx.x < 40) {
        float v = 0.f;
        int idx = tid;
        for (int hop = 0; hop < 3; ++hop) {
          v = LL ? ll_read(tin, idx & 2047, tag_in, &ll_ok)
                 : __hip_atomic_load(xin + (idx & 2047), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          idx = (int)(v * 0.f) + tid + hop;  // dependent address
        }
        for (int c = blockIdx.x * kThreads + tid; c < 2048; c += 40 * kThreads) {
          if (LL) {
            const float xv = ll_read(tin, c, tag_in, &ll_ok);
            __hip_atomic_store(tout + c, ll_pack(2.f * xv + 0.f * v, tag_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __hip_atomic_store(xout + c, 2.f * __hip_atomic_load(xin + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 0.f * v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (PERSIST && g + 1 < g_hi && a.prefetch_across) first_load(g + 1);
    } else {
      if (LL) {
        for (int c = tid; c < K; c += kThreads) xs[c] = (bf16_t)ll_read(tin, c, tag_in, &ll_ok);
      } else {
        for (int c = tid; c < K; c += kThreads)
          xs[c] = (bf16_t)__hip_atomic_load(xin + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (LL && !__syncthreads_and(ll_ok)) {
        if (tid == 0) __hip_atomic_store(a.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __syncthreads();
      const int n_units = p.n_tiles;
      for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x, parity ^= 1) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int seg = wave; seg < n_seg; seg += kWaves) {
          bf16x8 cur[kSeg];
#pragma unroll
          for (int j = 0; j < kSeg; ++j) cur[j] = nxt[j];
          if (seg + kWaves < n_seg) load_seg(g, unit, seg + kWaves);
          else if (unit + (int)gridDim.x < n_units) load_seg(g, unit + gridDim.x, wave);
          else if (PERSIST && g + 1 < g_hi && a.prefetch_across) first_load(g + 1);
          const bf16_t* xp = xrow + (size_t)seg * kSeg * 32;
#pragma unroll
          for (int j = 0; j < kSeg; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j], *reinterpret_cast<const bf16x8*>(xp + j * 32), acc, 0, 0, 0);
        }
        float* rbuf = red + parity * kWaves * 256;
        *reinterpret_cast<f32x4*>(rbuf + wave * 256 + lane * 4) = acc;
        __syncthreads();
        const int nl = 4 * (tid >> 6) + (tid & 3), m = (tid >> 2) & 15;
        if (tid < 256 && m == 0) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < kWaves; ++w) s += rbuf[w * 256 + tid];
          if (LL) __hip_atomic_store(tout + unit * 16 + nl, ll_pack(s, tag_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_store(xout + unit * 16 + nl, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if ((int)blockIdx.x >= n_units && PERSIST && g + 1 < g_hi && a.prefetch_across) first_load(g + 1);
    }
    if (MODE == 1 && g + 1 < g_hi) {
      if (!a.prefetch_across) {
        // the comparison case: nothing in flight across the barrier
      }
      __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have landed
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(a.flags + blockIdx.x, (unsigned)(g - g_lo + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!a.prefetch_across) first_load(g + 1);  // requested only after arriving
    }
    if (LL && g + 1 < g_hi && !a.prefetch_across) first_load(g + 1);  // after publishing, before the wait
  }
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 16;
  const int per_cu = argc > 2 ? atoi(argv[2]) : 1;  // resident workgroups per CU
  hipStream_t st; CK(hipStreamCreate(&st));
  // Llama-3.2-1B: qkv 2048 -> 3072, [attention], o 2048 -> 2048, gate/up 2048 -> 16384, down 8192 -> 2048
  struct Shape { int N, K; } shapes[4] = {{3072, 2048}, {2048, 2048}, {16384, 2048}, {2048, 8192}};
  for (int with_idle = 0; with_idle < 2; ++with_idle) {
    Args a{};
    a.n_phases = with_idle ? 5 : 4;
    a.idle_phase = with_idle ? 1 : -1;
    size_t total = 0;
    std::vector<bf16_t*> bufs;
    for (int p = 0, s = 0; p < a.n_phases; ++p) {
      if (p == a.idle_phase) { a.ph[p] = Phase{nullptr, 0, 0, 2048 / 32}; continue; }
      const Shape sh = shapes[s++];
      const size_t elems = (size_t)sh.N * sh.K;
      bf16_t* w; CK(hipMalloc(&w, elems * layers * 2));
      // every weight = 2 / K: an all-equal x doubles per phase (exact in bf16 / f32)
      std::vector<bf16_t> h(elems);
      const bf16_t v = (bf16_t)(2.0f / sh.K);
      for (auto& e : h) e = v;
      for (int l = 0; l < layers; ++l) CK(hipMemcpy(w + (size_t)l * elems, h.data(), elems * 2, hipMemcpyHostToDevice));
      a.ph[p] = Phase{w, elems, sh.N / 16, sh.K / 32};
      total += elems * 2;
      bufs.push_back(w);
    }
    CK(hipMalloc(&a.xb[0], 16384 * 4)); CK(hipMalloc(&a.xb[1], 16384 * 4));
    CK(hipMalloc(&a.flags, 512 * 4)); CK(hipMalloc(&a.failed, 4));
    CK(hipMemset(a.failed, 0, 4));
    const int G = layers * a.n_phases;
    const size_t lds = 2 * (kMaxK + 8) * 2 + 2 * kWaves * 256 * 4 + 16 + (per_cu == 1 ? 64 * 1024 : 0);  // > 80 KB: one workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> ones(16384, 1.0f), back(16384);
    auto check = [&](const char* what) {
      CK(hipMemcpy(back.data(), a.xb[G & 1], 2048 * 4, hipMemcpyDeviceToHost));
      const float want = ldexpf(1.0f, G);
      int bad = 0;
      for (int i = 0; i < 2048; ++i) bad += back[i] != want;
      unsigned f; CK(hipMemcpy(&f, a.failed, 4, hipMemcpyDeviceToHost));
      if (bad || f) printf("  %s: %d wrong values (got %g, want %g), failed flag %u\n", what, bad, back[0], want, f);
      return bad == 0 && f == 0;
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (1) a launch per phase, in a graph
    {
      hipGraph_t gr; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      for (int g = 0; g < G; ++g) {
        const Phase& p = a.ph[g % a.n_phases];
        const int blocks = (g % a.n_phases) == a.idle_phase ? 40 : (p.n_tiles < 256 * per_cu ? p.n_tiles : 256 * per_cu);
        hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(kThreads), lds, st, a, g, g + 1);
      }
      CK(hipStreamEndCapture(st, &gr));
      CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        CK(hipMemcpyAsync(a.xb[0], ones.data(), 16384 * 4, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      check("graph");
      printf("[%d per CU] %s: %d layers, %.1f MB per layer: launch per phase %.2f us per layer (%.2f TB/s)\n",
             per_cu, with_idle ? "with a 40-workgroup latency phase" : "linear phases only", layers, total / 1e6,
             best * 1e3f / layers, total * layers / (best * 1e-3) / 1e12);
    }
    // (2) one resident kernel
    for (int across = 0; across < 2; ++across) {
      a.prefetch_across = across;
      float best = 1e9f;
      bool ok = true;
      for (int r = 0; r < 5 && ok; ++r) {
        CK(hipMemcpyAsync(a.xb[0], ones.data(), 16384 * 4, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(a.flags, 0, 512 * 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(chain_kernel<1>, dim3(256 * per_cu), dim3(kThreads), lds, st, a, 0, G);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        ok = check("resident");
      }
      printf("    resident kernel, weights requested %s the barrier: %.2f us per layer (%.2f TB/s)%s\n",
             across ? "before" : "after", best * 1e3f / layers, total * layers / (best * 1e-3) / 1e12, ok ? "" : "  FAILED");
      if (!ok) return 1;
    }
    // (3) one resident kernel, no barrier: activations travel as {value, tag} words and every
    // reader polls the words it needs
    {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipMalloc(&a.xt[0], 16384 * 8)); CK(hipMalloc(&a.xt[1], 16384 * 8));
      std::vector<unsigned long long> init(16384);
      for (auto& w : init) w = ((unsigned long long)0u << 32) | 0x3f800000u;   // tag 0, value 1.0
      for (int across = 0; across < 2; ++across) {
        a.prefetch_across = across;
        float best = 1e9f;
        bool ok = true;
        for (int r = 0; r < 5 && ok; ++r) {
          CK(hipMemcpyAsync(a.xt[0], init.data(), 16384 * 8, hipMemcpyHostToDevice, st));
          CK(hipMemsetAsync(a.xt[1], 0xff, 16384 * 8, st));
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(chain_kernel<2>, dim3(256 * per_cu), dim3(kThreads), lds, st, a, 0, G);
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
          std::vector<unsigned long long> got(2048);
          CK(hipMemcpy(got.data(), a.xt[G & 1], 2048 * 8, hipMemcpyDeviceToHost));
          unsigned f; CK(hipMemcpy(&f, a.failed, 4, hipMemcpyDeviceToHost));
          int bad = 0;
          const float want = ldexpf(1.0f, G);
          for (int i = 0; i < 2048; ++i) {
            float v; unsigned u = (unsigned)got[i]; memcpy(&v, &u, 4);
            bad += v != want || (unsigned)(got[i] >> 32) != (unsigned)G;
          }
          if (bad || f) { printf("  tagged words: %d wrong values, failed flag %u\n", bad, f); ok = false; }
        }
        printf("    resident kernel, tagged words (no barrier), next weights requested %s the wait: %.2f us per layer (%.2f TB/s)%s\n",
               across ? "before" : "after", best * 1e3f / layers, total * layers / (best * 1e-3) / 1e12, ok ? "" : "  FAILED");
        if (!ok) return 1;
      }
      CK(hipFree(a.xt[0])); CK(hipFree(a.xt[1]));
    }
    for (auto* w : bufs) CK(hipFree(w));
    CK(hipFree(a.xb[0])); CK(hipFree(a.xb[1])); CK(hipFree(a.flags)); CK(hipFree(a.failed));
  }
  return 0;
}
