// What one all-to-all exchange between the workgroups of a resident kernel costs: every one of
// 256 workgroups contributes 8 values of a 2048-vector and then needs the whole vector, N times
// in a row, each round depending on the one before (the activation vector between two
// weight-streaming phases of a decoder layer at one token row).
//
//   flat     every workgroup polls all 2048 tagged 8-byte words with agent-coherent (sc1) loads:
//            256 readers of the same 16 KB, served from the memory side
//   two-hop  one workgroup per XCD (the first to take a ticket there; the XCD is read from the
//            hardware register, not inferred from the workgroup index) polls the words as above
//            and re-publishes the vector as tagged words in a buffer of its XCD; the others poll
//            THAT buffer with sc0 loads, which bypass the CU's L1 but are served by the XCD's own
//            L2, where their leader's write-through stores land
//
// with the chip otherwise idle, and with every workgroup's other 8 waves streaming a large buffer
// (2 wave loads of 1 KiB in flight per wave: the full HBM rate).  Every wait is bounded; the
// vector's checksum is verified each round.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/exchange_hops experiments/exchange_hops.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned long long u64;

constexpr int kVec = 2048, kWgs = 256, kPer = kVec / kWgs;
constexpr u64 kTicks = 50000000ull;  // 0.5 s

struct Args {
  u64* flat;          // [2][kVec] tagged words, by round parity
  u64* xcd;           // [8][2][kVec] tagged words per XCD
  unsigned* ticket;   // [8] first workgroup of an XCD to arrive leads it
  unsigned* failed;
  unsigned* xcc_of;   // [kWgs] XCD of every workgroup (for the report)
  float* result;      // [kWgs] final value per workgroup
  const u32x4* stream;
  size_t stream16;    // 16-byte elements per workgroup region
  int rounds, two_hop, with_stream;
};

__device__ __forceinline__ u64 ld_sc1(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_sc0(const u64* p) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sc0(u64* p, u64 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u64 pack(float v, unsigned tag) { return ((u64)tag << 32) | __float_as_uint(v); }

__global__ __launch_bounds__(576) void exchange_kernel(Args a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  __shared__ int stop;
  __shared__ int leader_s;
  if (tid == 0) stop = 0;
  __syncthreads();
  if (wave < 8) {
    if (!a.with_stream) return;
    const u32x4* p = a.stream + (size_t)wg * a.stream16 + (size_t)wave * 64 + lane;
    const size_t n_chunks = a.stream16 / 512;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r0 = __builtin_nontemporal_load(p), r1 = __builtin_nontemporal_load(p + 512);
    for (size_t c = 2;; c += 2) {
      if (c + 2 > n_chunks) c = 2;   // wrap: keep streaming until the exchange rounds are done
      acc ^= r0; r0 = __builtin_nontemporal_load(p + c * 512);
      acc ^= r1; r1 = __builtin_nontemporal_load(p + (c + 1) * 512);
      if (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    if ((acc[0] ^ acc[1]) == 0x1234567u) a.failed[1] = 1;
    return;
  }
  // ---- the exchange wave
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID, 4 bits
  if (lane == 0) {
    a.xcc_of[wg] = xcc;
    leader_s = atomicAdd(a.ticket + (xcc & 7), 1u) == 0;
  }
  const bool leader = __shfl(leader_s, 0, 64) != 0;   // same wave wrote it: LDS round trip
  float mine = 1.0f;                                   // this workgroup's contribution
  bool ok = true;
  const u64 t0 = wall_clock64();
  for (int it = 0; it < a.rounds && ok; ++it) {
    const unsigned tag = (unsigned)it + 1u;
    u64* fl = a.flat + (size_t)(it & 1) * kVec;
    if (lane < kPer) st_sc1(fl + wg * kPer + lane, pack(mine, tag));
    float sum = 0.f;
    const bool poll_flat = !a.two_hop || leader;
    const u64* src = poll_flat ? fl : a.xcd + ((size_t)(xcc & 7) * 2 + (it & 1)) * kVec;
    u64 w[32];
    for (;;) {
      bool all = true;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const u64* q = src + i * 64 + lane;
        w[i] = (poll_flat || a.two_hop == 1) ? ld_sc1(q) : 0;
      }
      if (!poll_flat && a.two_hop >= 2) {
        // drop this CU's L1 copies, then read through the XCD's L2
        if (a.two_hop == 2) asm volatile("buffer_inv sc0" ::: "memory");
        else asm volatile("buffer_inv sc1" ::: "memory");
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const u64* q = src + i * 64 + lane;
          asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(w[i]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) all &= (unsigned)(w[i] >> 32) == tag;
      if (__all(all)) break;
      if (wall_clock64() - t0 > kTicks) { ok = false; break; }
    }
    if (a.two_hop && leader) {
      u64* dst = a.xcd + ((size_t)(xcc & 7) * 2 + (it & 1)) * kVec;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (a.two_hop == 1) st_sc1(dst + i * 64 + lane, w[i]);
        else st_sc0(dst + i * 64 + lane, w[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += __uint_as_float((unsigned)w[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    // every value of round `it` is 2^it (exact): sum = 2048 * 2^it; next contribution 2^(it+1)
    if (sum != 2048.0f * mine) ok = false;
    mine = sum * (1.0f / 1024.0f);
  }
  if (!ok && lane == 0) atomicAdd(a.failed, 1u);
  if (lane == 0) {
    a.result[wg] = mine;
    __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  Args a{};
  const int rounds = 64;   // values double every round: 2^64 stays exact in f32
  const size_t stream_bytes = (size_t)1 << 30;
  u32x4* sb; CK(hipMalloc(&sb, stream_bytes)); CK(hipMemset(sb, 1, stream_bytes));
  CK(hipMalloc(&a.flat, 2 * kVec * 8)); CK(hipMalloc(&a.xcd, 8 * 2 * kVec * 8));
  CK(hipMalloc(&a.ticket, 8 * 4)); CK(hipMalloc(&a.failed, 8)); CK(hipMalloc(&a.xcc_of, kWgs * 4));
  CK(hipMalloc(&a.result, kWgs * 4));
  a.stream = sb; a.stream16 = stream_bytes / 16 / kWgs / 512 * 512; a.rounds = rounds;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int with_stream = 0; with_stream < 2; ++with_stream) {
    for (int two_hop = 0; two_hop < 4; ++two_hop) {
      a.two_hop = two_hop; a.with_stream = with_stream;
      float best = 1e9f; unsigned failed = 0; float res[kWgs]; unsigned xcc_of[kWgs];
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(a.flat, 0, 2 * kVec * 8, st)); CK(hipMemsetAsync(a.xcd, 0, 8 * 2 * kVec * 8, st));
        CK(hipMemsetAsync(a.ticket, 0, 32, st)); CK(hipMemsetAsync(a.failed, 0, 8, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(exchange_kernel, dim3(kWgs), dim3(576), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        CK(hipMemcpy(&failed, a.failed, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(res, a.result, sizeof(res), hipMemcpyDeviceToHost));
        CK(hipMemcpy(xcc_of, a.xcc_of, sizeof(xcc_of), hipMemcpyDeviceToHost));
        if (failed) break;
      }
      if (!with_stream && !two_hop) {
        printf("XCC_ID of workgroups 0..15:");
        for (int i = 0; i < 16; ++i) printf(" %u", xcc_of[i]);
        printf("\n");
      }
      int bad = 0, rr = 0;
      const float want = ldexpf(1.0f, rounds);
      for (int i = 0; i < kWgs; ++i) { bad += res[i] != want; rr += xcc_of[i] == (unsigned)(i % 8); }
      printf("%s, %s: %.2f us per exchange (%d rounds)%s; workgroup i on XCD i %% 8: %d of %d\n",
             with_stream ? "beside a full-rate stream" : "idle chip", two_hop == 0 ? "flat (256 readers)" : two_hop == 1 ? "two-hop, XCD buffers read with sc1 loads"
             : two_hop == 2 ? "two-hop, XCD buffers through the L2 (buffer_inv sc0 + sc0 loads)"
                            : "two-hop, XCD buffers through the L2 (buffer_inv sc1 + sc0 loads)",
             best * 1e3f / rounds, rounds, (failed || bad) ? "  FAILED" : "", rr, kWgs);
    }
  }
  return 0;
}
