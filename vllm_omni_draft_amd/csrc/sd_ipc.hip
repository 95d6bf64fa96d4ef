// One-shot collectives over peer-mapped device memory (xGMI between the GPUs of a node), for
// the small messages of a tensor-parallel verify pass: rows x hidden bf16 after each of the two
// row-parallel projections of a layer (6 x 8192 x 2 B = 96 KB at 70B), where a ring's 2(N-1)
// latency-bound steps cost more than every rank reading every peer's vector once.
//
// Every rank owns one exchange buffer (fine-grained device memory, exported with
// hipIpcGetMemHandle, mapped by its peers): a header of per-slice flags and two data
// regions used alternately.  A call cuts the message into slices of kSliceBytes, one workgroup
// each; workgroup w of rank r
//   1. copies its slice of the input into r's data region (parity of its call counter),
//   2. publishes flag[r][w] = counter (system-scope release),
//   3. waits until flag[q][w] >= counter for every peer q (system-scope acquire, bounded),
//   4. reads slice w from every rank's region, IN RANK ORDER (so all ranks compute the same
//      bits), and writes the sum (all-reduce) or the pieces side by side (all-gather).
// No workgroup waits for another workgroup of its own grid, so the grid need not be resident
// at once.  A region may be overwritten two calls later: by then every peer has published
// the counter of the call in between, which it does only after finishing this one.  The call
// counters live in device memory, so a captured graph replays correctly.  A wait that gives up
// (a peer that never arrives) raises the communicator's error counter instead of hanging.
#include <cstdlib>
#include <cstring>

#include "sd_common.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSliceBytes = 8192;
constexpr int kThreads = 256;
constexpr size_t kHeaderBytes = 16384;   // flags[kMaxSlices] + counters[kMaxSlices] (u32 each)
constexpr int kMaxSlices = 2048;         // 16 MiB per message at most
// How long a wait for a peer may last (100 MHz wall clock): 20 s by default.  Two ranks of a
// 70B model as two processes on one card needed more than 4 s at their first exchange (the
// card time-slices processes); a dead peer still ends the wait, and the step raises.
constexpr unsigned long long kSpinTicks = 2000000000ull;

struct IpcPeers {
  unsigned char* buf[kMaxRanks];  // buf[rank] = this rank's own buffer
  int rank, world;
  size_t region_bytes;            // bytes of one data region
  unsigned* errors;               // device counter (own memory)
  unsigned long long spin_ticks;  // how long a wait may last before it gives up
};

__device__ __forceinline__ unsigned* flags_of(unsigned char* b) { return reinterpret_cast<unsigned*>(b); }
__device__ __forceinline__ unsigned* counters_of(unsigned char* b) {
  return reinterpret_cast<unsigned*>(b) + kMaxSlices;
}
__device__ __forceinline__ unsigned char* region_of(unsigned char* b, size_t region_bytes, unsigned parity) {
  return b + kHeaderBytes + (size_t)parity * region_bytes;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// GATHER == false: buf (bf16 [count]) <- sum over ranks.  GATHER == true: recv <- pieces.
template <bool GATHER>
__global__ __launch_bounds__(kThreads) void ipc_exchange_kernel(IpcPeers p, const unsigned char* send,
                                                                unsigned char* out, size_t bytes) {
  __shared__ unsigned s_seq;
  const int tid = threadIdx.x;
  unsigned char* own = p.buf[p.rank];
  const int n_slices = (int)((bytes + kSliceBytes - 1) / kSliceBytes);
  // A workgroup walks its slices one after the other (messages of more than kMaxGrid slices:
  // prefill passes); every slice is an exchange of its own, so no workgroup of this grid waits
  // for another one of this grid.
  for (int w = blockIdx.x; w < n_slices; w += gridDim.x) {
  if (tid == 0) s_seq = counters_of(own)[w] + 1;
  __syncthreads();
  const unsigned seq = s_seq;
  const size_t lo = (size_t)w * kSliceBytes;
  const size_t n = min(kSliceBytes, bytes - lo);   // a multiple of 16
  // 1. publish the slice
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(send + lo);
    u32x4* dst = reinterpret_cast<u32x4*>(region_of(own, p.region_bytes, seq & 1) + lo);
    for (size_t i = tid; i < n / 16; i += kThreads) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  // 2. the flag goes up BEFORE any lane of this wave starts to wait (a store in one branch of a
  // divergent wave could be scheduled after the other branch's spin loop: both ranks would
  // then wait for flags that are never raised)
  if (tid == 0)
    __hip_atomic_store(flags_of(own) + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. one lane per peer waits
  if (tid < p.world && tid != p.rank) {
    const unsigned* f = flags_of(p.buf[tid]) + w;
    const unsigned long long t0 = wall_clock64();
    // (flag - seq) as a signed difference: correct across counter wrap-around
    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (wall_clock64() - t0 > p.spin_ticks) {
        atomicAdd(p.errors, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // every thread: nothing stale from before the wait
  // 4. combine in rank order
  if (GATHER) {
    for (int q = 0; q < p.world; ++q) {
      const u32x4* src = reinterpret_cast<const u32x4*>(region_of(p.buf[q], p.region_bytes, seq & 1) + lo);
      u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)q * bytes + lo);
      for (size_t i = tid; i < n / 16; i += kThreads) dst[i] = __builtin_nontemporal_load(src + i);
    }
  } else {
    for (size_t i = tid; i < n / 16; i += kThreads) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < p.world; ++q) {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(region_of(p.buf[q], p.region_bytes, seq & 1) + lo);
        const bf16x8 v = __builtin_nontemporal_load(src + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
      reinterpret_cast<bf16x8*>(out + lo)[i] = o;
    }
  }
  if (tid == 0) counters_of(own)[w] = seq;
  __syncthreads();   // the next slice rewrites s_seq
  }
}

// Workgroups per launch at most: a waiting grid must leave compute units free.  Where the
// ranks are processes on ONE card (rehearsals), a peer's next kernel may need a whole unit's
// LDS; a grid that holds a little LDS on every unit while it waits for that peer would never
// see it arrive.
constexpr int kMaxGrid = 96;

}  // namespace

struct sd_ipc_comm {
  IpcPeers peers;
  size_t max_bytes;
};

extern "C" size_t sd_ipc_buffer_bytes(size_t max_message_bytes) {
  const size_t region = (max_message_bytes + kSliceBytes - 1) / kSliceBytes * kSliceBytes;
  return kHeaderBytes + 2 * region;
}

extern "C" int sd_ipc_buffer_create(size_t bytes, void** ptr, void* handle_out) {
  SD_REQUIRE(ptr && handle_out && bytes >= kHeaderBytes, SD_ERR_INVALID,
             "sd_ipc_buffer_create: bad arguments (%zu bytes)", bytes);
  static_assert(sizeof(hipIpcMemHandle_t) == SD_IPC_HANDLE_BYTES, "handle size");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  SD_REQUIRE(e == hipSuccess && p, SD_ERR_HIP, "sd_ipc_buffer_create: allocation: %s",
             hipGetErrorString(e));
  e = hipMemset(p, 0, bytes);   // flags and counters start at 0 (synchronous on purpose)
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    sd_set_error("sd_ipc_buffer_create: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  *ptr = p;
  return SD_OK;
}

extern "C" int sd_ipc_buffer_open(const void* handle, void** ptr) {
  SD_REQUIRE(handle && ptr, SD_ERR_INVALID, "sd_ipc_buffer_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof h);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  SD_REQUIRE(e == hipSuccess && p, SD_ERR_HIP, "sd_ipc_buffer_open: %s", hipGetErrorString(e));
  *ptr = p;
  return SD_OK;
}

extern "C" int sd_ipc_buffer_close(void* ptr) {
  if (!ptr) return SD_OK;
  const hipError_t e = hipIpcCloseMemHandle(ptr);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_ipc_buffer_close: %s", hipGetErrorString(e));
  return SD_OK;
}

extern "C" int sd_ipc_buffer_destroy(void* ptr) {
  if (!ptr) return SD_OK;
  const hipError_t e = hipFree(ptr);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_ipc_buffer_destroy: %s", hipGetErrorString(e));
  return SD_OK;
}

extern "C" sd_ipc_comm* sd_ipc_comm_create(int rank, int world_size, void* const* bufs,
                                           size_t max_message_bytes) {
  if (!bufs || world_size < 1 || world_size > kMaxRanks || rank < 0 || rank >= world_size ||
      max_message_bytes < 16 || max_message_bytes > kMaxSlices * kSliceBytes) {
    sd_set_error("sd_ipc_comm_create: bad arguments (rank %d of %d, %zu bytes)", rank, world_size,
                 max_message_bytes);
    return nullptr;
  }
  for (int q = 0; q < world_size; ++q)
    if (!bufs[q]) {
      sd_set_error("sd_ipc_comm_create: buffer of rank %d is NULL", q);
      return nullptr;
    }
  sd_ipc_comm* c = new sd_ipc_comm();
  for (int q = 0; q < kMaxRanks; ++q) c->peers.buf[q] = q < world_size ? (unsigned char*)bufs[q] : nullptr;
  c->peers.rank = rank;
  c->peers.world = world_size;
  c->peers.region_bytes = (max_message_bytes + kSliceBytes - 1) / kSliceBytes * kSliceBytes;
  c->max_bytes = max_message_bytes;
  c->peers.spin_ticks = kSpinTicks;
  if (const char* ms = getenv("SD_IPC_TIMEOUT_MS")) {   // tests shorten the wait
    const long v = atol(ms);
    if (v > 0) c->peers.spin_ticks = (unsigned long long)v * 100000ull;
  }
  hipError_t e = hipMalloc((void**)&c->peers.errors, sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->peers.errors, 0, sizeof(unsigned));
  if (e != hipSuccess) {
    sd_set_error("sd_ipc_comm_create: %s", hipGetErrorString(e));
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void sd_ipc_comm_destroy(sd_ipc_comm* c) {
  if (!c) return;
  (void)hipFree(c->peers.errors);
  delete c;
}

extern "C" int sd_ipc_comm_set_timeout_ms(sd_ipc_comm* c, int milliseconds) {
  SD_REQUIRE(c && milliseconds >= 0, SD_ERR_INVALID, "sd_ipc_comm_set_timeout_ms: bad arguments");
  c->peers.spin_ticks = milliseconds ? (unsigned long long)milliseconds * 100000ull : kSpinTicks;
  return SD_OK;
}

extern "C" int sd_ipc_comm_errors(sd_ipc_comm* c, int* n_timeouts) {
  SD_REQUIRE(c && n_timeouts, SD_ERR_INVALID, "sd_ipc_comm_errors: null pointer");
  unsigned n = 0;
  const hipError_t e = hipMemcpy(&n, c->peers.errors, sizeof n, hipMemcpyDeviceToHost);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_ipc_comm_errors: %s", hipGetErrorString(e));
  *n_timeouts = (int)n;
  return SD_OK;
}

extern "C" int sd_ipc_allreduce(void* buf, size_t n_bf16, void* ctx, sd_stream_t stream) {
  sd_ipc_comm* c = (sd_ipc_comm*)ctx;
  SD_REQUIRE(c && buf, SD_ERR_INVALID, "sd_ipc_allreduce: null pointer");
  const size_t bytes = n_bf16 * 2;
  SD_REQUIRE(bytes % 16 == 0 && sd_aligned16(buf), SD_ERR_INVALID,
             "sd_ipc_allreduce: %zu elements: the message must be a multiple of 16 bytes, 16-byte aligned",
             n_bf16);
  SD_REQUIRE(bytes <= c->max_bytes, SD_ERR_WORKSPACE,
             "sd_ipc_allreduce: %zu bytes exceed the exchange buffers (%zu)", bytes, c->max_bytes);
  if (bytes == 0) return SD_OK;
  const int slices = (int)((bytes + kSliceBytes - 1) / kSliceBytes);
  hipLaunchKernelGGL((ipc_exchange_kernel<false>), dim3(slices < kMaxGrid ? slices : kMaxGrid), dim3(kThreads), 0, (hipStream_t)stream,
                     c->peers, (const unsigned char*)buf, (unsigned char*)buf, bytes);
  SD_CHECK_LAUNCH("sd_ipc_allreduce");
  return SD_OK;
}

extern "C" int sd_ipc_allgather(const void* send, void* recv, size_t bytes_per_rank, void* ctx,
                                sd_stream_t stream) {
  sd_ipc_comm* c = (sd_ipc_comm*)ctx;
  SD_REQUIRE(c && send && recv, SD_ERR_INVALID, "sd_ipc_allgather: null pointer");
  SD_REQUIRE(bytes_per_rank % 16 == 0 && sd_aligned16(send) && sd_aligned16(recv), SD_ERR_INVALID,
             "sd_ipc_allgather: %zu bytes per rank: must be a multiple of 16, buffers 16-byte aligned",
             bytes_per_rank);
  SD_REQUIRE(bytes_per_rank <= c->max_bytes, SD_ERR_WORKSPACE,
             "sd_ipc_allgather: %zu bytes exceed the exchange buffers (%zu)", bytes_per_rank, c->max_bytes);
  if (bytes_per_rank == 0) return SD_OK;
  const int slices = (int)((bytes_per_rank + kSliceBytes - 1) / kSliceBytes);
  hipLaunchKernelGGL((ipc_exchange_kernel<true>), dim3(slices < kMaxGrid ? slices : kMaxGrid), dim3(kThreads), 0, (hipStream_t)stream,
                     c->peers, (const unsigned char*)send, (unsigned char*)recv, bytes_per_rank);
  SD_CHECK_LAUNCH("sd_ipc_allgather");
  return SD_OK;
}
