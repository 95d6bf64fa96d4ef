// Can kernels of two processes on one GPU wait for each other through IPC-shared device memory?
// Usage: ipc_probe <rank 0|1> <dir>   (start both; each writes its handle to <dir>/h<rank>)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

__global__ void handshake(unsigned* mine, const unsigned* theirs, unsigned seq, unsigned* result) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(mine, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    unsigned seen = 0;
    while ((seen = __hip_atomic_load(theirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < seq) {
      if (wall_clock64() - t0 > 20000000ull) break;  // 200 ms at 100 MHz
      __builtin_amdgcn_s_sleep(8);
    }
    result[0] = seen;
    result[1] = (unsigned)((wall_clock64() - t0) / 100);  // microseconds
  }
}

int main(int argc, char** argv) {
  const int rank = atoi(argv[1]);
  const char* dir = argv[2];
  unsigned* buf;
  CHECK(hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocFinegrained));
  CHECK(hipMemset(buf, 0, 4096));
  hipIpcMemHandle_t h;
  CHECK(hipIpcGetMemHandle(&h, buf));
  char path[512], tmp[512];
  snprintf(tmp, sizeof tmp, "%s/h%d.tmp", dir, rank);
  snprintf(path, sizeof path, "%s/h%d", dir, rank);
  FILE* f = fopen(tmp, "wb");
  fwrite(&h, sizeof h, 1, f);
  fclose(f);
  rename(tmp, path);
  snprintf(path, sizeof path, "%s/h%d", dir, 1 - rank);
  hipIpcMemHandle_t ph;
  for (int i = 0; i < 2000; ++i) {
    f = fopen(path, "rb");
    if (f) break;
    usleep(5000);
  }
  if (!f) { fprintf(stderr, "rank %d: no peer handle\n", rank); return 2; }
  if (fread(&ph, sizeof ph, 1, f) != 1) return 2;
  fclose(f);
  unsigned* peer;
  CHECK(hipIpcOpenMemHandle((void**)&peer, ph, hipIpcMemLazyEnablePeerAccess));
  unsigned* result;
  CHECK(hipMalloc(&result, 64));
  for (unsigned seq = 1; seq <= 5; ++seq) {
    hipLaunchKernelGGL(handshake, dim3(1), dim3(64), 0, 0, buf, peer, seq, result);
    CHECK(hipDeviceSynchronize());
    unsigned r[2];
    CHECK(hipMemcpy(r, result, 8, hipMemcpyDeviceToHost));
    printf("rank %d seq %u: saw %u after %u us %s\n", rank, seq, r[0], r[1], r[0] >= seq ? "OK" : "TIMEOUT");
    fflush(stdout);
  }
  usleep(200000);  // let the peer finish before the memory goes away
  CHECK(hipIpcCloseMemHandle(peer));
  CHECK(hipFree(buf));
  return 0;
}
