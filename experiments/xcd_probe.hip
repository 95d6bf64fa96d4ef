// Which XCD runs which workgroup: every workgroup of a 1-D, a 2-D and a 3-D grid records the
// XCC_ID hardware register.  Build: hipcc --offload-arch=gfx950 -O2 -o build/exp/xcd_probe
// experiments/xcd_probe.hip; run on the GPU box.  The prefetch blocks of the attention
// launch rely on "linear workgroup id mod 8" being the XCD, in every launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out) {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (threadIdx.x == 0) out[lin] = __builtin_amdgcn_s_getreg(20 | (3 << 11));  // XCC_ID[3:0]
}

static int run(dim3 grid, const char* name) {
  const int n = grid.x * grid.y * grid.z;
  int* d;
  if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) return 1;
  int bad = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d, 0xff, n * sizeof(int));
    hipLaunchKernelGGL(probe, grid, dim3(256), 0, 0, d);
    std::vector<int> h(n);
    if (hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) bad += (h[i] != (i & 7));
    if (rep == 0) {
      printf("%s first 24:", name);
      for (int i = 0; i < 24 && i < n; ++i) printf(" %d", h[i]);
      printf("\n");
    }
  }
  printf("%s: %d workgroups x 3 launches, %d not on XCD (linear id mod 8)\n", name, n, bad);
  hipFree(d);
  return 0;
}

int main() {
  return run(dim3(512), "1-D 512") | run(dim3(8, 64), "2-D 8x64") | run(dim3(8, 6, 8), "3-D 8x6x8") |
         run(dim3(40), "1-D 40") | run(dim3(4, 10), "2-D 4x10");
}
