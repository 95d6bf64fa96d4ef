#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
extern "C" __global__ void prefetch_kernel(const u32x4* p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  u32x4 acc = {0,0,0,0};
  for (; i < n16; i += stride * 4) {
    u32x4 a = p[i];
    u32x4 b = (i + stride < n16) ? p[i + stride] : acc;
    u32x4 c = (i + 2*stride < n16) ? p[i + 2*stride] : acc;
    u32x4 d = (i + 3*stride < n16) ? p[i + 3*stride] : acc;
    acc ^= a ^ b ^ c ^ d;
  }
  asm volatile("" :: "v"(acc));
}
extern "C" int sd_prefetch(const void* p, size_t bytes, int blocks, void* stream) {
  hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, bytes / 16);
  return (int)hipGetLastError();
}
