// Probe kernels for the Infinity Cache: a dense read of a range (every byte through the
// vector path) and a sparse touch (one dword per 128-byte line, 8 KiB per wave instruction).
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
// one dword per 128-byte line
extern "C" __global__ void touch_kernel(const uint32_t* p, size_t n_lines) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n_lines; i += stride * 4) {
    uint32_t a = p[i * 32];
    uint32_t b = p[(i + stride < n_lines ? i + stride : i) * 32];
    uint32_t c = p[(i + 2 * stride < n_lines ? i + 2 * stride : i) * 32];
    uint32_t d = p[(i + 3 * stride < n_lines ? i + 3 * stride : i) * 32];
    acc ^= a ^ b ^ c ^ d;
  }
  asm volatile("" :: "v"(acc));
}
extern "C" int sd_prefetch(const void* p, size_t bytes, int blocks, void* stream) {
  hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, bytes / 16);
  return (int)hipGetLastError();
}
extern "C" int sd_touch(const void* p, size_t bytes, int blocks, void* stream) {
  hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)p, bytes / 128);
  return (int)hipGetLastError();
}
