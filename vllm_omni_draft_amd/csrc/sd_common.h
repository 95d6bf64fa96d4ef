// Shared device helpers for the gfx950 speculative-decoding kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sd_amd.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define SD_WAVE 64

// Error text is per host thread; set by the launch wrappers.
void sd_set_error(const char* fmt, ...);
// True if sd_norm_linear / sd_norm_qkv_rope can hold M normalised rows of K columns in LDS.
bool sd_fused_norm_fits(int M, int K, bool layer_norm);

// Weight ranges that the idle compute units of an attention launch read ahead of the launches
// that stream them (extra workgroups behind the attention grid): a matrix in 16-row tiles of
// tile_bytes, its first n_units work units (tiles, or half tiles: the 128-byte halves of every
// 256 bytes), unit u wanted by the XCD (u % consumer_grid) % 8.
struct sd_pf_range {
  const void* base;
  unsigned tile_bytes, n_units, half, consumer_grid;
};
// The range sd_linear (and its fused-norm forms) reads first for this matrix, at most max_bytes.
sd_pf_range sd_linear_pf_range(const void* W, int N, int K, int epi, int fp8, int fused_norm,
                               size_t max_bytes);
// Applies to the next attention launch of this host thread (then forgotten); n <= 2.
void sd_attn_prefetch_next(const sd_pf_range* r, int n);

#define SD_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      sd_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)

#define SD_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      sd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SD_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

static inline bool sd_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }
// Round-trip through bf16: what a bf16 tensor op would have stored.
__device__ __forceinline__ float round_bf16(float v) { return (float)(bf16_t)v; }

// Eight e4m3 weights (two 32-bit words) as the bf16 MFMA operand: every e4m3 value is exact in
// bf16, so the f32 conversion is truncated to its upper half.
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint32_t w0, uint32_t w1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
  auto pack = [](f32x2 v) {
    return __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
  };
  const u32x4 r = {pack(a), pack(b), pack(c), pack(d)};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Rotary position of frequency index i (0 <= i < D/2) of token row m of sequence b whose
// cache position is cache_pos: the cache position itself, or the multimodal one.
struct RopePos {
  const int32_t* pos;    // [3][rows] explicit positions, or null
  const int32_t* delta;  // [B] offset of all three components, or null
  int s_t, s_th;         // i < s_t: temporal, i < s_th: height, else width
  int rows;
};
__host__ inline RopePos sd_make_rope_pos(const sd_rope_positions* r, int rows) {
  RopePos o{nullptr, nullptr, 0, 0, rows};
  if (r) {
    o.pos = r->pos;
    o.delta = r->delta;
    o.s_t = r->section_t;
    o.s_th = r->section_t + r->section_h;
  }
  return o;
}
__device__ __forceinline__ int sd_rope_position(const RopePos& r, int cache_pos, int b, int m, int i) {
  if (r.pos) return r.pos[(size_t)(i < r.s_t ? 0 : (i < r.s_th ? 1 : 2)) * r.rows + m];
  return r.delta ? cache_pos + r.delta[b] : cache_pos;
}

// Slot of token t (0..15) inside a V page row: bits 2 and 3 swapped, so that
// the 8 consecutive slots a lane reads are the keys its accumulator registers
// hold after the K*Q^T MFMA.
__host__ __device__ __forceinline__ int sd_v_slot(int t) {
  return (t & 3) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3);
}
