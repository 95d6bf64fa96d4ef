// Weight-streaming dense ops of a decoder layer at <= 8 token rows: the
// linear layers (one pass over the weight matrix at HBM rate, every token row
// riding along in registers), RMSNorm / LayerNorm and the embedding gather.
//
// sd_linear: one wave owns R consecutive output rows; its 64 lanes stride the
// K dimension in 16-byte pieces (1 KiB contiguous per wave load), accumulate
// bf16 pair products with v_dot2c_f32_bf16 into f32, and butterfly-reduce at
// the end.  The order of additions for an output element is a function of K
// alone, so a token row gets the same bits whether it runs alone (decode) or
// beside k other rows (verify).
#include "sd_common.h"

namespace {

template <int N>
struct IntC { static constexpr int value = N; };

__device__ __forceinline__ float dot8(bf16x8 w, bf16x8 x, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1),
                                        __builtin_shufflevector(x, x, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3),
                                        __builtin_shufflevector(x, x, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 4, 5),
                                        __builtin_shufflevector(x, x, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 6, 7),
                                        __builtin_shufflevector(x, x, 6, 7), acc, false);
  return acc;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_tanh_f(float v) {
  const float c = 0.7978845608028654f;
  return 0.5f * v * (1.0f + tanhf(c * (v + 0.044715f * v * v * v)));
}

struct LinArgs {
  const bf16_t* W;
  const bf16_t* x;
  int64_t ldx;
  const bf16_t* bias;
  const bf16_t* residual;
  void* out;
  int64_t ldo;
  int N, K;
};

// M token rows, R output rows per wave, EPI epilogue (sd_epilogue).
template <int M, int R, int EPI>
__global__ __launch_bounds__(256) void linear_kernel(LinArgs a) {
  constexpr bool GATED = EPI == SD_EPI_SILU_MUL;
  constexpr int RR = GATED ? 2 * R : R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * R;
  if (n0 >= a.N) return;

  const bf16_t* wrow[RR];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int n = min(n0 + r, a.N - 1);
    wrow[r] = a.W + (size_t)n * a.K;
    if (GATED) wrow[R + r] = a.W + (size_t)(a.N + n) * a.K;
  }
  float acc[RR][M];
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

#pragma unroll 2
  for (int k = lane * 8; k < a.K; k += 512) {
    bf16x8 wv[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r)
      wv[r] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow[r] + k));
    bf16x8 xv[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
      xv[m] = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.ldx + k);
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) acc[r][m] = dot8(wv[r], xv[m], acc[r][m]);
  }

#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);

#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (lane != r * M + m) continue;
      const int n = n0 + r;
      if (n >= a.N) continue;
      float y = acc[r][m];
      if (a.bias) y += (float)a.bias[n];
      const size_t o = (size_t)m * a.ldo + n;
      if (EPI == SD_EPI_F32) {
        reinterpret_cast<float*>(a.out)[o] = y;
      } else {
        bf16_t* ob = reinterpret_cast<bf16_t*>(a.out);
        if (EPI == SD_EPI_STORE) {
          ob[o] = (bf16_t)y;
        } else if (EPI == SD_EPI_RESIDUAL) {
          ob[o] = (bf16_t)(round_bf16(y) + (float)a.residual[o]);
        } else if (EPI == SD_EPI_RELU) {
          ob[o] = (bf16_t)fmaxf(round_bf16(y), 0.f);
        } else if (EPI == SD_EPI_GELU_TANH) {
          ob[o] = (bf16_t)gelu_tanh_f(round_bf16(y));
        } else {  // SILU_MUL: gate row n, up row N + n
          float u = acc[GATED ? R + r : r][m];
          if (a.bias) u += (float)a.bias[a.N + n];
          const float g = round_bf16(silu_f(round_bf16(y)));
          ob[o] = (bf16_t)(g * round_bf16(u));
        }
      }
    }
  }
}

template <int M, int EPI>
int launch_linear_me(const LinArgs& a, hipStream_t st) {
  constexpr int R = EPI == SD_EPI_SILU_MUL ? 2 : 4;
  const int tasks = (a.N + R - 1) / R;
  hipLaunchKernelGGL((linear_kernel<M, R, EPI>), dim3((tasks + 3) / 4), dim3(256), 0, st, a);
  SD_CHECK_LAUNCH("sd_linear");
  return SD_OK;
}

template <int M>
int launch_linear_m(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case SD_EPI_STORE: return launch_linear_me<M, SD_EPI_STORE>(a, st);
    case SD_EPI_RESIDUAL: return launch_linear_me<M, SD_EPI_RESIDUAL>(a, st);
    case SD_EPI_SILU_MUL: return launch_linear_me<M, SD_EPI_SILU_MUL>(a, st);
    case SD_EPI_RELU: return launch_linear_me<M, SD_EPI_RELU>(a, st);
    case SD_EPI_F32: return launch_linear_me<M, SD_EPI_F32>(a, st);
    case SD_EPI_GELU_TANH: return launch_linear_me<M, SD_EPI_GELU_TANH>(a, st);
  }
  sd_set_error("sd_linear: unknown epilogue %d", epi);
  return SD_ERR_INVALID;
}

__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                      const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int64_t ldo,
                                                      int H, float eps) {
  __shared__ float s_part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int c = tid * 8; c < H; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j];
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) s_part[tid >> 6] = ss;
  __syncthreads();
  const float tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  const float rs = rsqrtf(tot / (float)H + eps);
  bf16_t* orow = out + (size_t)row * ldo;
  for (int c = tid * 8; c < H; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(round_bf16((float)v[j] * rs) * (float)g[j]);
    *reinterpret_cast<bf16x8*>(orow + c) = o;
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                        const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bparam,
                                                        bf16_t* __restrict__ out, int64_t ldo,
                                                        int H, float eps) {
  __shared__ float s_part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = tid * 8; c < H; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[j];
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) s_part[tid >> 6] = s;
  __syncthreads();
  const float mean = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)H;
  __syncthreads();
  float ss = 0.f;
  for (int c = tid * 8; c < H; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j] - mean;
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) s_part[tid >> 6] = ss;
  __syncthreads();
  const float var = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)H;
  const float rs = rsqrtf(var + eps);
  bf16_t* orow = out + (size_t)row * ldo;
  for (int c = tid * 8; c < H; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + c);
    bf16x8 bb = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bparam) bb = *reinterpret_cast<const bf16x8*>(bparam + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = (bf16_t)(((float)v[j] - mean) * rs * (float)g[j] + (float)bb[j]);
    *reinterpret_cast<bf16x8*>(orow + c) = o;
  }
}

__global__ __launch_bounds__(256) void embed_kernel(const int32_t* __restrict__ ids,
                                                    int64_t id_stride_seq, int64_t id_offset,
                                                    const bf16_t* __restrict__ table,
                                                    const bf16_t* __restrict__ pos_table,
                                                    int pos_bias, const int32_t* __restrict__ seq_lens,
                                                    int pos_offset, bf16_t* __restrict__ out,
                                                    int q_len, int H, int vocab) {
  const int row = blockIdx.x;
  const int b = row / q_len, t = row % q_len;
  int id = ids[(size_t)b * id_stride_seq + id_offset + t];
  id = min(max(id, 0), vocab - 1);
  const bf16_t* src = table + (size_t)id * H;
  const bf16_t* psrc = nullptr;
  if (pos_table) {
    const int pos = max(seq_lens[b] + pos_offset + t, 0);
    psrc = pos_table + (size_t)(pos + pos_bias) * H;
  }
  bf16_t* dst = out + (size_t)row * H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c);
    if (psrc) {
      const bf16x8 p = *reinterpret_cast<const bf16x8*>(psrc + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)p[j]);
    }
    *reinterpret_cast<bf16x8*>(dst + c) = v;
  }
}

}  // namespace

extern "C" int sd_linear(const void* W, const void* x, int64_t ldx, const void* bias,
                         const void* residual, void* out, int64_t ldo, int M, int N, int K,
                         int epilogue, sd_stream_t stream) {
  SD_REQUIRE(W && x && out, SD_ERR_INVALID, "sd_linear: null pointer");
  SD_REQUIRE(M >= 0 && M <= 8, SD_ERR_UNSUPPORTED, "sd_linear: M=%d outside 0..8", M);
  SD_REQUIRE(N >= 1 && K >= 8 && K % 8 == 0, SD_ERR_INVALID,
             "sd_linear: need N>=1 and K a positive multiple of 8 (N=%d K=%d)", N, K);
  SD_REQUIRE(ldx % 8 == 0 && ldx >= K && ldo >= N && sd_aligned16(W) && sd_aligned16(x),
             SD_ERR_INVALID, "sd_linear: rows must be 16-byte aligned (ldx=%lld)",
             (long long)ldx);
  SD_REQUIRE(epilogue != SD_EPI_RESIDUAL || residual, SD_ERR_INVALID,
             "sd_linear: residual epilogue without residual pointer");
  if (M == 0) return SD_OK;
  LinArgs a{(const bf16_t*)W, (const bf16_t*)x, ldx, (const bf16_t*)bias,
            (const bf16_t*)residual, out, ldo, N, K};
  hipStream_t st = (hipStream_t)stream;
  switch (M) {
    case 1: return launch_linear_m<1>(a, epilogue, st);
    case 2: return launch_linear_m<2>(a, epilogue, st);
    case 3: return launch_linear_m<3>(a, epilogue, st);
    case 4: return launch_linear_m<4>(a, epilogue, st);
    case 5: return launch_linear_m<5>(a, epilogue, st);
    case 6: return launch_linear_m<6>(a, epilogue, st);
    case 7: return launch_linear_m<7>(a, epilogue, st);
    default: return launch_linear_m<8>(a, epilogue, st);
  }
}

extern "C" int sd_rmsnorm(const void* x, int64_t ldx, const void* w, void* out, int64_t ldo,
                          int rows, int H, float eps, sd_stream_t stream) {
  SD_REQUIRE(x && w && out && rows >= 0 && H >= 8 && H % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0,
             SD_ERR_INVALID, "sd_rmsnorm: bad arguments (rows=%d H=%d)", rows, H);
  if (rows == 0) return SD_OK;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, H, eps);
  SD_CHECK_LAUNCH("sd_rmsnorm");
  return SD_OK;
}

extern "C" int sd_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* out,
                            int64_t ldo, int rows, int H, float eps, sd_stream_t stream) {
  SD_REQUIRE(x && w && out && rows >= 0 && H >= 8 && H % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0,
             SD_ERR_INVALID, "sd_layernorm: bad arguments (rows=%d H=%d)", rows, H);
  if (rows == 0) return SD_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out,
                     ldo, H, eps);
  SD_CHECK_LAUNCH("sd_layernorm");
  return SD_OK;
}

extern "C" int sd_embed(const int32_t* ids, int64_t id_stride_seq, int64_t id_offset,
                        const void* table, const void* pos_table, int pos_bias,
                        const int32_t* seq_lens, int pos_offset, void* out, int B, int q_len,
                        int H, int vocab, sd_stream_t stream) {
  SD_REQUIRE(ids && table && out && B >= 0 && q_len >= 1 && H % 8 == 0 && vocab >= 1,
             SD_ERR_INVALID, "sd_embed: bad arguments");
  SD_REQUIRE(!pos_table || seq_lens, SD_ERR_INVALID, "sd_embed: pos_table needs seq_lens");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(embed_kernel, dim3(B * q_len), dim3(256), 0, (hipStream_t)stream, ids,
                     id_stride_seq, id_offset, (const bf16_t*)table, (const bf16_t*)pos_table,
                     pos_bias, seq_lens, pos_offset, (bf16_t*)out, q_len, H, vocab);
  SD_CHECK_LAUNCH("sd_embed");
  return SD_OK;
}
