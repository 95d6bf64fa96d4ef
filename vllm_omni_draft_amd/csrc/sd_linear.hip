// Weight-streaming dense ops of a decoder layer at <= 8 token rows: the
// linear layers (one pass over the weight matrix at HBM rate, every token row
// riding along), with RMSNorm / LayerNorm fused in front and rotary embedding +
// KV-cache append fused behind, plus stand-alone norms and the embedding gather.
//
// linear_kernel: a workgroup of 4 waves stages the (optionally normalised)
// activations x[M][K-chunk] in LDS once; each wave owns R output rows and its
// 64 lanes stride the K dimension in 16-byte pieces (1 KiB contiguous per wave
// load, non-temporal: weights are read once), multiply bf16 pairs with
// v_dot2c_f32_bf16 into f32 accumulators and butterfly-reduce at the end.  The
// next iteration's weight loads are issued before the current one is consumed
// and stay in flight across the chunk barriers.
//
// Order of additions for an output element: lane-local in increasing k, then a
// fixed butterfly — a function of K alone.  A token row therefore gets the
// same bits whether it runs alone (decode) or beside k other rows (verify),
// whatever the chunking.
#include "sd_common.h"

namespace {

enum { NORM_NONE = 0, NORM_RMS = 1, NORM_LAYER = 2 };
constexpr int EPI_QKV_ROPE = 100;  // internal: sd_linear_qkv_rope
constexpr int kLinThreads = 256;
constexpr int kLinWaves = kLinThreads / SD_WAVE;

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

struct RopeArgs {
  const float* cs;  // [max_pos, D] cos | sin, or nullptr
  int max_pos;
  bf16_t* kc;
  bf16_t* vc;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset, q_len, Hq, Hkv, D;
};

struct LinArgs {
  const bf16_t* W;
  const bf16_t* x;
  int64_t ldx;
  const bf16_t* bias;
  const bf16_t* residual;
  void* out;
  int64_t ldo;
  int N, K, KC;            // KC: K-chunk staged in LDS (multiple of 512, or K)
  const bf16_t* norm_w;    // fused norm parameters
  const bf16_t* norm_b;
  float eps;
  RopeArgs rope;
};

// Sum over the workgroup of M per-thread values; result in every thread.
template <int M>
__device__ __forceinline__ void block_sum_rows(float (&v)[M], float* red /*[waves][M]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = wave_sum(v[m]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < M; ++m) red[wave * M + m] = v[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kLinWaves; ++w) t += red[w * M + m];
    v[m] = t;
  }
}

// M token rows, R output rows per wave, U 512-element K steps per iteration.
template <int M, int R, int U, int EPI, int NORM>
__global__ __launch_bounds__(kLinThreads) void linear_kernel(LinArgs a) {
  constexpr bool GATED = EPI == SD_EPI_SILU_MUL;
  constexpr bool QKV = EPI == EPI_QKV_ROPE;
  constexpr int RR = GATED ? 2 * R : R;
  static_assert(!QKV || R == 4, "the rotary epilogue pairs rows (i, i + D/2): R must be 4");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem_raw);                       // [M][KC]
  float* red = reinterpret_cast<float*>(smem_raw + (size_t)M * a.KC * 2);  // [waves][M]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int task = blockIdx.x * kLinWaves + wave;
  // Output rows of this wave.
  int rows[R];
  if (QKV) {
    const int quarter = a.rope.D / 4;  // tasks per head
    const int head = task / quarter, i0 = 2 * (task % quarter);
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = head * a.rope.D + i0 + (r & 1) + (r >> 1) * (a.rope.D / 2);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = task * R + r;
  }
  const bf16_t* wrow[RR];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int n = min(rows[r], a.N - 1);
    wrow[r] = a.W + (size_t)n * a.K;
    if (GATED) wrow[R + r] = a.W + (size_t)(a.N + n) * a.K;
  }

  float acc[RR][M];
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

  // Weight fragments of the next iteration (absolute k of lane piece u: kb + 512u).
  bf16x8 nxt[RR][U];
  auto load_w = [&](int kb, int kend) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = kb + u * 512;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (kk < kend) v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow[r] + kk));
        nxt[r][u] = v;
      }
    }
  };
  load_w(lane * 8, min(a.KC, a.K));

  for (int kc0 = 0; kc0 < a.K; kc0 += a.KC) {
    const int kc_len = min(a.KC, a.K - kc0);
    // ---- stage x[:, kc0 : kc0 + kc_len] (normalised if asked) into LDS
    if (NORM == NORM_NONE) {
      for (int i = tid * 8; i < M * kc_len; i += kLinThreads * 8) {
        const int m = i / kc_len, c = i - m * kc_len;
        *reinterpret_cast<bf16x8*>(xs + (size_t)m * a.KC + c) =
            *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.ldx + kc0 + c);
      }
    } else {  // single chunk: kc_len == K == hidden size
      float s1[M], s2[M];
#pragma unroll
      for (int m = 0; m < M; ++m) s1[m] = s2[m] = 0.f;
      if (NORM == NORM_LAYER) {
        for (int c = tid * 8; c < a.K; c += kLinThreads * 8)
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.ldx + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s1[m] += (float)v[j];
          }
        block_sum_rows<M>(s1, red);
#pragma unroll
        for (int m = 0; m < M; ++m) s1[m] = s1[m] / (float)a.K;  // mean
      }
      for (int c = tid * 8; c < a.K; c += kLinThreads * 8)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.ldx + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j] - s1[m];
            s2[m] += f * f;
          }
        }
      block_sum_rows<M>(s2, red);
#pragma unroll
      for (int m = 0; m < M; ++m) s2[m] = rsqrtf(s2[m] / (float)a.K + a.eps);
      for (int c = tid * 8; c < a.K; c += kLinThreads * 8) {
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(a.norm_w + c);
        bf16x8 bb = {0, 0, 0, 0, 0, 0, 0, 0};
        if (NORM == NORM_LAYER && a.norm_b) bb = *reinterpret_cast<const bf16x8*>(a.norm_b + c);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.ldx + c);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (NORM == NORM_RMS)
              o[j] = (bf16_t)(round_bf16((float)v[j] * s2[m]) * (float)g[j]);
            else
              o[j] = (bf16_t)(((float)v[j] - s1[m]) * s2[m] * (float)g[j] + (float)bb[j]);
          }
          *reinterpret_cast<bf16x8*>(xs + (size_t)m * a.KC + c) = o;
        }
      }
    }
    __syncthreads();

    // ---- stream this wave's weight rows over the chunk
    const int step = 512 * U;
    const int n_it = (kc_len + step - 1) / step;
    for (int it = 0; it < n_it; ++it) {
      bf16x8 cur[RR][U];
#pragma unroll
      for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u) cur[r][u] = nxt[r][u];
      if (it + 1 < n_it) {
        load_w(kc0 + (it + 1) * step + lane * 8, kc0 + kc_len);
      } else if (kc0 + a.KC < a.K) {
        load_w(kc0 + a.KC + lane * 8, min(a.K, kc0 + 2 * a.KC));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = it * step + u * 512 + lane * 8;  // chunk-relative
        if (kk < kc_len) {
          bf16x8 xv[M];
#pragma unroll
          for (int m = 0; m < M; ++m)
            xv[m] = *reinterpret_cast<const bf16x8*>(xs + (size_t)m * a.KC + kk);
#pragma unroll
          for (int r = 0; r < RR; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = dot8(cur[r][u], xv[m], acc[r][m]);
        }
      }
    }
    if (kc0 + a.KC < a.K) __syncthreads();  // LDS is restaged next
  }

#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);

  if (QKV) {
    // lane p*M + m finishes the pair (rows[p], rows[p+2]) of token row m
    const RopeArgs& rp = a.rope;
    const int D = rp.D, half = D / 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (lane != p * M + m) continue;
        const int n_lo = rows[p], n_hi = rows[p + 2];
        if (n_hi >= a.N) continue;
        float lo = acc[p][m], hi = acc[p + 2][m];
        if (a.bias) {
          lo += (float)a.bias[n_lo];
          hi += (float)a.bias[n_hi];
        }
        lo = round_bf16(lo);
        hi = round_bf16(hi);
        const int head = n_lo / D, i = n_lo - head * D;  // i < half
        const int b = m / rp.q_len, t = m - b * rp.q_len;
        const int pos = rp.seq_lens[b] + rp.pos_offset + t;
        if (pos < 0) continue;
        if (head < rp.Hq + rp.Hkv && rp.cs) {
          const float* cs = rp.cs + (size_t)min(pos, rp.max_pos - 1) * D;
          const float c = cs[i], s = cs[half + i];
          const float nlo = lo * c - hi * s, nhi = hi * c + lo * s;
          lo = nlo;
          hi = nhi;
        }
        if (head < rp.Hq) {
          bf16_t* q = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + (size_t)head * D;
          q[i] = (bf16_t)lo;
          q[i + half] = (bf16_t)hi;
        } else {
          const int page = rp.block_tables[(size_t)b * rp.max_blocks + (pos >> 4)];
          const int slot = pos & 15;
          if (head < rp.Hq + rp.Hkv) {
            bf16_t* kp = rp.kc + ((size_t)page * rp.Hkv + (head - rp.Hq)) * D * 16;
            const int i2 = i + half;
            kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = (bf16_t)lo;
            kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = (bf16_t)hi;
          } else {
            bf16_t* vp = rp.vc + ((size_t)page * rp.Hkv + (head - rp.Hq - rp.Hkv)) * D * 16;
            const int sl = sd_v_slot(slot);
            vp[i * 16 + sl] = (bf16_t)lo;
            vp[(i + half) * 16 + sl] = (bf16_t)hi;
          }
        }
      }
    }
    return;
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (lane != r * M + m) continue;
      const int n = rows[r];
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

// ------------------------------------------------------------------ launch
constexpr size_t kLdsSoft = 64 * 1024;   // keeps two workgroups per CU
constexpr size_t kLdsHard = 144 * 1024;  // fused norm needs the whole row

// Chooses the K-chunk; returns 0 if the fused norm cannot hold x in LDS.
int choose_kc(int M, int K, bool norm) {
  const size_t row_bytes = (size_t)K * 2;
  if ((size_t)M * row_bytes <= kLdsSoft) return K;
  if (norm) return (size_t)M * row_bytes <= kLdsHard ? K : 0;
  int kc = (int)(kLdsSoft / ((size_t)M * 2)) / 512 * 512;
  return kc < 512 ? 512 : kc;
}

template <int M, int R, int U, int EPI, int NORM>
int launch_one(LinArgs& a, int tasks, hipStream_t st) {
  const size_t lds = (size_t)M * a.KC * 2 + kLinWaves * M * sizeof(float);
  if (lds > 48 * 1024) {
    static bool attr_set = false;  // once per instantiation and process
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(linear_kernel<M, R, U, EPI, NORM>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsHard + 1024);
      if (e != hipSuccess) {
        sd_set_error("sd_linear: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return SD_ERR_HIP;
      }
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((linear_kernel<M, R, U, EPI, NORM>), dim3((tasks + kLinWaves - 1) / kLinWaves),
                     dim3(kLinThreads), lds, st, a);
  SD_CHECK_LAUNCH("sd_linear");
  return SD_OK;
}

template <int M, int EPI, int NORM>
int launch_me(LinArgs& a, hipStream_t st) {
  if constexpr (EPI == EPI_QKV_ROPE) {
    return launch_one<M, 4, 2, EPI, NORM>(a, a.N / 4, st);
  } else if constexpr (EPI == SD_EPI_SILU_MUL) {
    return launch_one<M, 2, 2, EPI, NORM>(a, (a.N + 1) / 2, st);
  } else {
    // few output rows: halve the rows per wave so that every CU still gets waves
    if (a.N <= 4096) return launch_one<M, 2, 4, EPI, NORM>(a, (a.N + 1) / 2, st);
    return launch_one<M, 4, 2, EPI, NORM>(a, (a.N + 3) / 4, st);
  }
}

template <int M, int NORM>
int launch_m(LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case SD_EPI_STORE: return launch_me<M, SD_EPI_STORE, NORM>(a, st);
    case SD_EPI_SILU_MUL: return launch_me<M, SD_EPI_SILU_MUL, NORM>(a, st);
    case SD_EPI_RELU: return launch_me<M, SD_EPI_RELU, NORM>(a, st);
    case SD_EPI_F32: return launch_me<M, SD_EPI_F32, NORM>(a, st);
    case EPI_QKV_ROPE: return launch_me<M, EPI_QKV_ROPE, NORM>(a, st);
    case SD_EPI_RESIDUAL:
      if (NORM == NORM_NONE) return launch_me<M, SD_EPI_RESIDUAL, NORM_NONE>(a, st);
      break;
    case SD_EPI_GELU_TANH:
      if (NORM == NORM_NONE) return launch_me<M, SD_EPI_GELU_TANH, NORM_NONE>(a, st);
      break;
  }
  sd_set_error("sd_linear: epilogue %d not available with norm mode %d", epi, NORM);
  return SD_ERR_UNSUPPORTED;
}

template <int NORM>
int launch_n(LinArgs& a, int M, int epi, hipStream_t st) {
  switch (M) {
    case 1: return launch_m<1, NORM>(a, epi, st);
    case 2: return launch_m<2, NORM>(a, epi, st);
    case 3: return launch_m<3, NORM>(a, epi, st);
    case 4: return launch_m<4, NORM>(a, epi, st);
    case 5: return launch_m<5, NORM>(a, epi, st);
    case 6: return launch_m<6, NORM>(a, epi, st);
    case 7: return launch_m<7, NORM>(a, epi, st);
    default: return launch_m<8, NORM>(a, epi, st);
  }
}

int linear_common(LinArgs& a, int M, int epi, int norm, hipStream_t st) {
  SD_REQUIRE(a.W && a.x && a.out, SD_ERR_INVALID, "sd_linear: null pointer");
  SD_REQUIRE(M >= 0 && M <= 8, SD_ERR_UNSUPPORTED, "sd_linear: M=%d outside 0..8", M);
  SD_REQUIRE(a.N >= 1 && a.K >= 8 && a.K % 8 == 0, SD_ERR_INVALID,
             "sd_linear: need N>=1 and K a positive multiple of 8 (N=%d K=%d)", a.N, a.K);
  SD_REQUIRE(a.ldx % 8 == 0 && a.ldx >= a.K && sd_aligned16(a.W) && sd_aligned16(a.x),
             SD_ERR_INVALID, "sd_linear: rows must be 16-byte aligned (ldx=%lld)", (long long)a.ldx);
  SD_REQUIRE(epi != SD_EPI_RESIDUAL || a.residual, SD_ERR_INVALID,
             "sd_linear: residual epilogue without residual pointer");
  SD_REQUIRE(norm == NORM_NONE || a.norm_w, SD_ERR_INVALID, "sd_linear: fused norm without weight");
  if (M == 0) return SD_OK;
  a.KC = choose_kc(M, a.K, norm != NORM_NONE);
  SD_REQUIRE(a.KC > 0, SD_ERR_UNSUPPORTED,
             "sd_linear: fused norm needs M*K*2 <= %zu bytes of LDS (M=%d K=%d)", kLdsHard, M, a.K);
  switch (norm) {
    case NORM_NONE: return launch_n<NORM_NONE>(a, M, epi, st);
    case NORM_RMS: return launch_n<NORM_RMS>(a, M, epi, st);
    case NORM_LAYER: return launch_n<NORM_LAYER>(a, M, epi, st);
  }
  sd_set_error("sd_linear: unknown norm mode %d", norm);
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
  const float tot = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
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
  const float mean = (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]) / (float)H;
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
  const float var = (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]) / (float)H;
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
  SD_REQUIRE(epilogue >= SD_EPI_STORE && epilogue <= SD_EPI_GELU_TANH, SD_ERR_INVALID,
             "sd_linear: unknown epilogue %d", epilogue);
  SD_REQUIRE(ldo >= N, SD_ERR_INVALID, "sd_linear: ldo < N");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.residual = (const bf16_t*)residual; a.out = out; a.ldo = ldo; a.N = N; a.K = K;
  return linear_common(a, M, epilogue, NORM_NONE, (hipStream_t)stream);
}

extern "C" int sd_norm_linear(const void* W, const void* x, int64_t ldx, const void* norm_w,
                              const void* norm_b, float eps, int layer_norm, const void* bias,
                              void* out, int64_t ldo, int M, int N, int K, int epilogue,
                              sd_stream_t stream) {
  SD_REQUIRE(epilogue == SD_EPI_STORE || epilogue == SD_EPI_SILU_MUL || epilogue == SD_EPI_RELU ||
                 epilogue == SD_EPI_F32, SD_ERR_UNSUPPORTED,
             "sd_norm_linear: epilogue %d not supported with a fused norm", epilogue);
  SD_REQUIRE(ldo >= N, SD_ERR_INVALID, "sd_norm_linear: ldo < N");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.out = out; a.ldo = ldo; a.N = N; a.K = K;
  a.norm_w = (const bf16_t*)norm_w; a.norm_b = (const bf16_t*)norm_b; a.eps = eps;
  return linear_common(a, M, epilogue, layer_norm ? NORM_LAYER : NORM_RMS, (hipStream_t)stream);
}

extern "C" int sd_norm_qkv_rope(const void* W, const void* x, int64_t ldx, const void* norm_w,
                                const void* norm_b, float eps, int layer_norm, const void* bias,
                                void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                                void* k_cache, void* v_cache, const int32_t* block_tables,
                                int max_blocks_per_seq, const int32_t* seq_lens, int pos_offset,
                                int B, int q_len, int Hq, int Hkv, int D, int K,
                                sd_stream_t stream) {
  SD_REQUIRE(q_out && k_cache && v_cache && block_tables && seq_lens, SD_ERR_INVALID,
             "sd_norm_qkv_rope: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && D % 16 == 0 && D >= 16,
             SD_ERR_INVALID, "sd_norm_qkv_rope: bad shape (D=%d)", D);
  SD_REQUIRE(!rope_cs || max_pos >= 1, SD_ERR_INVALID, "sd_norm_qkv_rope: max_pos < 1");
  SD_REQUIRE(ldq >= (int64_t)Hq * D, SD_ERR_INVALID, "sd_norm_qkv_rope: ldq too small");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.out = q_out; a.ldo = ldq; a.N = (Hq + 2 * Hkv) * D; a.K = K;
  a.norm_w = (const bf16_t*)norm_w; a.norm_b = (const bf16_t*)norm_b; a.eps = eps;
  a.rope = RopeArgs{rope_cs, max_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, block_tables,
                    max_blocks_per_seq, seq_lens, pos_offset, q_len, Hq, Hkv, D};
  const int norm = norm_w ? (layer_norm ? NORM_LAYER : NORM_RMS) : NORM_NONE;
  return linear_common(a, B * q_len, EPI_QKV_ROPE, norm, (hipStream_t)stream);
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
