// CPU build of the C ABI in include/sd_amd.h (C++ / OpenMP, fp32 accumulation).
//
// A separate library (libsd_cpu.so) with the same symbols, data layouts (tiled weights, paged
// KV cache, token window) and rounding points (bf16 between ops) as the gfx950 library, for
// hosts without a GPU: the plumbing configuration (OPT-125m target + draft, greedy, one
// process) and the CPU baseline of the benchmark.  It is selected when the host code is told
// to (SD_AMD_BACKEND=cpu) and never stands in for the GPU library: nothing falls back to it.
// Pointers are host pointers, `stream` is ignored, every call is synchronous.  Not provided
// (SD_ERR_UNSUPPORTED): fp8 weights, multimodal rotary positions, tensor-parallel groups,
// stream capture.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <functional>
#include <vector>

#include <omp.h>

#include "../../include/sd_amd.h"

namespace {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
#define REQUIRE(cond, code, ...) \
  do {                           \
    if (!(cond)) {               \
      set_error(__VA_ARGS__);    \
      return (code);             \
    }                            \
  } while (0)
#define UNSUPPORTED(name)                                              \
  do {                                                                 \
    set_error("%s: not provided by the CPU build of the library", name); \
    return SD_ERR_UNSUPPORTED;                                         \
  } while (0)

typedef uint16_t bf16;
inline float bf2f(bf16 v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline bf16 f2bf(float f) {  // round to nearest even; NaN stays NaN
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16)(u >> 16);
}
inline float rb(float f) { return bf2f(f2bf(f)); }
inline int v_slot(int t) { return (t & 3) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3); }

// ---- tiled weights: tile (nt, ks) = [64 lanes][8]: lane l holds tile row l & 15, columns
// 32 ks + 8 (l >> 4) .. + 8
inline int tile_src_row(int mode, int p0, int rows_src, int nt, int r) {
  if (mode == SD_TILE_PLAIN) return nt * 16 + r;
  if (mode == SD_TILE_GATED) {  // p0 = F
    const int n = nt * 8 + (r & 7);
    return n < p0 ? (r < 8 ? n : p0 + n) : rows_src;
  }
  const int tph = p0 / 16, head = nt / tph, i0 = (nt % tph) * 8;  // ROTARY, p0 = D
  return head * p0 + (r < 8 ? i0 + r : i0 + p0 / 2 + (r - 8));
}

// sums[r][m] (r = 16 tile rows, m < M) = sum_c W_tile[r][c] * x[m][c]
void tile_dot(const bf16* W, int KS, int nt, const float* x, int64_t ldx, int M, float* sums) {
  for (int i = 0; i < 16 * M; ++i) sums[i] = 0.f;
  const bf16* wt = W + (size_t)nt * KS * 512;
  for (int ks = 0; ks < KS; ++ks)
    for (int q = 0; q < 4; ++q) {
      const int c0 = ks * 32 + q * 8;
      for (int r = 0; r < 16; ++r) {
        const bf16* w8 = wt + ((size_t)ks * 64 + q * 16 + r) * 8;
        float wf[8];
        for (int j = 0; j < 8; ++j) wf[j] = bf2f(w8[j]);
        for (int m = 0; m < M; ++m) {
          const float* xr = x + (size_t)m * ldx + c0;
          float s = 0.f;
          for (int j = 0; j < 8; ++j) s += wf[j] * xr[j];
          sums[r * M + m] += s;
        }
      }
    }
}

inline float silu(float v) { return v / (1.0f + expf(-v)); }
inline float gelu_tanh(float v) {
  const float c = 0.7978845608028654f;
  return 0.5f * v * (1.0f + tanhf(c * (v + 0.044715f * v * v * v)));
}

// x rows as f32 (from bf16)
std::vector<float> rows_f32(const bf16* x, int64_t ldx, int M, int K) {
  std::vector<float> o((size_t)M * K);
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < K; ++c) o[(size_t)m * K + c] = bf2f(x[(size_t)m * ldx + c]);
  return o;
}

void norm_rows(std::vector<float>& x, int M, int K, const bf16* w, const bf16* b, float eps,
               bool layer_norm) {
  for (int m = 0; m < M; ++m) {
    float* r = x.data() + (size_t)m * K;
    if (layer_norm) {
      double s = 0;
      for (int c = 0; c < K; ++c) s += r[c];
      const float mean = (float)(s / K);
      double ss = 0;
      for (int c = 0; c < K; ++c) ss += (double)(r[c] - mean) * (r[c] - mean);
      const float rs = 1.0f / sqrtf((float)(ss / K) + eps);
      for (int c = 0; c < K; ++c)
        r[c] = rb((r[c] - mean) * rs * bf2f(w[c]) + (b ? bf2f(b[c]) : 0.f));
    } else {
      double ss = 0;
      for (int c = 0; c < K; ++c) ss += (double)r[c] * r[c];
      const float rs = 1.0f / sqrtf((float)(ss / K) + eps);
      for (int c = 0; c < K; ++c) r[c] = rb(rb(r[c] * rs) * bf2f(w[c]));
    }
  }
}

// out = epilogue(x W^T): x f32 [M][K] (already bf16-valued), W tiled.
int linear_f32(const bf16* W, const float* x, int M, int N, int K, const bf16* bias,
               const bf16* residual, void* out, int64_t ldo, int epi) {
  const bool gated = epi == SD_EPI_SILU_MUL;
  const int n_tiles = gated ? (N + 7) / 8 : (N + 15) / 16;
  const int KS = K / 32;
#pragma omp parallel
  {
    std::vector<float> sums((size_t)16 * M);
#pragma omp for schedule(static)
    for (int nt = 0; nt < n_tiles; ++nt) {
      tile_dot(W, KS, nt, x, K, M, sums.data());
      for (int m = 0; m < M; ++m) {
        if (gated) {
          for (int r = 0; r < 8; ++r) {
            const int n = nt * 8 + r;
            if (n >= N) break;
            float g = sums[r * M + m] + (bias ? bf2f(bias[n]) : 0.f);
            float u = sums[(r + 8) * M + m] + (bias ? bf2f(bias[N + n]) : 0.f);
            g = rb(silu(rb(g)));
            ((bf16*)out)[(size_t)m * ldo + n] = f2bf(g * rb(u));
          }
          continue;
        }
        for (int r = 0; r < 16; ++r) {
          const int n = nt * 16 + r;
          if (n >= N) break;
          const float y = sums[r * M + m] + (bias ? bf2f(bias[n]) : 0.f);
          const size_t o = (size_t)m * ldo + n;
          if (epi == SD_EPI_F32) ((float*)out)[o] = y;
          else if (epi == SD_EPI_STORE) ((bf16*)out)[o] = f2bf(y);
          else if (epi == SD_EPI_RESIDUAL) ((bf16*)out)[o] = f2bf(rb(y) + bf2f(residual[o]));
          else if (epi == SD_EPI_RELU) ((bf16*)out)[o] = f2bf(fmaxf(rb(y), 0.f));
          else ((bf16*)out)[o] = f2bf(gelu_tanh(rb(y)));
        }
      }
    }
  }
  return SD_OK;
}

struct Rope {
  const float* cs;
  int max_pos;
  bf16* kc;
  bf16* vc;
  const int32_t* bt;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset, q_len, Hq, Hkv, D;
  const int32_t* rot_pos = nullptr;  // [rows] explicit rotary positions (token trees), or null
};

// QKV projection (ROTARY tiles) + rotary embedding + cache append; q rows to q_out.
int qkv_rope_f32(const bf16* W, const float* x, int M, int K, const bf16* bias, bf16* q_out,
                 int64_t ldq, const Rope& rp) {
  const int D = rp.D, half = D / 2, tph = D / 16;
  const int n_tiles = (rp.Hq + 2 * rp.Hkv) * tph, KS = K / 32;
#pragma omp parallel
  {
    std::vector<float> sums((size_t)16 * M);
#pragma omp for schedule(static)
    for (int nt = 0; nt < n_tiles; ++nt) {
      tile_dot(W, KS, nt, x, K, M, sums.data());
      const int head = nt / tph, i0 = (nt % tph) * 8;
      for (int m = 0; m < M; ++m) {
        const int b = m / rp.q_len, t = m - b * rp.q_len;
        const int pos = rp.seq_lens[b] + rp.pos_offset + t;
        if (pos < 0) continue;
        for (int r = 0; r < 8; ++r) {
          const int i = i0 + r;
          float lo = sums[r * M + m], hi = sums[(r + 8) * M + m];
          if (bias) {
            lo += bf2f(bias[head * D + i]);
            hi += bf2f(bias[head * D + i + half]);
          }
          lo = rb(lo);
          hi = rb(hi);
          if (head < rp.Hq + rp.Hkv && rp.cs) {
            const int rpos = rp.rot_pos ? std::max(rp.rot_pos[m], 0) : pos;
            const float* cs = rp.cs + (size_t)std::min(rpos, rp.max_pos - 1) * D;
            const float nlo = lo * cs[i] - hi * cs[half + i], nhi = hi * cs[i] + lo * cs[half + i];
            lo = nlo;
            hi = nhi;
          }
          if (head < rp.Hq) {
            bf16* q = q_out + (size_t)m * ldq + (size_t)head * D;
            q[i] = f2bf(lo);
            q[i + half] = f2bf(hi);
            continue;
          }
          const int page = rp.bt[(size_t)b * rp.max_blocks + (pos >> 4)], slot = pos & 15;
          if (head < rp.Hq + rp.Hkv) {
            bf16* kp = rp.kc + ((size_t)page * rp.Hkv + (head - rp.Hq)) * D * 16;
            const int i2 = i + half;
            kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = f2bf(lo);
            kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = f2bf(hi);
          } else {
            bf16* vp = rp.vc + ((size_t)page * rp.Hkv + (head - rp.Hq - rp.Hkv)) * D * 16;
            vp[i * 16 + v_slot(slot)] = f2bf(lo);
            vp[(i + half) * 16 + v_slot(slot)] = f2bf(hi);
          }
        }
      }
    }
  }
  return SD_OK;
}

// Causal attention of q_len new tokens per sequence over the paged cache.
void attention_f32(const bf16* q, int64_t ldq, const bf16* kc, const bf16* vc, const int32_t* bt,
                   int max_blocks, const int32_t* seq_lens, int pos_offset, bf16* out, int64_t ldo,
                   int B, int q_len, int Hq, int Hkv, int D, float scale,
                   const uint64_t* tree_mask = nullptr) {
  const int G = Hq / Hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int m = 0; m < B * q_len; ++m)
    for (int h = 0; h < Hq; ++h) {
      const int b = m / q_len, t = m - b * q_len, kvh = h / G;
      const int n = seq_lens[b] + pos_offset + t + 1;  // keys 0..n-1
      bf16* o = out + (size_t)m * ldo + (size_t)h * D;
      if (n <= 0) {
        for (int d = 0; d < D; ++d) o[d] = 0;
        continue;
      }
      std::vector<float> qf(D), s(n), acc(D, 0.f);
      for (int d = 0; d < D; ++d) qf[d] = bf2f(q[(size_t)m * ldq + (size_t)h * D + d]);
      float mx = -INFINITY;
      const int pos0 = n - 1 - t;  // slot of the sequence's first new token
      for (int p = 0; p < n; ++p) {
        if (tree_mask && p >= pos0 && !((tree_mask[m] >> (p - pos0)) & 1ull)) {
          s[p] = -INFINITY;  // a new token outside this row's ancestors
          continue;
        }
        const bf16* kp = kc + ((size_t)bt[(size_t)b * max_blocks + (p >> 4)] * Hkv + kvh) * D * 16;
        float dot = 0.f;
        for (int d = 0; d < D; ++d) dot += qf[d] * bf2f(kp[((d >> 3) * 16 + (p & 15)) * 8 + (d & 7)]);
        s[p] = dot * scale;
        mx = fmaxf(mx, s[p]);
      }
      float l = 0.f;
      for (int p = 0; p < n; ++p) {
        s[p] = expf(s[p] - mx);
        l += s[p];
      }
      for (int p = 0; p < n; ++p) {
        const bf16* vp = vc + ((size_t)bt[(size_t)b * max_blocks + (p >> 4)] * Hkv + kvh) * D * 16;
        const float w = rb(s[p]);  // the GPU kernel feeds bf16 probabilities to the P.V product
        const int sl = v_slot(p & 15);
        for (int d = 0; d < D; ++d) acc[d] += w * bf2f(vp[d * 16 + sl]);
      }
      for (int d = 0; d < D; ++d) o[d] = f2bf(acc[d] / l);
    }
}

int first_argmax(const float* row, int V) {
  float best = -INFINITY;
  int bi = 0;
  bool any = false;
  for (int i = 0; i < V; ++i)
    if (row[i] > best) {
      best = row[i];
      bi = i;
      any = true;
    }
  return any ? bi : 0;
}

// Inverse-CDF draw from norm(max(p - q, 0)) with an f64 running sum (see sd_amd.h).
int sample_residual(const float* p, const float* q, int V, float u, const int* excl = nullptr,
                    int n_excl = 0) {
  auto mass = [&](int i) {
    for (int e = 0; e < n_excl; ++e)
      if (excl[e] == i) return 0.f;
    return fmaxf(p[i] - (q ? q[i] : 0.f), 0.f);
  };
  double total = 0.0;
  int last_pos = -1;
  for (int i = 0; i < V; ++i) {
    const float r = mass(i);
    total += (double)r;
    if (r > 0.f) last_pos = i;
  }
  if (!(total > 0.0)) return first_argmax(p, V);
  const double thr = (double)u * total;
  double c = 0.0;
  for (int i = 0; i < V; ++i) {
    c += (double)mass(i);
    if (c > thr) return i;
  }
  return last_pos;
}

void count(int64_t* counters, int j, int L) {
  if (!counters) return;
#pragma omp atomic
  counters[0] += j;
#pragma omp atomic
  counters[1] += j + 1;
#pragma omp atomic
  counters[2] += L;
}

void apply_emitted(const int64_t* o, int b, int32_t* seq_lens, int32_t* tok_buf, int32_t* gen_tokens,
                   int64_t gen_stride, int gen_capacity, int32_t* gen_counts, int32_t* step_out,
                   int k) {
  int n_emit = 0;
  while (n_emit <= k && o[n_emit] >= 0) ++n_emit;
  if (step_out) {
    int32_t* rec = step_out + (size_t)b * (k + 3);
    rec[0] = n_emit;
    rec[1] = seq_lens[b] + n_emit;
    for (int i = 0; i <= k; ++i) rec[2 + i] = i < n_emit ? (int32_t)o[i] : -1;
  }
  if (n_emit == 0) return;
  int32_t* tb = tok_buf + (size_t)b * (k + 2);
  const int32_t last = (int32_t)o[n_emit - 1];
  const int32_t prev = n_emit >= 2 ? (int32_t)o[n_emit - 2] : tb[1];
  const int g = gen_counts[b];
  for (int i = 0; i < n_emit; ++i)
    if (gen_tokens && g + i < gen_capacity) gen_tokens[(size_t)b * gen_stride + g + i] = (int32_t)o[i];
  gen_counts[b] = g + n_emit;
  seq_lens[b] += n_emit;
  tb[0] = prev;
  tb[1] = last;
}

int finish_partials(const float* pv, const int32_t* pi, int row) {
  float best = pv[(size_t)row * SD_ARGMAX_PARTS];
  int bi = pi[(size_t)row * SD_ARGMAX_PARTS];
  for (int i = 1; i < SD_ARGMAX_PARTS; ++i) {
    const float v = pv[(size_t)row * SD_ARGMAX_PARTS + i];
    const int id = pi[(size_t)row * SD_ARGMAX_PARTS + i];
    if (v > best || (v == best && id < bi)) {
      best = v;
      bi = id;
    }
  }
  return bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

// ------------------------------------------------------------------ library
extern "C" int sd_version(void) { return 100; }
extern "C" int sd_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return SD_ERR_INVALID;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = '\0';
  return SD_OK;
}
extern "C" int sd_device_info(int* cu_count, char* name, size_t n) {
  if (cu_count) *cu_count = omp_get_max_threads();
  if (name && n) {
    strncpy(name, "cpu", n - 1);
    name[n - 1] = '\0';
  }
  return SD_OK;
}

// ------------------------------------------------------------------ acceptance
extern "C" int sd_rejection_sample(const float* tp, const int64_t* bonus, const float* dp,
                                   const int64_t* dids, const float* ua, const float* ur,
                                   const int32_t* lens, int64_t* out, int64_t* counters, int B, int k,
                                   int V, sd_stream_t) {
  REQUIRE(tp && bonus && dp && dids && ua && ur && out, SD_ERR_INVALID, "sd_rejection_sample: null pointer");
  REQUIRE(B >= 0 && k >= 1 && k <= 63 && V >= 1, SD_ERR_INVALID, "sd_rejection_sample: bad shape");
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    const int L = std::min(std::max(lens ? lens[b] : k, 0), k);
    const float* pb = tp + (size_t)b * (k + 1) * V;
    const float* qb = dp + (size_t)b * k * V;
    int j = 0;
    for (; j < L; ++j) {
      const int64_t x = dids[(size_t)b * k + j];
      bool ok = false;
      if (x >= 0 && x < V) ok = ua[(size_t)b * k + j] * qb[(size_t)j * V + x] < pb[(size_t)j * V + x];
      if (!ok) break;
    }
    int64_t tok;
    if (j == L && L == k) tok = bonus[b];
    else tok = sample_residual(pb + (size_t)j * V, j < L ? qb + (size_t)j * V : nullptr, V,
                               ur[(size_t)b * (k + 1) + j]);
    for (int i = 0; i <= k; ++i)
      out[(size_t)b * (k + 1) + i] = i < j ? dids[(size_t)b * k + i] : (i == j ? tok : -1);
    count(counters, j, L);
  }
  return SD_OK;
}

extern "C" int sd_rejection_sample_split(const float* tp, const int64_t* bonus, const float* dp,
                                         const int64_t* dids, const float* ua, const float* ur,
                                         const int32_t* lens, int64_t* out, int64_t* counters, int B,
                                         int k, int V, void*, sd_stream_t s) {
  return sd_rejection_sample(tp, bonus, dp, dids, ua, ur, lens, out, counters, B, k, V, s);
}

extern "C" int sd_typical_acceptance(const float* tp, const int64_t* bonus, const int64_t* dids,
                                     const int32_t* lens, float cap, float alpha, int64_t* out,
                                     int64_t* counters, int B, int k, int V, sd_stream_t) {
  REQUIRE(tp && bonus && dids && out && B >= 0 && k >= 1 && V >= 1, SD_ERR_INVALID,
          "sd_typical_acceptance: bad arguments");
  for (int b = 0; b < B; ++b) {
    const int L = std::min(std::max(lens ? lens[b] : k, 0), k);
    const float* pb = tp + (size_t)b * (k + 1) * V;
    int j = 0;
    for (; j < L; ++j) {
      const float* row = pb + (size_t)j * V;
      double acc = 0.0;
      for (int i = 0; i < V; ++i) acc += (double)(row[i] * logf(row[i] + 1e-5f));
      const float thr = (float)fmin((double)cap, (double)alpha * exp(acc));
      const int64_t x = dids[(size_t)b * k + j];
      const float cand = (x >= 0 && x < V) ? row[x] : -1.f;
      if (!(cand > thr)) break;
    }
    const int64_t tok = (j == L && L == k) ? bonus[b] : first_argmax(pb + (size_t)j * V, V);
    for (int i = 0; i <= k; ++i)
      out[(size_t)b * (k + 1) + i] = i < j ? dids[(size_t)b * k + i] : (i == j ? tok : -1);
    count(counters, j, L);
  }
  return SD_OK;
}

extern "C" int sd_greedy_verify(const int64_t* tids, const int64_t* dids, const int32_t* lens,
                                int64_t* out, int64_t* counters, int B, int k, sd_stream_t) {
  REQUIRE(tids && dids && out && B >= 0 && k >= 1, SD_ERR_INVALID, "sd_greedy_verify: bad arguments");
  for (int b = 0; b < B; ++b) {
    const int L = std::min(std::max(lens ? lens[b] : k, 0), k);
    int j = 0;
    while (j < L && dids[(size_t)b * k + j] == tids[(size_t)b * (k + 1) + j]) ++j;
    for (int i = 0; i <= k; ++i)
      out[(size_t)b * (k + 1) + i] =
          i < j ? dids[(size_t)b * k + i] : (i == j ? tids[(size_t)b * (k + 1) + j] : -1);
    count(counters, j, L);
  }
  return SD_OK;
}

extern "C" int sd_argmax_rows(const float* logits, int64_t ld, int64_t* ids, int R, int V, sd_stream_t) {
  REQUIRE(logits && ids && R >= 0 && V >= 1, SD_ERR_INVALID, "sd_argmax_rows: bad arguments");
#pragma omp parallel for
  for (int r = 0; r < R; ++r) ids[r] = first_argmax(logits + (size_t)r * ld, V);
  return SD_OK;
}

extern "C" int sd_softmax_rows(const float* logits, int64_t ld, float temperature, float* probs,
                               int64_t ldp, int R, int V, sd_stream_t) {
  REQUIRE(logits && probs && R >= 0 && V >= 1 && temperature > 0.f, SD_ERR_INVALID,
          "sd_softmax_rows: bad arguments");
  const float inv_t = 1.0f / temperature;
#pragma omp parallel for
  for (int r = 0; r < R; ++r) {
    const float* row = logits + (size_t)r * ld;
    float* out = probs + (size_t)r * ldp;
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) mx = fmaxf(mx, row[i] * inv_t);
    double sum = 0.0;
    for (int i = 0; i < V; ++i) sum += (double)expf(row[i] * inv_t - mx);
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < V; ++i) out[i] = expf(row[i] * inv_t - mx) * inv;
  }
  return SD_OK;
}

extern "C" int sd_sample_rows(const float* probs, int64_t ld, const float* u, int64_t* ids, int R,
                              int V, sd_stream_t) {
  REQUIRE(probs && u && ids && R >= 0 && V >= 1, SD_ERR_INVALID, "sd_sample_rows: bad arguments");
#pragma omp parallel for
  for (int r = 0; r < R; ++r) ids[r] = sample_residual(probs + (size_t)r * ld, nullptr, V, u[r]);
  return SD_OK;
}

extern "C" int sd_add_logit_bias(float* logits, int64_t ld, const float* bias, int64_t ld_bias,
                                 const int32_t* enabled, int rows_per_seq, int R, int V, sd_stream_t) {
  REQUIRE(logits && bias && enabled && R >= 0 && V >= 1 && ld >= V && ld_bias >= V && rows_per_seq >= 1,
          SD_ERR_INVALID, "sd_add_logit_bias: bad arguments");
#pragma omp parallel for
  for (int r = 0; r < R; ++r) {
    const int seq = r / rows_per_seq;
    if (!enabled[seq]) continue;
    float* row = logits + (size_t)r * ld;
    const float* add = bias + (size_t)seq * ld_bias;
    for (int i = 0; i < V; ++i) row[i] += add[i];
  }
  return SD_OK;
}

extern "C" int sd_token_logprobs(const float* logits, int64_t ld, const int32_t* tokens, int R, int V,
                                 int n_top, int32_t* out_ids, float* out_lp, int32_t* out_ranks,
                                 sd_stream_t) {
  REQUIRE(logits && tokens && out_ids && out_lp && out_ranks && R >= 0 && V >= 1 && ld >= V &&
              n_top >= 0 && n_top <= SD_LOGPROBS_TOP_MAX,
          SD_ERR_INVALID, "sd_token_logprobs: bad arguments");
#pragma omp parallel for
  for (int r = 0; r < R; ++r) {
    const int tok = tokens[r];
    if (tok < 0 || tok >= V) continue;
    const float* row = logits + (size_t)r * ld;
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) mx = fmaxf(mx, row[i]);
    const float chosen = row[tok];
    double sum = 0.0;
    int ge = 0;
    for (int i = 0; i < V; ++i) {
      sum += (double)expf(row[i] - mx);
      ge += row[i] >= chosen;
    }
    const float lse = mx + (float)log(sum);
    const size_t o = (size_t)r * (n_top + 1);
    out_ids[o] = tok;
    out_lp[o] = chosen - lse;
    out_ranks[r] = ge;
    float pv = INFINITY;
    int pi = -1;
    for (int j = 0; j < n_top; ++j) {   // successive maxima in (value desc, index asc) order
      float best = -INFINITY;
      int bi = -1;
      for (int i = 0; i < V; ++i) {
        const float v = row[i];
        if ((v < pv || (v == pv && i > pi)) && v > best) {
          best = v;
          bi = i;
        }
      }
      const bool none = bi < 0;
      out_ids[o + 1 + j] = none ? -1 : bi;
      out_lp[o + 1 + j] = none ? -INFINITY : best - lse;
      pv = none ? -INFINITY : best;
      pi = none ? V : bi;
    }
  }
  return SD_OK;
}

extern "C" size_t sd_sampling_scratch_bytes(int R) { return R > 0 ? 64 : 0; }
extern "C" int sd_filter_probs_seq(float* probs, int64_t ld, const int32_t* top_k, const float* top_p,
                                   const float* min_p, int rows_per_seq, int R, int V, sd_stream_t) {
  REQUIRE(probs && R >= 0 && V >= 1 && ld >= V && rows_per_seq >= 1, SD_ERR_INVALID,
          "sd_filter_probs_seq: bad arguments");
  std::vector<float> sorted(V);
  for (int r = 0; r < R; ++r) {
    const int s = r / rows_per_seq;
    const int k = top_k ? top_k[s] : 0;
    const float P = top_p ? top_p[s] : 1.f, mp = min_p ? min_p[s] : 0.f;
    const bool use_k = k > 0 && k < V, use_p = P < 1.f, use_m = mp > 0.f;
    if (!use_k && !use_p && !use_m) continue;
    float* row = probs + (size_t)r * ld;
    std::copy(row, row + V, sorted.begin());
    std::sort(sorted.begin(), sorted.end(), std::greater<float>());
    if (!(sorted[0] > 0.f)) continue;
    float tau = 0.f;
    if (use_k) tau = sorted[k - 1];
    if (use_m) tau = std::max(tau, mp * sorted[0]);
    tau = std::max(tau, std::numeric_limits<float>::denorm_min());
    if (use_p) {
      double base = 0.0;
      for (int i = 0; i < V && sorted[i] >= tau; ++i) base += sorted[i];
      const double target = (double)P * base;
      double acc = 0.0;
      float cut = tau;
      for (int i = 0; i < V && sorted[i] >= tau; ++i) {   // distinct values, largest first
        int j = i;
        double group = 0.0;
        while (j < V && sorted[j] == sorted[i]) group += sorted[j++];
        acc += group;
        if (acc >= target) {
          cut = sorted[i];
          break;
        }
        i = j - 1;
      }
      tau = std::max(tau, cut);
    }
    double kept = 0.0;
    for (int i = 0; i < V; ++i) kept += row[i] >= tau ? row[i] : 0.f;
    const float inv = kept > 0.0 ? (float)(1.0 / kept) : 0.f;
    for (int i = 0; i < V; ++i) row[i] = row[i] >= tau ? row[i] * inv : 0.f;
  }
  return SD_OK;
}

extern "C" int sd_softmax_rows_seq(const float* logits, int64_t ld, const float* temps, int rows_per_seq,
                                   float* probs, int64_t ldp, int R, int V, void*, sd_stream_t s) {
  REQUIRE(logits && probs && temps && R >= 0 && V >= 1 && rows_per_seq >= 1, SD_ERR_INVALID,
          "sd_softmax_rows_seq: bad arguments");
  for (int r = 0; r < R; ++r) {
    const float T = temps[r / rows_per_seq];
    if (T > 0.f) {
      const int rc = sd_softmax_rows(logits + (size_t)r * ld, ld, T, probs + (size_t)r * ldp, ldp, 1, V, s);
      if (rc) return rc;
    } else {
      const int best = first_argmax(logits + (size_t)r * ld, V);
      for (int i = 0; i < V; ++i) probs[(size_t)r * ldp + i] = i == best ? 1.f : 0.f;
    }
  }
  return SD_OK;
}
extern "C" int sd_softmax_rows_split(const float* l, int64_t ld, float t, float* p, int64_t ldp, int R,
                                     int V, void*, sd_stream_t s) {
  return sd_softmax_rows(l, ld, t, p, ldp, R, V, s);
}
extern "C" int sd_sample_rows_split(const float* p, int64_t ld, const float* u, int64_t* ids, int R,
                                    int V, void*, sd_stream_t s) {
  return sd_sample_rows(p, ld, u, ids, R, V, s);
}

namespace {
inline float uniform24(unsigned long long seed, unsigned long long c, unsigned long long i) {
  unsigned long long z = seed ^ (c * 0x9e3779b97f4a7c15ull) ^ (i << 32 | (unsigned)i);
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
}  // namespace

extern "C" int sd_fill_uniforms_seq(float* out, const sd_uniform_region* regions, int n_regions,
                                    const uint64_t* seeds, uint64_t* counters, int B, sd_stream_t) {
  REQUIRE(out && regions && seeds && counters && B >= 0 && n_regions >= 1 &&
              n_regions <= SD_UNIFORM_REGIONS_MAX, SD_ERR_INVALID, "sd_fill_uniforms_seq: bad arguments");
  for (int b = 0; b < B; ++b) {
    int j0 = 0;
    for (int r = 0; r < n_regions; ++r) {
      const sd_uniform_region& g = regions[r];
      REQUIRE(g.n >= 0 && g.offset >= 0, SD_ERR_INVALID, "sd_fill_uniforms_seq: bad region");
      for (int i = 0; i < g.n; ++i)
        out[g.offset + (int64_t)i * g.stride_j + (int64_t)b * g.stride_b] =
            uniform24(seeds[b], counters[b], (unsigned long long)(j0 + i));
      j0 += g.n;
    }
    counters[b] += 1;
  }
  return SD_OK;
}

extern "C" int sd_fill_uniforms(float* out, int n, uint64_t seed, uint64_t* counter, sd_stream_t) {
  REQUIRE(out && counter && n >= 0, SD_ERR_INVALID, "sd_fill_uniforms: bad arguments");
  const unsigned long long c = *counter;
  for (int i = 0; i < n; ++i) {
    unsigned long long z = seed ^ (c * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)i << 32 | (unsigned)i);
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    out[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
  }
  *counter = c + 1;
  return SD_OK;
}

// ------------------------------------------------------------------ cache and attention
extern "C" int sd_rope_append(void* qkv, const float* cs, int max_pos, void* kc, void* vc,
                              const int32_t* bt, int max_blocks, const int32_t* seq_lens,
                              int pos_offset, int B, int q_len, int Hq, int Hkv, int D, sd_stream_t) {
  REQUIRE(qkv && kc && vc && bt && seq_lens && D % 16 == 0, SD_ERR_INVALID, "sd_rope_append: bad arguments");
  const int half = D / 2;
  for (int row = 0; row < B * q_len; ++row) {
    const int b = row / q_len, t = row % q_len;
    const int pos = seq_lens[b] + pos_offset + t;
    if (pos < 0) continue;
    const int page = bt[(size_t)b * max_blocks + (pos >> 4)], slot = pos & 15;
    bf16* x = (bf16*)qkv + (size_t)row * (Hq + 2 * Hkv) * D;
    const float* c = cs ? cs + (size_t)std::min(pos, max_pos - 1) * D : nullptr;
    for (int head = 0; head < Hq + 2 * Hkv; ++head)
      for (int i = 0; i < half; ++i) {
        bf16* hp = x + (size_t)head * D;
        float lo = bf2f(hp[i]), hi = bf2f(hp[i + half]);
        if (head < Hq + Hkv && c) {
          const float nlo = lo * c[i] - hi * c[half + i], nhi = hi * c[i] + lo * c[half + i];
          lo = nlo;
          hi = nhi;
        }
        if (head < Hq) {
          hp[i] = f2bf(lo);
          hp[i + half] = f2bf(hi);
        } else if (head < Hq + Hkv) {
          bf16* kp = (bf16*)kc + ((size_t)page * Hkv + (head - Hq)) * D * 16;
          const int i2 = i + half;
          kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = f2bf(lo);
          kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = f2bf(hi);
        } else {
          bf16* vp = (bf16*)vc + ((size_t)page * Hkv + (head - Hq - Hkv)) * D * 16;
          vp[i * 16 + v_slot(slot)] = f2bf(lo);
          vp[(i + half) * 16 + v_slot(slot)] = f2bf(hi);
        }
      }
  }
  return SD_OK;
}

extern "C" size_t sd_paged_attention_workspace_bytes(int, int, int, int, int, int) { return 16; }

extern "C" int sd_paged_attention_blocked(const void* q, int64_t ldq, const void* kc, const void* vc,
                                          const int32_t* bt, int max_blocks, const int32_t* seq_lens,
                                          int pos_offset, void* out, int64_t ldo, int B, int q_total,
                                          int q_block, int Hq, int Hkv, int D, float scale, int max_ctx,
                                          int split_tokens, void*, size_t, sd_stream_t) {
  (void)q_block; (void)split_tokens;
  REQUIRE(q && kc && vc && bt && seq_lens && out, SD_ERR_INVALID,
          "sd_paged_attention: null pointer (the CPU build has no partials-only form)");
  REQUIRE(B >= 0 && q_total >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && D >= 8 && D % 8 == 0,
          SD_ERR_INVALID, "sd_paged_attention: bad shape");
  REQUIRE(max_ctx >= 1 && max_blocks * 16 >= max_ctx, SD_ERR_INVALID,
          "sd_paged_attention: max_ctx %d exceeds block table (%d pages)", max_ctx, max_blocks);
  attention_f32((const bf16*)q, ldq, (const bf16*)kc, (const bf16*)vc, bt, max_blocks, seq_lens,
                pos_offset, (bf16*)out, ldo, B, q_total, Hq, Hkv, D, scale);
  return SD_OK;
}
extern "C" int sd_paged_attention_tree(const void* q, int64_t ldq, const void* kc, const void* vc,
                                       const int32_t* bt, int max_blocks, const int32_t* seq_lens,
                                       int pos_offset, const uint64_t* tree_mask, void* out, int64_t ldo,
                                       int B, int q_len, int Hq, int Hkv, int D, float scale, int max_ctx,
                                       int, void*, size_t, sd_stream_t) {
  REQUIRE(q && kc && vc && bt && seq_lens && out && tree_mask, SD_ERR_INVALID,
          "sd_paged_attention_tree: null pointer (the CPU build has no partials-only form)");
  REQUIRE(B >= 0 && q_len >= 1 && q_len <= 64 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && D % 8 == 0,
          SD_ERR_INVALID, "sd_paged_attention_tree: bad shape");
  REQUIRE(max_ctx >= 1 && max_blocks * 16 >= max_ctx, SD_ERR_INVALID,
          "sd_paged_attention_tree: max_ctx %d exceeds block table (%d pages)", max_ctx, max_blocks);
  attention_f32((const bf16*)q, ldq, (const bf16*)kc, (const bf16*)vc, bt, max_blocks, seq_lens,
                pos_offset, (bf16*)out, ldo, B, q_len, Hq, Hkv, D, scale, tree_mask);
  return SD_OK;
}

extern "C" int sd_map_tokens(int32_t* tokens, int64_t stride, int col, const int32_t* map, int n_map, int B,
                             sd_stream_t) {
  REQUIRE(tokens && map && n_map >= 1 && B >= 0 && col >= 0 && col < stride, SD_ERR_INVALID,
          "sd_map_tokens: bad arguments");
  for (int b = 0; b < B; ++b) {
    int32_t* t = tokens + (size_t)b * stride + col;
    *t = map[std::min(std::max(*t, 0), n_map - 1)];
  }
  return SD_OK;
}

extern "C" int sd_tree_prepare(const int32_t* tok_buf, int64_t tok_stride, int tok_col, int32_t* tree_tok,
                               const int32_t* tree_depth, const int32_t* seq_lens, int pos_offset,
                               int32_t* rope_pos, int B, int q_len, sd_stream_t) {
  REQUIRE(tok_buf && tree_tok && tree_depth && seq_lens && rope_pos && B >= 0 && q_len >= 1 && q_len <= 64 &&
              tok_col >= 0 && tok_col < tok_stride, SD_ERR_INVALID, "sd_tree_prepare: bad arguments");
  for (int b = 0; b < B; ++b) {
    tree_tok[(size_t)b * q_len] = tok_buf[(size_t)b * tok_stride + tok_col];
    for (int t = 0; t < q_len; ++t)
      rope_pos[(size_t)b * q_len + t] = seq_lens[b] + pos_offset + tree_depth[(size_t)b * q_len + t];
  }
  return SD_OK;
}

extern "C" int sd_tree_accept_sample(const float* probs, int64_t ld, const int32_t* tree_tok,
                                     const int32_t* tree_parent, const int32_t* tree_len, int q_len, int k,
                                     const float* u_accept, const float* u_final, int64_t* out_ids,
                                     int32_t* path, int32_t* n_path, int64_t* counters, int B, int V,
                                     sd_stream_t) {
  REQUIRE(probs && tree_tok && tree_parent && tree_len && u_accept && u_final && out_ids && path &&
              n_path && B >= 0 && q_len >= 1 && q_len <= 64 && k >= 1 && k < 64 && V >= 1 && ld >= V,
          SD_ERR_INVALID, "sd_tree_accept_sample: bad arguments");
  for (int b = 0; b < B; ++b) {
    const int len = std::min(std::max(tree_len[b], 1), q_len);
    const int32_t* tok = tree_tok + (size_t)b * q_len;
    const int32_t* par = tree_parent + (size_t)b * q_len;
    const float* ua = u_accept + (size_t)b * q_len;
    int depth[64] = {0}, max_depth = 0;
    for (int i = 1; i < len; ++i) {
      depth[i] = depth[std::min(std::max(par[i], 0), i - 1)] + 1;
      max_depth = std::max(max_depth, depth[i]);
    }
    int cur = 0, j = 0, n_excl = 0, excl[64];
    for (;;) {
      const float* row = probs + ((size_t)b * q_len + cur) * ld;
      double removed = 0.0;
      n_excl = 0;
      int next = -1;
      if (j < k) {
        for (int c = 1; c < len; ++c) {
          if (par[c] != cur) continue;
          const int x = tok[c];
          if (x < 0 || x >= V) continue;
          bool dup = false;   // a rejected sibling carried this token: its mass is gone
          for (int e = 0; e < n_excl; ++e) dup |= excl[e] == x;
          if (dup) continue;
          const double px = (double)row[x];
          if ((double)ua[c] * (1.0 - removed) < px) {
            next = c;
            break;
          }
          removed += px;
          excl[n_excl++] = x;
        }
      }
      if (next < 0) break;
      out_ids[(size_t)b * (k + 1) + j] = tok[next];
      path[(size_t)b * k + j] = next;
      cur = next;
      ++j;
    }
    out_ids[(size_t)b * (k + 1) + j] =
        sample_residual(probs + ((size_t)b * q_len + cur) * ld, nullptr, V, u_final[b], excl, n_excl);
    for (int i = j + 1; i <= k; ++i) out_ids[(size_t)b * (k + 1) + i] = -1;
    for (int i = j; i < k; ++i) path[(size_t)b * k + i] = -1;
    n_path[b] = j;
    if (counters) {
      counters[0] += j;
      counters[1] += j + 1;
      counters[2] += std::min(max_depth, k);
    }
  }
  return SD_OK;
}

extern "C" int sd_tree_accept_greedy(const int64_t* target_ids, const int32_t* tree_tok,
                                     const int32_t* tree_parent, const int32_t* tree_len, int q_len, int k,
                                     int64_t* out_ids, int32_t* path, int32_t* n_path, int64_t* counters,
                                     int B, sd_stream_t) {
  REQUIRE(target_ids && tree_tok && tree_parent && tree_len && out_ids && path && n_path && B >= 0 &&
              q_len >= 1 && q_len <= 64 && k >= 1 && k < 64, SD_ERR_INVALID,
          "sd_tree_accept_greedy: bad arguments");
  for (int b = 0; b < B; ++b) {
    const int len = std::min(std::max(tree_len[b], 1), q_len);
    const int32_t* tok = tree_tok + (size_t)b * q_len;
    const int32_t* par = tree_parent + (size_t)b * q_len;
    int depth[64] = {0}, max_depth = 0;
    for (int i = 1; i < len; ++i) {
      depth[i] = depth[std::min(std::max(par[i], 0), i - 1)] + 1;
      max_depth = std::max(max_depth, depth[i]);
    }
    int cur = 0, j = 0;
    for (; j < k; ++j) {
      const int64_t want = target_ids[(size_t)b * q_len + cur];
      int child = -1;
      for (int i = 1; i < len && child < 0; ++i)
        if (par[i] == cur && (int64_t)tok[i] == want) child = i;
      if (child < 0) break;
      out_ids[(size_t)b * (k + 1) + j] = want;
      path[(size_t)b * k + j] = child;
      cur = child;
    }
    out_ids[(size_t)b * (k + 1) + j] = target_ids[(size_t)b * q_len + cur];
    for (int i = j + 1; i <= k; ++i) out_ids[(size_t)b * (k + 1) + i] = -1;
    for (int i = j; i < k; ++i) path[(size_t)b * k + i] = -1;
    n_path[b] = j;
    if (counters) {
      counters[0] += j;
      counters[1] += j + 1;
      counters[2] += std::min(max_depth, k);
    }
  }
  return SD_OK;
}

extern "C" int sd_kv_compact(void* k_cache, void* v_cache, int64_t layer_stride_bytes, int n_layers,
                             const int32_t* bt, int max_blocks, const int32_t* seq_lens, int pos_offset,
                             const int32_t* path, const int32_t* n_path, int k, int B, int Hkv, int D,
                             sd_stream_t) {
  REQUIRE(k_cache && v_cache && bt && seq_lens && path && n_path && B >= 0 && n_layers >= 1 && Hkv >= 1 &&
              k >= 1 && D % 8 == 0, SD_ERR_INVALID, "sd_kv_compact: bad arguments");
  for (int li = 0; li < n_layers; ++li) {
    bf16* kc = (bf16*)((char*)k_cache + (size_t)li * layer_stride_bytes);
    bf16* vc = (bf16*)((char*)v_cache + (size_t)li * layer_stride_bytes);
    for (int b = 0; b < B; ++b) {
      const int n = std::min(std::max(n_path[b], 0), k), pos0 = seq_lens[b] + pos_offset;
      for (int j = 0; j < n; ++j) {
        const int src = pos0 + path[(size_t)b * k + j], dst = pos0 + j + 1;
        if (src == dst) continue;
        for (int h = 0; h < Hkv; ++h) {
          const size_t sp = ((size_t)bt[(size_t)b * max_blocks + (src >> 4)] * Hkv + h) * D * 16;
          const size_t dp = ((size_t)bt[(size_t)b * max_blocks + (dst >> 4)] * Hkv + h) * D * 16;
          for (int d = 0; d < D; ++d) {
            kc[dp + ((d >> 3) * 16 + (dst & 15)) * 8 + (d & 7)] = kc[sp + ((d >> 3) * 16 + (src & 15)) * 8 + (d & 7)];
            vc[dp + d * 16 + v_slot(dst & 15)] = vc[sp + d * 16 + v_slot(src & 15)];
          }
        }
      }
    }
  }
  return SD_OK;
}

extern "C" int sd_paged_attention(const void* q, int64_t ldq, const void* kc, const void* vc,
                                  const int32_t* bt, int max_blocks, const int32_t* seq_lens,
                                  int pos_offset, void* out, int64_t ldo, int B, int q_len, int Hq,
                                  int Hkv, int D, float scale, int max_ctx, int split_tokens, void* ws,
                                  size_t wsb, sd_stream_t s) {
  return sd_paged_attention_blocked(q, ldq, kc, vc, bt, max_blocks, seq_lens, pos_offset, out, ldo, B,
                                    q_len, q_len, Hq, Hkv, D, scale, max_ctx, split_tokens, ws, wsb, s);
}
extern "C" int sd_paged_attention_static(const void* q, int64_t ldq, const void* kc, const void* vc,
                                         const int32_t* bt, int max_blocks, const int32_t* seq_lens,
                                         int pos_offset, void* out, int64_t ldo, int B, int q_len,
                                         int Hq, int Hkv, int D, float scale, int max_ctx,
                                         int split_tokens, sd_stream_t s) {
  return sd_paged_attention_blocked(q, ldq, kc, vc, bt, max_blocks, seq_lens, pos_offset, out, ldo, B,
                                    q_len, q_len, Hq, Hkv, D, scale, max_ctx, split_tokens, nullptr, 0, s);
}

// ------------------------------------------------------------------ dense ops
extern "C" size_t sd_tiled_weight_elems(int N, int K, int mode) {
  if (N < 1 || K < 32 || K % 32) return 0;
  const int tiles = mode == SD_TILE_GATED ? (N + 7) / 8 : (N + 15) / 16;
  return (size_t)tiles * 16 * K;
}

extern "C" int sd_tile_weights(const void* src, void* dst, int N, int K, int mode, int param,
                               sd_stream_t) {
  REQUIRE(src && dst && N >= 1 && K >= 32 && K % 32 == 0, SD_ERR_INVALID,
          "sd_tile_weights: bad arguments (N=%d K=%d)", N, K);
  int tiles, rows_src, p0 = 0;
  if (mode == SD_TILE_PLAIN) {
    tiles = (N + 15) / 16;
    rows_src = N;
  } else if (mode == SD_TILE_GATED) {
    tiles = (N + 7) / 8;
    rows_src = 2 * N;
    p0 = N;
  } else if (mode == SD_TILE_ROTARY) {
    REQUIRE(param >= 16 && param % 16 == 0 && N % param == 0, SD_ERR_INVALID,
            "sd_tile_weights: rotary mode needs head_dim %% 16 == 0 and N %% head_dim == 0");
    tiles = N / 16;
    rows_src = N;
    p0 = param;
  } else {
    set_error("sd_tile_weights: unknown mode %d", mode);
    return SD_ERR_INVALID;
  }
  const int KS = K / 32;
  const bf16* s = (const bf16*)src;
  bf16* d = (bf16*)dst;
#pragma omp parallel for
  for (int nt = 0; nt < tiles; ++nt)
    for (int ks = 0; ks < KS; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int row = tile_src_row(mode, p0, rows_src, nt, l & 15);
        bf16* o = d + (((size_t)nt * KS + ks) * 64 + l) * 8;
        for (int j = 0; j < 8; ++j)
          o[j] = row < rows_src ? s[(size_t)row * K + ks * 32 + (l >> 4) * 8 + j] : (bf16)0;
      }
  return SD_OK;
}

extern "C" int sd_tile_weights_fp8(const void*, void*, int, int, int, int, sd_stream_t) {
  UNSUPPORTED("sd_tile_weights_fp8");
}

namespace {
int linear_any(const void* W, const float* w_scale, const void* x, int64_t ldx, const void* norm_w,
               const void* norm_b, float eps, int layer_norm, const void* bias, const void* residual,
               void* out, int64_t ldo, int M, int N, int K, int epi, const char* name) {
  REQUIRE(!w_scale, SD_ERR_UNSUPPORTED, "%s: fp8 weights are not provided by the CPU build", name);
  REQUIRE(W && x && out && M >= 0 && M <= 4096 && N >= 1 && K >= 32 && K % 32 == 0 && ldo >= N,
          SD_ERR_INVALID, "%s: bad arguments (M=%d N=%d K=%d)", name, M, N, K);
  REQUIRE(epi >= SD_EPI_STORE && epi <= SD_EPI_GELU_TANH, SD_ERR_INVALID, "%s: unknown epilogue %d", name, epi);
  REQUIRE(epi != SD_EPI_RESIDUAL || residual, SD_ERR_INVALID, "%s: residual epilogue without residual", name);
  if (M == 0) return SD_OK;
  std::vector<float> xf = rows_f32((const bf16*)x, ldx, M, K);
  if (norm_w) norm_rows(xf, M, K, (const bf16*)norm_w, (const bf16*)norm_b, eps, layer_norm != 0);
  return linear_f32((const bf16*)W, xf.data(), M, N, K, (const bf16*)bias, (const bf16*)residual, out,
                    ldo, epi);
}
}  // namespace

extern "C" int sd_linear(const void* W, const float* ws, const void* x, int64_t ldx, const void* bias,
                         const void* residual, void* out, int64_t ldo, int M, int N, int K, int epi,
                         sd_stream_t) {
  return linear_any(W, ws, x, ldx, nullptr, nullptr, 0.f, 0, bias, residual, out, ldo, M, N, K, epi, "sd_linear");
}
// fp8 activations exist for the GPU's fp8 matrix instruction only
extern "C" int sd_row_scales_fp8(const void*, int64_t, int, int, float*, sd_stream_t) {
  set_error("sd_row_scales_fp8: not available in the CPU build");
  return SD_ERR_UNSUPPORTED;
}
extern "C" int sd_linear_wide_a8(const void*, const float*, const float*, const void*, int64_t, const void*,
                                 const void*, void*, int64_t, int, int, int, int, void*, size_t, sd_stream_t) {
  set_error("sd_linear_wide_a8: not available in the CPU build");
  return SD_ERR_UNSUPPORTED;
}
extern "C" int sd_qkv_rope_wide_a8(const void*, const float*, const float*, const void*, int64_t, const void*,
                                   void*, int64_t, const float*, int, void*, void*, const int32_t*, int,
                                   const int32_t*, int, int, int, int, int, int, int,
                                   const sd_rope_positions*, void*, size_t, sd_stream_t) {
  set_error("sd_qkv_rope_wide_a8: not available in the CPU build");
  return SD_ERR_UNSUPPORTED;
}
extern "C" size_t sd_linear_wide_workspace_bytes(int, int, int, int) { return 16; }
extern "C" int sd_linear_wide(const void* W, const float* ws, const void* x, int64_t ldx,
                              const void* bias, const void* residual, void* out, int64_t ldo, int M,
                              int N, int K, int epi, void*, size_t, sd_stream_t) {
  return linear_any(W, ws, x, ldx, nullptr, nullptr, 0.f, 0, bias, residual, out, ldo, M, N, K, epi,
                    "sd_linear_wide");
}
extern "C" int sd_norm_linear(const void* W, const float* ws, const void* x, int64_t ldx,
                              const void* norm_w, const void* norm_b, float eps, int layer_norm,
                              const void* bias, void* out, int64_t ldo, int M, int N, int K, int epi,
                              sd_stream_t) {
  REQUIRE(norm_w, SD_ERR_INVALID, "sd_norm_linear: fused norm without weight");
  return linear_any(W, ws, x, ldx, norm_w, norm_b, eps, layer_norm, bias, nullptr, out, ldo, M, N, K, epi,
                    "sd_norm_linear");
}
extern "C" int sd_norm_linear_argmax(const void* W, const float* ws, const void* x, int64_t ldx,
                                     const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                     float* part_val, int32_t* part_idx, int M, int N, int K,
                                     sd_stream_t st) {
  REQUIRE(part_val && part_idx, SD_ERR_INVALID, "sd_norm_linear_argmax: null pointer");
  std::vector<float> logits((size_t)std::max(M, 0) * N);
  const int rc = sd_norm_linear(W, ws, x, ldx, norm_w, norm_b, eps, layer_norm, nullptr, logits.data(), N, M,
                                N, K, SD_EPI_F32, st);
  if (rc) return rc;
  return sd_argmax_partial(logits.data(), N, part_val, part_idx, M, N, st);
}
extern "C" int sd_linear_attn_merge(const void*, const float*, const void*, size_t, const int32_t*, int,
                                    int, int, int, int, int, int, int, const void*, const void*, void*,
                                    int64_t, int, int, sd_stream_t) {
  UNSUPPORTED("sd_linear_attn_merge");
}

// The fused two-layer launches exist to hide GPU launch boundaries; the CPU build has none
// to hide: sd_model_forward ignores chain_state and the entry points are not provided.
extern "C" size_t sd_chain_state_bytes(int hidden) {
  return hidden < 1 ? 0 : 256 + (size_t)SD_CHAIN_MAX_ROWS * hidden * 8;
}
extern "C" int sd_chain_state_init(void*, size_t, int, sd_stream_t) { UNSUPPORTED("sd_chain_state_init"); }
extern "C" int sd_chain_errors(const void*, int*) { UNSUPPORTED("sd_chain_errors"); }
extern "C" int sd_chain_supported(int, int, int, int) { return 0; }
extern "C" int sd_chain_o_up(void*, const void*, const float*, const void*, const void*, int64_t,
                             const void*, size_t, const int32_t*, int, int, int, int, int, int, int, int,
                             void*, int, const void*, float, const void*, const float*, const void*,
                             void*, int64_t, int, sd_stream_t) {
  UNSUPPORTED("sd_chain_o_up");
}
extern "C" int sd_chain_down_qkv(void*, const void*, const float*, const void*, const void*, int64_t, int,
                                 void*, int, const void*, float, const void*, const float*, const void*,
                                 void*, int64_t, const float*, int, void*, void*, const int32_t*, int,
                                 const int32_t*, int, int, int, int, int, int, sd_stream_t) {
  UNSUPPORTED("sd_chain_down_qkv");
}

extern "C" int sd_norm_qkv_rope(const void* W, const float* ws, const void* x, int64_t ldx,
                                const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                const void* bias, void* q_out, int64_t ldq, const float* rope_cs,
                                int max_pos, void* kc, void* vc, const int32_t* bt, int max_blocks,
                                const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                                int Hkv, int D, int K, const sd_rope_positions* rope_pos, sd_stream_t) {
  REQUIRE(!ws, SD_ERR_UNSUPPORTED, "sd_norm_qkv_rope: fp8 weights are not provided by the CPU build");
  REQUIRE(!rope_pos || (!rope_pos->delta && (!rope_pos->pos || rope_pos->section_t == D / 2)),
          SD_ERR_UNSUPPORTED,
          "sd_norm_qkv_rope: multimodal rotary sections are not provided by the CPU build");
  REQUIRE(W && x && q_out && kc && vc && bt && seq_lens && D % 16 == 0 && K % 32 == 0 &&
              ldq >= (int64_t)Hq * D, SD_ERR_INVALID, "sd_norm_qkv_rope: bad arguments");
  const int M = B * q_len;
  if (M == 0) return SD_OK;
  std::vector<float> xf = rows_f32((const bf16*)x, ldx, M, K);
  if (norm_w) norm_rows(xf, M, K, (const bf16*)norm_w, (const bf16*)norm_b, eps, layer_norm != 0);
  Rope rp{rope_cs, max_pos, (bf16*)kc, (bf16*)vc, bt, max_blocks, seq_lens, pos_offset, q_len, Hq, Hkv, D};
  if (rope_pos) rp.rot_pos = rope_pos->pos;
  return qkv_rope_f32((const bf16*)W, xf.data(), M, K, (const bf16*)bias, (bf16*)q_out, ldq, rp);
}
extern "C" int sd_qkv_rope_wide(const void* W, const float* ws, const void* x, int64_t ldx,
                                const void* bias, void* q_out, int64_t ldq, const float* rope_cs,
                                int max_pos, void* kc, void* vc, const int32_t* bt, int max_blocks,
                                const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                                int Hkv, int D, int K, const sd_rope_positions* rp, void*, size_t,
                                sd_stream_t s) {
  return sd_norm_qkv_rope(W, ws, x, ldx, nullptr, nullptr, 0.f, 0, bias, q_out, ldq, rope_cs, max_pos, kc,
                          vc, bt, max_blocks, seq_lens, pos_offset, B, q_len, Hq, Hkv, D, K, rp, s);
}

extern "C" int sd_rmsnorm(const void* x, int64_t ldx, const void* w, void* out, int64_t ldo, int rows,
                          int H, float eps, sd_stream_t) {
  REQUIRE(x && w && out && rows >= 0 && H >= 1, SD_ERR_INVALID, "sd_rmsnorm: bad arguments");
  std::vector<float> xf = rows_f32((const bf16*)x, ldx, rows, H);
  norm_rows(xf, rows, H, (const bf16*)w, nullptr, eps, false);
  for (int m = 0; m < rows; ++m)
    for (int c = 0; c < H; ++c) ((bf16*)out)[(size_t)m * ldo + c] = f2bf(xf[(size_t)m * H + c]);
  return SD_OK;
}
extern "C" int sd_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* out,
                            int64_t ldo, int rows, int H, float eps, sd_stream_t) {
  REQUIRE(x && w && out && rows >= 0 && H >= 1, SD_ERR_INVALID, "sd_layernorm: bad arguments");
  std::vector<float> xf = rows_f32((const bf16*)x, ldx, rows, H);
  norm_rows(xf, rows, H, (const bf16*)w, (const bf16*)b, eps, true);
  for (int m = 0; m < rows; ++m)
    for (int c = 0; c < H; ++c) ((bf16*)out)[(size_t)m * ldo + c] = f2bf(xf[(size_t)m * H + c]);
  return SD_OK;
}

extern "C" int sd_embed(const int32_t* ids, int64_t id_stride, int64_t id_offset, const void* table,
                        int tiled, const void* pos_table, int pos_bias, const int32_t* seq_lens,
                        int pos_offset, void* out, int B, int q_len, int H, int vocab, sd_stream_t) {
  REQUIRE(ids && table && out && B >= 0 && q_len >= 1 && H % 32 == 0 && vocab >= 1, SD_ERR_INVALID,
          "sd_embed: bad arguments");
  REQUIRE(!pos_table || seq_lens, SD_ERR_INVALID, "sd_embed: pos_table needs seq_lens");
  const bf16* tb = (const bf16*)table;
  const int KS = H / 32;
  for (int row = 0; row < B * q_len; ++row) {
    const int b = row / q_len, t = row % q_len;
    int id = ids[(size_t)b * id_stride + id_offset + t];
    id = std::min(std::max(id, 0), vocab - 1);
    const bf16* ps = nullptr;
    if (pos_table) {
      const int pos = std::max(seq_lens[b] + pos_offset + t, 0);
      ps = (const bf16*)pos_table + (size_t)(pos + pos_bias) * H;
    }
    bf16* dst = (bf16*)out + (size_t)row * H;
    for (int c = 0; c < H; ++c) {
      const bf16 v = tiled ? tb[(((size_t)(id >> 4) * KS + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (id & 15)) * 8 + (c & 7)]
                           : tb[(size_t)id * H + c];
      dst[c] = ps ? f2bf(bf2f(v) + bf2f(ps[c])) : v;
    }
  }
  return SD_OK;
}

// ------------------------------------------------------------------ decoder model
struct sd_model {
  sd_model_desc d;
  std::vector<sd_layer_weights> layers;
};

extern "C" sd_model* sd_model_create(const sd_model_desc* desc) {
  if (!desc || !desc->layers || desc->n_layers < 1) {
    set_error("sd_model_create: null descriptor or no layers");
    return nullptr;
  }
  const sd_model_desc& d = *desc;
  if ((d.arch != SD_ARCH_LLAMA && d.arch != SD_ARCH_OPT) || d.hidden % 32 || d.ffn % 32 ||
      d.head_dim % 16 || d.n_heads % d.n_kv_heads || !d.embed || !d.lm_head || !d.final_norm_w) {
    set_error("sd_model_create: unsupported shape");
    return nullptr;
  }
  if (d.weight_fp8 || d.allreduce || d.lm_vocab || d.mrope_section_t || d.mrope_section_h) {
    set_error("sd_model_create: fp8 weights, tensor parallelism, vocabulary-parallel heads and "
              "multimodal rotary sections are not provided by the CPU build");
    return nullptr;
  }
  if ((d.arch == SD_ARCH_LLAMA && !d.rope_cs) || (d.arch == SD_ARCH_OPT && !d.pos_table)) {
    set_error("sd_model_create: missing rope / position table");
    return nullptr;
  }
  sd_model* m = new sd_model;
  m->d = d;
  m->layers.assign(d.layers, d.layers + d.n_layers);
  m->d.layers = m->layers.data();
  return m;
}
extern "C" void sd_model_destroy(sd_model* m) { delete m; }

namespace {
inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
}
extern "C" size_t sd_model_workspace_bytes(const sd_model* m, int max_rows, int, int) {
  if (!m || max_rows < 1) return 0;
  const sd_model_desc& d = m->d;
  const size_t r = max_rows;
  return a256(r * d.hidden * 2) * 2 + a256(r * (d.n_heads + 2 * d.n_kv_heads) * d.head_dim * 2) +
         a256(r * d.n_heads * d.head_dim * 2) + a256(r * d.ffn * 2) + 256;
}

extern "C" int sd_model_forward(sd_model* m, const sd_kv_cache* kv, const sd_forward_args* a,
                                void* workspace, size_t workspace_bytes, sd_stream_t st) {
  REQUIRE(m && kv && a && workspace, SD_ERR_INVALID, "sd_model_forward: null argument");
  const sd_model_desc& d = m->d;
  const int rows = a->B * a->q_len;
  REQUIRE(a->B >= 1 && a->q_len >= 1 && rows <= 4096, SD_ERR_UNSUPPORTED, "sd_model_forward: %d rows", rows);
  const bool tree = a->tree_mask != nullptr;
  REQUIRE(!a->input_embeds && !a->rope_delta && (tree || !a->rope_pos), SD_ERR_UNSUPPORTED,
          "sd_model_forward: multimodal inputs are not provided by the CPU build");
  REQUIRE(!tree || (d.arch == SD_ARCH_LLAMA && a->rope_pos && a->q_len <= 64), SD_ERR_UNSUPPORTED,
          "sd_model_forward: a token tree needs a rotary model, rope_pos and at most 64 rows per sequence");
  REQUIRE(a->token_ids && a->seq_lens && kv->k_cache && kv->v_cache && kv->block_tables, SD_ERR_INVALID,
          "sd_model_forward: null buffer");
  REQUIRE(workspace_bytes >= sd_model_workspace_bytes(m, rows, 0, 0), SD_ERR_WORKSPACE,
          "sd_model_forward: workspace too small");
  const int H = d.hidden, D = d.head_dim, Hq = d.n_heads, Hkv = d.n_kv_heads, F = d.ffn;
  const bool llama = d.arch == SD_ARCH_LLAMA;
  char* base = (char*)workspace;
  bf16* h = (bf16*)base;
  base += a256((size_t)rows * H * 2) * 2;
  bf16* q = (bf16*)base;
  base += a256((size_t)rows * (Hq + 2 * Hkv) * D * 2);
  bf16* attn = (bf16*)base;
  base += a256((size_t)rows * Hq * D * 2);
  bf16* act = (bf16*)base;
  int rc = sd_embed(a->token_ids, a->tok_stride, a->tok_offset, d.embed, d.embed_tiled, d.pos_table,
                    d.pos_bias, a->seq_lens, a->pos_offset, h, a->B, a->q_len, H, d.vocab, st);
  if (rc) return rc;
  const sd_rope_positions tree_pos{a->rope_pos, nullptr, D / 2, 0};
  for (int li = 0; li < d.n_layers; ++li) {
    const sd_layer_weights& l = m->layers[li];
    void* kc = (char*)kv->k_cache + (size_t)li * kv->layer_stride_bytes;
    void* vc = (char*)kv->v_cache + (size_t)li * kv->layer_stride_bytes;
    rc = sd_norm_qkv_rope(l.qkv_w, nullptr, h, H, l.attn_norm_w, l.attn_norm_b, d.norm_eps, !llama, l.qkv_b,
                          q, (int64_t)Hq * D, llama ? d.rope_cs : nullptr, d.max_pos, kc, vc,
                          kv->block_tables, kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, a->B,
                          a->q_len, Hq, Hkv, D, H, tree ? &tree_pos : nullptr, st);
    if (rc) return rc;
    attention_f32(q, (int64_t)Hq * D, (const bf16*)kc, (const bf16*)vc, kv->block_tables,
                  kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, attn, (int64_t)Hq * D, a->B,
                  a->q_len, Hq, Hkv, D, d.attn_scale, a->tree_mask);
    rc = sd_linear(l.o_w, nullptr, attn, (int64_t)Hq * D, l.o_b, h, h, H, rows, H, Hq * D, SD_EPI_RESIDUAL, st);
    if (rc) return rc;
    rc = sd_norm_linear(l.up_w, nullptr, h, H, l.mlp_norm_w, l.mlp_norm_b, d.norm_eps, !llama, l.up_b, act,
                        F, rows, F, H, llama ? SD_EPI_SILU_MUL : SD_EPI_RELU, st);
    if (rc) return rc;
    rc = sd_linear(l.down_w, nullptr, act, F, l.down_b, h, h, H, rows, H, F, SD_EPI_RESIDUAL, st);
    if (rc) return rc;
  }
  if (a->argmax_val && a->argmax_idx)   // the slice maxima of the rows' logits (logits optional)
    return sd_norm_linear_argmax(d.lm_head, nullptr, h, H, d.final_norm_w, d.final_norm_b, d.norm_eps,
                                 !llama, a->argmax_val, a->argmax_idx, rows, d.vocab, H, st);
  if (a->logits)
    return sd_norm_linear(d.lm_head, nullptr, h, H, d.final_norm_w, d.final_norm_b, d.norm_eps, !llama,
                          nullptr, a->logits, d.vocab, rows, d.vocab, H, SD_EPI_F32, st);
  return SD_OK;
}

// ------------------------------------------------------------------ tensor parallelism, graphs
extern "C" int sd_tp_unique_id(void*) { UNSUPPORTED("sd_tp_unique_id"); }
extern "C" sd_tp_comm* sd_tp_comm_create(const void*, int, int) {
  set_error("sd_tp_comm_create: not provided by the CPU build of the library");
  return nullptr;
}
extern "C" void sd_tp_comm_destroy(sd_tp_comm*) {}
extern "C" int sd_tp_allreduce(void*, size_t, void*, sd_stream_t) { UNSUPPORTED("sd_tp_allreduce"); }
extern "C" int sd_tp_allgather(const void*, void*, size_t, void*, sd_stream_t) { UNSUPPORTED("sd_tp_allgather"); }
extern "C" size_t sd_ipc_buffer_bytes(size_t) { return 0; }
extern "C" int sd_ipc_buffer_create(size_t, void**, void*) { UNSUPPORTED("sd_ipc_buffer_create"); }
extern "C" int sd_ipc_buffer_open(const void*, void**) { UNSUPPORTED("sd_ipc_buffer_open"); }
extern "C" int sd_ipc_buffer_close(void*) { UNSUPPORTED("sd_ipc_buffer_close"); }
extern "C" int sd_ipc_buffer_destroy(void*) { UNSUPPORTED("sd_ipc_buffer_destroy"); }
extern "C" sd_ipc_comm* sd_ipc_comm_create(int, int, void* const*, size_t) {
  set_error("sd_ipc_comm_create: not provided by the CPU build of the library");
  return nullptr;
}
extern "C" void sd_ipc_comm_destroy(sd_ipc_comm*) {}
// the draft model's three-launch layer is a GPU-side launch optimisation: the CPU build runs
// the ordinary layer (sd_forward_args.draft_fused is ignored) and refuses the kernels
extern "C" int sd_draft_layer_supported(int, int, int, int, int, int, int, int) { return 0; }
extern "C" int sd_draft_attn_oproj(const void*, int64_t, const void*, const void*, const int32_t*, int,
                                   const int32_t*, int, const void*, float*, float*, int, int, int, int,
                                   int, int, int, float, int, sd_stream_t) {
  UNSUPPORTED("sd_draft_attn_oproj");
}
extern "C" int sd_draft_mlp(const void*, int64_t, const float*, int, void*, const void*, float, const void*,
                            const void*, float*, int, int, int, sd_stream_t) {
  UNSUPPORTED("sd_draft_mlp");
}
extern "C" int sd_norm_linear_adds(const void* W, const float* ws, const void* x, int64_t ldx,
                                   const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                   const void* bias, void* out, int64_t ldo, int M, int N, int K,
                                   int epilogue, const sd_row_adds* adds, sd_stream_t stream) {
  if (adds && adds->add) UNSUPPORTED("sd_norm_linear_adds");
  return sd_norm_linear(W, ws, x, ldx, norm_w, norm_b, eps, layer_norm, bias, out, ldo, M, N, K, epilogue,
                        stream);
}
extern "C" int sd_norm_qkv_rope_adds(const void* W, const float* ws, const void* x, int64_t ldx,
                                     const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                     const void* bias, void* q_out, int64_t ldq, const float* rope_cs,
                                     int max_pos, void* k_cache, void* v_cache,
                                     const int32_t* block_tables, int max_blocks_per_seq,
                                     const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                                     int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                                     const sd_row_adds* adds, sd_stream_t stream) {
  if (adds && adds->add) UNSUPPORTED("sd_norm_qkv_rope_adds");
  return sd_norm_qkv_rope(W, ws, x, ldx, norm_w, norm_b, eps, layer_norm, bias, q_out, ldq, rope_cs,
                          max_pos, k_cache, v_cache, block_tables, max_blocks_per_seq, seq_lens,
                          pos_offset, B, q_len, Hq, Hkv, D, K, rope_pos, stream);
}
extern "C" int sd_ipc_comm_errors(sd_ipc_comm*, int*) { UNSUPPORTED("sd_ipc_comm_errors"); }
extern "C" int sd_ipc_comm_set_timeout_ms(sd_ipc_comm*, int) { UNSUPPORTED("sd_ipc_comm_set_timeout_ms"); }
extern "C" int sd_ipc_allreduce(void*, size_t, void*, sd_stream_t) { UNSUPPORTED("sd_ipc_allreduce"); }
extern "C" int sd_ipc_allgather(const void*, void*, size_t, void*, sd_stream_t) { UNSUPPORTED("sd_ipc_allgather"); }
extern "C" int sd_graph_begin(sd_stream_t) { UNSUPPORTED("sd_graph_begin"); }
extern "C" int sd_graph_end(sd_stream_t, sd_graph**) { UNSUPPORTED("sd_graph_end"); }
extern "C" int sd_graph_launch(sd_graph*, sd_stream_t) { UNSUPPORTED("sd_graph_launch"); }
extern "C" void sd_graph_destroy(sd_graph*) {}

// ------------------------------------------------------------------ step glue
extern "C" int sd_store_proposal(const int64_t* ids, int32_t* tok_buf, int k, int step, int B, sd_stream_t) {
  REQUIRE(ids && tok_buf && B >= 0 && k >= 1 && step >= 0 && step < k, SD_ERR_INVALID,
          "sd_store_proposal: bad arguments");
  for (int b = 0; b < B; ++b) tok_buf[(size_t)b * (k + 2) + 2 + step] = (int32_t)ids[b];
  return SD_OK;
}
extern "C" int sd_set_proposals(const int32_t* p, int32_t* tok_buf, int B, int k, sd_stream_t) {
  REQUIRE(p && tok_buf && B >= 0 && k >= 1, SD_ERR_INVALID, "sd_set_proposals: bad arguments");
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < k; ++j) tok_buf[(size_t)b * (k + 2) + 2 + j] = p[(size_t)b * k + j];
  return SD_OK;
}
extern "C" int sd_gather_proposals(const int32_t* tok_buf, int64_t* ids, int B, int k, sd_stream_t) {
  REQUIRE(tok_buf && ids && B >= 0 && k >= 1, SD_ERR_INVALID, "sd_gather_proposals: bad arguments");
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < k; ++j) ids[(size_t)b * k + j] = tok_buf[(size_t)b * (k + 2) + 2 + j];
  return SD_OK;
}
extern "C" int sd_advance_state(const int64_t* out_ids, int32_t* seq_lens, int32_t* tok_buf,
                                int32_t* gen_tokens, int64_t gen_stride, int gen_capacity,
                                int32_t* gen_counts, int32_t* step_out, int B, int k, sd_stream_t) {
  REQUIRE(out_ids && seq_lens && tok_buf && gen_counts && B >= 0 && k >= 0, SD_ERR_INVALID,
          "sd_advance_state: bad arguments");
  for (int b = 0; b < B; ++b)
    apply_emitted(out_ids + (size_t)b * (k + 1), b, seq_lens, tok_buf, gen_tokens, gen_stride,
                  gen_capacity, gen_counts, step_out, k);
  return SD_OK;
}

extern "C" int sd_topk_tokens(const float* logits, int64_t ld, int V, int row_offset, int row_step, int rows,
                              int b, int32_t* out_tok, int64_t out_stride, int out_col, int B, sd_stream_t) {
  REQUIRE(logits && out_tok && V >= 1 && ld >= V && row_offset >= 0 && row_step >= 1 && B >= 0 && b >= 1 &&
              rows >= 1 && rows <= row_step && b <= SD_TOPK_TOKENS_MAX && b <= V && out_col >= 0 &&
              out_stride >= out_col + (int64_t)rows * b,
          SD_ERR_INVALID, "sd_topk_tokens: bad arguments");
#pragma omp parallel for
  for (int sr = 0; sr < B * rows; ++sr) {
    const int s = sr / rows, r = sr % rows;
    const float* row = logits + (size_t)(row_offset + s * row_step + r) * ld;
    int win[SD_TOPK_TOKENS_MAX];
    for (int c = 0; c < b; ++c) {
      float best = -INFINITY;
      int bi = -1;
      for (int i = 0; i < V; ++i) {
        bool taken = false;
        for (int e = 0; e < c; ++e) taken |= win[e] == i;
        if (!taken && row[i] > best) { best = row[i]; bi = i; }
      }
      if (bi < 0) {   // fewer than b finite entries: the free indices follow
        bi = 0;
        for (int e = 0; e < c; ++e) if (win[e] == bi) { ++bi; e = -1; }
      }
      win[c] = bi;
      out_tok[(size_t)s * out_stride + out_col + r * b + c] = bi;
    }
  }
  return SD_OK;
}
extern "C" int sd_argmax_partial(const float* logits, int64_t ld, float* pv, int32_t* pi, int R, int V,
                                 sd_stream_t) {
  REQUIRE(logits && pv && pi && R >= 0 && V >= 1, SD_ERR_INVALID, "sd_argmax_partial: bad arguments");
  const int chunk = ((V + SD_ARGMAX_PARTS - 1) / SD_ARGMAX_PARTS + 3) & ~3;
#pragma omp parallel for collapse(2)
  for (int r = 0; r < R; ++r)
    for (int part = 0; part < SD_ARGMAX_PARTS; ++part) {
      const int lo = part * chunk, hi = std::min(lo + chunk, V);
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int i = lo; i < hi; ++i)
        if (logits[(size_t)r * ld + i] > best) {
          best = logits[(size_t)r * ld + i];
          bi = i;
        }
      pv[(size_t)r * SD_ARGMAX_PARTS + part] = best;
      pi[(size_t)r * SD_ARGMAX_PARTS + part] = bi;
    }
  return SD_OK;
}
extern "C" int sd_argmax_combine(const float* vals, const int32_t* idx, int n_ranks, int R,
                                 int shard_vocab, int64_t rank_stride, float* out_val, int32_t* out_idx,
                                 sd_stream_t) {
  REQUIRE(vals && idx && out_val && out_idx && n_ranks >= 1 && R >= 0, SD_ERR_INVALID,
          "sd_argmax_combine: bad arguments");
  for (int i = 0; i < R * SD_ARGMAX_PARTS; ++i) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int r = 0; r < n_ranks; ++r) {
      const float v = vals[(size_t)r * rank_stride + i];
      int id = idx[(size_t)r * rank_stride + i];
      if (id != 0x7fffffff) id += r * shard_vocab;
      if (v > best || (v == best && id < bi)) {
        best = v;
        bi = id;
      }
    }
    out_val[i] = best;
    out_idx[i] = bi;
  }
  return SD_OK;
}
extern "C" int sd_argmax_from_partials(const float* pv, const int32_t* pi, int64_t* ids, int R, sd_stream_t) {
  REQUIRE(pv && pi && ids && R >= 0, SD_ERR_INVALID, "sd_argmax_from_partials: bad arguments");
  for (int r = 0; r < R; ++r) ids[r] = finish_partials(pv, pi, r);
  return SD_OK;
}
extern "C" int sd_propose_from_partials(const float* pv, const int32_t* pi, int row_offset, int row_step,
                                        int32_t* tok_buf, int k, int step, int B, sd_stream_t) {
  REQUIRE(pv && pi && tok_buf && B >= 0 && k >= 1 && step >= 0 && step < k && row_offset >= 0 &&
              row_step >= 1, SD_ERR_INVALID, "sd_propose_from_partials: bad arguments");
  for (int b = 0; b < B; ++b)
    tok_buf[(size_t)b * (k + 2) + 2 + step] = finish_partials(pv, pi, row_offset + b * row_step);
  return SD_OK;
}
extern "C" int sd_verify_from_partials(const float* pv, const int32_t* pi, int32_t* tok_buf,
                                       const int32_t* lens, int64_t* out_ids, int64_t* counters,
                                       int32_t* seq_lens, int32_t* gen_tokens, int64_t gen_stride,
                                       int gen_capacity, int32_t* gen_counts, int32_t* step_out, int B,
                                       int k, sd_stream_t) {
  REQUIRE(pv && pi && tok_buf && seq_lens && gen_counts && B >= 0 && k >= 0 && k <= 15, SD_ERR_INVALID,
          "sd_verify_from_partials: bad arguments (B=%d k=%d)", B, k);
  for (int b = 0; b < B; ++b) {
    int target[16];
    for (int i = 0; i <= k; ++i) target[i] = finish_partials(pv, pi, b * (k + 1) + i);
    const int L = std::min(std::max(lens ? lens[b] : k, 0), k);
    const int32_t* prop = tok_buf + (size_t)b * (k + 2) + 2;
    int j = 0;
    while (j < L && prop[j] == target[j]) ++j;
    int64_t o[16];
    for (int i = 0; i <= k; ++i) o[i] = i < j ? prop[i] : (i == j ? target[j] : -1);
    if (out_ids)
      for (int i = 0; i <= k; ++i) out_ids[(size_t)b * (k + 1) + i] = o[i];
    count(counters, j, L);
    apply_emitted(o, b, seq_lens, tok_buf, gen_tokens, gen_stride, gen_capacity, gen_counts, step_out, k);
  }
  return SD_OK;
}

extern "C" int sd_debug_force_proposals(int32_t* tok_buf, int k, const int32_t* seq_lens,
                                        const int32_t* truth, int64_t truth_stride,
                                        const uint8_t* pattern, int pattern_len, int32_t* step_counter,
                                        int vocab, int B, sd_stream_t) {
  REQUIRE(tok_buf && seq_lens && truth && pattern && step_counter && B >= 0 && k >= 1 && vocab >= 2 &&
              pattern_len >= 1 && pattern_len <= SD_DEBUG_PATTERN_MAX, SD_ERR_INVALID,
          "sd_debug_force_proposals: bad arguments");
  for (int b = 0; b < B; ++b) {
    const int t = step_counter[b]++;
    const int a = std::min((int)pattern[t % pattern_len], k), n = seq_lens[b];
    for (int i = 0; i <= a && i < k; ++i) {
      if (n + i >= truth_stride) break;
      const int32_t tok = truth[(size_t)b * truth_stride + n + i];
      tok_buf[(size_t)b * (k + 2) + 2 + i] = i < a ? tok : (tok + 1) % vocab;
    }
  }
  return SD_OK;
}
