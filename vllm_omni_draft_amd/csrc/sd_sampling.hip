// Acceptance and sampling kernels: rejection sampling over probability rows,
// greedy verification, row argmax / softmax / inverse-CDF sampling, and the
// per-sequence device state updates of a speculative step.
//
// The accept mask of a sequence is one wavefront ballot: lane i tests position
// i, lanes past the proposal length vote "accepted", and the first rejected
// position is the lowest clear bit of the 64-bit mask.
#include "sd_common.h"

namespace {

constexpr int kSampleThreads = 1024;
constexpr int kSampleWaves = kSampleThreads / SD_WAVE;

__device__ __forceinline__ double block_sum_f64(double v, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += smem[w];
  return t;
}

__device__ __forceinline__ int block_max_i32(int v, int* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  int t = smem[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = max(t, smem[w]);
  return t;
}

// First index of the row maximum (ties: smallest index), whole block.
__device__ int block_argmax(const float* __restrict__ row, int V, float* smem_v,
                            int* smem_i) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x * 4; i < V; i += blockDim.x * 4) {
    if (i + 3 < V) {
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e[j] > best) { best = e[j]; bi = i + j; }
    } else {
      for (int j = 0; j < 4 && i + j < V; ++j)
        if (row[i + j] > best) { best = row[i + j]; bi = i + j; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { smem_v[wave] = best; smem_i[wave] = bi; }
  __syncthreads();
  best = smem_v[0];
  bi = smem_i[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
    if (smem_v[w] > best || (smem_v[w] == best && smem_i[w] < bi)) {
      best = smem_v[w];
      bi = smem_i[w];
    }
  }
  if (bi == 0x7fffffff) bi = 0;  // all -inf / NaN
  return bi;
}

// Inverse-CDF draw from norm(max(p - q, 0)) (q == nullptr: from p).  Returns
// the first index whose f64 running sum exceeds u * total; if rounding leaves
// none, the last index with positive mass; if the mass is zero, argmax(p).
// Must be called by every thread of the block.
__device__ int block_sample_residual(const float* __restrict__ p,
                                     const float* __restrict__ q, int V, float u) {
  __shared__ double s_d[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_found;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = blockDim.x >> 6;

  double local = 0.0;
  int last_pos = -1;
  for (int i = tid * 4; i < V; i += blockDim.x * 4) {
    const float4 pv = *reinterpret_cast<const float4*>(p + i);
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q) qv = *reinterpret_cast<const float4*>(q + i);
    const float r[4] = {fmaxf(pv.x - qv.x, 0.f), fmaxf(pv.y - qv.y, 0.f),
                        fmaxf(pv.z - qv.z, 0.f), fmaxf(pv.w - qv.w, 0.f)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      local += (double)r[j];
      if (r[j] > 0.f) last_pos = i + j;
    }
  }
  const double total = block_sum_f64(local, s_d);
  last_pos = block_max_i32(last_pos, s_i);
  if (!(total > 0.0)) return block_argmax(p, V, s_f, s_i);

  const double thr = (double)u * total;
  if (tid == 0) s_found = 0x7fffffff;
  __syncthreads();
  double carry = 0.0;
  const int tile = blockDim.x * 4;
  for (int base = 0; base < V; base += tile) {
    const int i = base + tid * 4;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < V) {
      const float4 pv = *reinterpret_cast<const float4*>(p + i);
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q) qv = *reinterpret_cast<const float4*>(q + i);
      r[0] = fmaxf(pv.x - qv.x, 0.f);
      r[1] = fmaxf(pv.y - qv.y, 0.f);
      r[2] = fmaxf(pv.z - qv.z, 0.f);
      r[3] = fmaxf(pv.w - qv.w, 0.f);
    }
    const double s = (double)r[0] + (double)r[1] + (double)r[2] + (double)r[3];
    // inclusive scan of s over the block
    double inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __syncthreads();  // previous tile's readers of s_d are done
    if (lane == 63) s_d[wave] = inc;
    __syncthreads();
    double wave_prefix = 0.0, tile_total = 0.0;
    for (int w = 0; w < nwaves; ++w) {
      if (w < wave) wave_prefix += s_d[w];
      tile_total += s_d[w];
    }
    const double excl = carry + wave_prefix + (inc - s);
    if (i < V && excl <= thr && excl + s > thr) {
      double c = excl;
      int idx = i + 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c += (double)r[j];
        if (c > thr) { idx = i + j; break; }
      }
      atomicMin(&s_found, idx);
    }
    carry += tile_total;
    __syncthreads();
    if (s_found != 0x7fffffff) break;  // uniform: read after the barrier
  }
  const int found = s_found;
  return found != 0x7fffffff ? found : last_pos;
}

__global__ __launch_bounds__(kSampleThreads) void rejection_sample_kernel(
    const float* __restrict__ target_probs, const int64_t* __restrict__ bonus_ids,
    const float* __restrict__ draft_probs, const int64_t* __restrict__ draft_ids,
    const float* __restrict__ u_accept, const float* __restrict__ u_recover,
    const int32_t* __restrict__ proposal_lens, int64_t* __restrict__ out_ids,
    unsigned long long* __restrict__ counters, int k, int V) {
  __shared__ int s_first;
  const int b = blockIdx.x, tid = threadIdx.x;
  int L = proposal_lens ? proposal_lens[b] : k;
  L = min(max(L, 0), k);
  const float* pb = target_probs + (size_t)b * (k + 1) * V;
  const float* qb = draft_probs + (size_t)b * k * V;

  if (tid < SD_WAVE) {
    bool ok = true;
    if (tid < L) {
      const int64_t x = draft_ids[(size_t)b * k + tid];
      ok = false;
      if (x >= 0 && x < V) {
        const float pv = pb[(size_t)tid * V + x];
        const float qv = qb[(size_t)tid * V + x];
        const float uq = __fmul_rn(u_accept[(size_t)b * k + tid], qv);
        ok = uq < pv;
      }
    }
    const unsigned long long rejected = ~__ballot(ok);
    const int first = rejected ? (__ffsll((long long)rejected) - 1) : SD_WAVE;
    if (tid == 0) s_first = min(first, L);
  }
  __syncthreads();
  const int j = s_first;

  int64_t tok;
  if (j == L && L == k) {
    tok = bonus_ids[b];
  } else {
    const float* prow = pb + (size_t)j * V;
    const float* qrow = (j < L) ? qb + (size_t)j * V : nullptr;
    tok = block_sample_residual(prow, qrow, V, u_recover[(size_t)b * (k + 1) + j]);
  }
  if (tid <= k) {
    int64_t v = -1;
    if (tid < j) v = draft_ids[(size_t)b * k + tid];
    else if (tid == j) v = tok;
    out_ids[(size_t)b * (k + 1) + tid] = v;
  }
  if (tid == 0 && counters) {
    atomicAdd(&counters[0], (unsigned long long)j);
    atomicAdd(&counters[1], (unsigned long long)(j + 1));
    atomicAdd(&counters[2], (unsigned long long)L);
  }
}

__global__ __launch_bounds__(SD_WAVE) void greedy_verify_kernel(
    const int64_t* __restrict__ target_ids, const int64_t* __restrict__ draft_ids,
    const int32_t* __restrict__ proposal_lens, int64_t* __restrict__ out_ids,
    unsigned long long* __restrict__ counters, int k) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int L = proposal_lens ? proposal_lens[b] : k;
  L = min(max(L, 0), k);
  bool ok = true;
  if (lane < L)
    ok = draft_ids[(size_t)b * k + lane] == target_ids[(size_t)b * (k + 1) + lane];
  const unsigned long long rejected = ~__ballot(ok);
  const int first = rejected ? (__ffsll((long long)rejected) - 1) : SD_WAVE;
  const int j = min(first, L);
  if (lane <= k) {
    int64_t v = -1;
    if (lane < j) v = draft_ids[(size_t)b * k + lane];
    else if (lane == j) v = target_ids[(size_t)b * (k + 1) + j];
    out_ids[(size_t)b * (k + 1) + lane] = v;
  }
  if (lane == 0 && counters) {
    atomicAdd(&counters[0], (unsigned long long)j);
    atomicAdd(&counters[1], (unsigned long long)(j + 1));
    atomicAdd(&counters[2], (unsigned long long)L);
  }
}

__global__ __launch_bounds__(kSampleThreads) void argmax_rows_kernel(
    const float* __restrict__ logits, int64_t ld, int64_t* __restrict__ ids, int V) {
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  const int r = blockIdx.x;
  const int idx = block_argmax(logits + (size_t)r * ld, V, s_f, s_i);
  if (threadIdx.x == 0) ids[r] = idx;
}

__global__ __launch_bounds__(kSampleThreads) void softmax_rows_kernel(
    const float* __restrict__ logits, int64_t ld, float inv_t,
    float* __restrict__ probs, int64_t ld_probs, int V) {
  __shared__ float s_f[kSampleWaves];
  __shared__ double s_d[kSampleWaves];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (size_t)r * ld;
  float* out = probs + (size_t)r * ld_probs;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += blockDim.x) mx = fmaxf(mx, row[i] * inv_t);
  mx = wave_max(mx);
  if (lane == 0) s_f[wave] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < kSampleWaves; ++w) mx = fmaxf(mx, s_f[w]);
  double sum = 0.0;
  for (int i = tid; i < V; i += blockDim.x) sum += (double)expf(row[i] * inv_t - mx);
  const double total = block_sum_f64(sum, s_d);
  const float inv = (float)(1.0 / total);
  for (int i = tid; i < V; i += blockDim.x) out[i] = expf(row[i] * inv_t - mx) * inv;
}

__global__ __launch_bounds__(kSampleThreads) void sample_rows_kernel(
    const float* __restrict__ probs, int64_t ld, const float* __restrict__ uniforms,
    int64_t* __restrict__ ids, int V) {
  const int r = blockIdx.x;
  const int idx = block_sample_residual(probs + (size_t)r * ld, nullptr, V, uniforms[r]);
  if (threadIdx.x == 0) ids[r] = idx;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// The proposal of sequence b at draft step `step`, or (synthetic acceptance control) the
// true token of that position with probability accept_prob.
__device__ __forceinline__ int32_t forced_proposal(int32_t tok, int b, int step,
                                                   const int32_t* __restrict__ seq_lens,
                                                   const int32_t* __restrict__ truth,
                                                   int64_t truth_stride, float accept_prob,
                                                   uint32_t seed) {
  if (truth) {
    const int pos = seq_lens[b] + step;  // position the proposal would occupy
    const uint32_t h = mix32(mix32(seed ^ (uint32_t)b * 0x9e3779b9u) + (uint32_t)pos);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    if (u < accept_prob && pos < truth_stride) tok = truth[(size_t)b * truth_stride + pos];
  }
  return tok;
}

__global__ void store_proposal_kernel(const int64_t* __restrict__ ids,
                                      int32_t* __restrict__ tok_buf, int k, int step,
                                      const int32_t* __restrict__ seq_lens,
                                      const int32_t* __restrict__ truth,
                                      int64_t truth_stride, float accept_prob,
                                      uint32_t seed, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  tok_buf[(size_t)b * (k + 2) + 2 + step] =
      forced_proposal((int32_t)ids[b], b, step, seq_lens, truth, truth_stride, accept_prob, seed);
}

// Applies the emitted tokens o[0..k] (-1 padded) of sequence b to the device state and
// writes the step record {n_emitted, sequence length, tokens} if asked.
__device__ __forceinline__ void apply_emitted(const int64_t* o, int b,
                                              int32_t* __restrict__ seq_lens,
                                              int32_t* __restrict__ tok_buf,
                                              int32_t* __restrict__ gen_tokens,
                                              int64_t gen_stride, int gen_capacity,
                                              int32_t* __restrict__ gen_counts,
                                              int32_t* __restrict__ step_out, int k) {
  int n_emit = 0;
  while (n_emit <= k && o[n_emit] >= 0) ++n_emit;
  int32_t* rec = step_out ? step_out + (size_t)b * (k + 3) : nullptr;
  if (rec) {
    rec[0] = n_emit;
    rec[1] = seq_lens[b] + n_emit;
    for (int i = 0; i <= k; ++i) rec[2 + i] = i < n_emit ? (int32_t)o[i] : -1;
  }
  if (n_emit == 0) return;  // nothing emitted (should not happen)
  int32_t* tb = tok_buf + (size_t)b * (k + 2);
  const int32_t last = (int32_t)o[n_emit - 1];
  const int32_t prev = n_emit >= 2 ? (int32_t)o[n_emit - 2] : tb[1];
  const int g = gen_counts[b];
  for (int i = 0; i < n_emit; ++i)
    if (gen_tokens && g + i < gen_capacity)
      gen_tokens[(size_t)b * gen_stride + g + i] = (int32_t)o[i];
  gen_counts[b] = g + n_emit;
  seq_lens[b] += n_emit;
  tb[0] = prev;
  tb[1] = last;
}

__global__ void gather_proposals_kernel(const int32_t* __restrict__ tok_buf,
                                        int64_t* __restrict__ draft_ids, int B, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * k) return;
  const int b = i / k, j = i % k;
  draft_ids[i] = tok_buf[(size_t)b * (k + 2) + 2 + j];
}

__global__ void advance_state_kernel(const int64_t* __restrict__ out_ids,
                                     int32_t* __restrict__ seq_lens,
                                     int32_t* __restrict__ tok_buf,
                                     int32_t* __restrict__ gen_tokens, int64_t gen_stride,
                                     int gen_capacity, int32_t* __restrict__ gen_counts,
                                     int32_t* __restrict__ step_out, int B, int k) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  apply_emitted(out_ids + (size_t)b * (k + 1), b, seq_lens, tok_buf, gen_tokens, gen_stride,
                gen_capacity, gen_counts, step_out, k);
}

constexpr int kArgmaxParts = SD_ARGMAX_PARTS;

// Stage 1 of a split row argmax: block (part, row) scans one 1/64 slice of the row.
__global__ __launch_bounds__(256) void argmax_partial_kernel(
    const float* __restrict__ logits, int64_t ld, float* __restrict__ part_val,
    int32_t* __restrict__ part_idx, int V) {
  __shared__ float s_f[4];
  __shared__ int s_i[4];
  const int part = blockIdx.x, r = blockIdx.y;
  const int chunk = ((V + kArgmaxParts - 1) / kArgmaxParts + 3) & ~3;
  const int lo = part * chunk, hi = min(lo + chunk, V);
  const float* row = logits + (size_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x * 4; i < hi; i += 256 * 4) {
    if (i + 3 < hi) {
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e[j] > best) { best = e[j]; bi = i + j; }
    } else {
      for (int j = 0; j < 4 && i + j < hi; ++j)
        if (row[i + j] > best) { best = row[i + j]; bi = i + j; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_f[wave] = best; s_i[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_f[w] > best || (s_f[w] == best && s_i[w] < bi)) { best = s_f[w]; bi = s_i[w]; }
    part_val[(size_t)r * kArgmaxParts + part] = best;
    part_idx[(size_t)r * kArgmaxParts + part] = bi;
  }
}

// Stage 2, one wave per row: the smallest index among the row's maxima.
__device__ __forceinline__ int wave_argmax_finish(const float* __restrict__ part_val,
                                                  const int32_t* __restrict__ part_idx, int row,
                                                  int lane) {
  float best = part_val[(size_t)row * kArgmaxParts + lane];
  int bi = part_idx[(size_t)row * kArgmaxParts + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  return bi == 0x7fffffff ? 0 : bi;
}

__global__ __launch_bounds__(SD_WAVE) void propose_from_partials_kernel(
    const float* __restrict__ part_val, const int32_t* __restrict__ part_idx, int row_offset,
    int row_step, int32_t* __restrict__ tok_buf, int k, int step,
    const int32_t* __restrict__ seq_lens, const int32_t* __restrict__ truth, int64_t truth_stride,
    float accept_prob, uint32_t seed) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int id = wave_argmax_finish(part_val, part_idx, row_offset + b * row_step, lane);
  if (lane == 0)
    tok_buf[(size_t)b * (k + 2) + 2 + step] =
        forced_proposal(id, b, step, seq_lens, truth, truth_stride, accept_prob, seed);
}

// Greedy verification straight from the target's argmax partials: wave j finishes row j
// of the sequence, wave 0 then matches the proposals in tok_buf and applies the result.
__global__ __launch_bounds__(SD_WAVE * 8) void verify_from_partials_kernel(
    const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
    int32_t* __restrict__ tok_buf, const int32_t* __restrict__ proposal_lens,
    int64_t* __restrict__ out_ids, unsigned long long* __restrict__ counters,
    int32_t* __restrict__ seq_lens, int32_t* __restrict__ gen_tokens, int64_t gen_stride,
    int gen_capacity, int32_t* __restrict__ gen_counts, int32_t* __restrict__ step_out, int k) {
  __shared__ int s_target[8];
  __shared__ int64_t s_out[8];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave <= k) {
    const int id = wave_argmax_finish(part_val, part_idx, b * (k + 1) + wave, lane);
    if (lane == 0) s_target[wave] = id;
  }
  __syncthreads();
  if (wave != 0) return;
  int L = proposal_lens ? proposal_lens[b] : k;
  L = min(max(L, 0), k);
  const int32_t* prop = tok_buf + (size_t)b * (k + 2) + 2;
  bool ok = true;
  if (lane < L) ok = prop[lane] == s_target[lane];
  const unsigned long long rejected = ~__ballot(ok);
  const int first = rejected ? (__ffsll((long long)rejected) - 1) : SD_WAVE;
  const int j = min(first, L);
  if (lane <= k) {
    int64_t v = -1;
    if (lane < j) v = prop[lane];
    else if (lane == j) v = s_target[j];
    s_out[lane] = v;
    if (out_ids) out_ids[(size_t)b * (k + 1) + lane] = v;
  }
  // s_out is written and read by lanes of this one wave only
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    if (counters) {
      atomicAdd(&counters[0], (unsigned long long)j);
      atomicAdd(&counters[1], (unsigned long long)(j + 1));
      atomicAdd(&counters[2], (unsigned long long)L);
    }
    apply_emitted(s_out, b, seq_lens, tok_buf, gen_tokens, gen_stride, gen_capacity, gen_counts,
                  step_out, k);
  }
}

}  // namespace

extern "C" int sd_rejection_sample(const float* target_probs, const int64_t* bonus_ids,
                                   const float* draft_probs, const int64_t* draft_ids,
                                   const float* u_accept, const float* u_recover,
                                   const int32_t* proposal_lens, int64_t* out_ids,
                                   int64_t* counters, int B, int k, int V,
                                   sd_stream_t stream) {
  SD_REQUIRE(B >= 0 && k >= 1 && k <= 63 && V >= 4, SD_ERR_INVALID,
             "sd_rejection_sample: need B>=0, 1<=k<=63, V>=4 (B=%d k=%d V=%d)", B, k, V);
  SD_REQUIRE(V % 4 == 0, SD_ERR_UNSUPPORTED, "sd_rejection_sample: V %% 4 != 0 (V=%d)", V);
  SD_REQUIRE(target_probs && bonus_ids && draft_probs && draft_ids && u_accept && u_recover &&
                 out_ids, SD_ERR_INVALID, "sd_rejection_sample: null pointer");
  SD_REQUIRE(sd_aligned16(target_probs) && sd_aligned16(draft_probs), SD_ERR_INVALID,
             "sd_rejection_sample: probability rows must be 16-byte aligned");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(rejection_sample_kernel, dim3(B), dim3(kSampleThreads), 0,
                     (hipStream_t)stream, target_probs, bonus_ids, draft_probs, draft_ids,
                     u_accept, u_recover, proposal_lens, out_ids,
                     (unsigned long long*)counters, k, V);
  SD_CHECK_LAUNCH("sd_rejection_sample");
  return SD_OK;
}

extern "C" int sd_greedy_verify(const int64_t* target_ids, const int64_t* draft_ids,
                                const int32_t* proposal_lens, int64_t* out_ids,
                                int64_t* counters, int B, int k, sd_stream_t stream) {
  SD_REQUIRE(B >= 0 && k >= 1 && k <= 63, SD_ERR_INVALID,
             "sd_greedy_verify: need B>=0, 1<=k<=63 (B=%d k=%d)", B, k);
  SD_REQUIRE(target_ids && draft_ids && out_ids, SD_ERR_INVALID,
             "sd_greedy_verify: null pointer");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(greedy_verify_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream,
                     target_ids, draft_ids, proposal_lens, out_ids,
                     (unsigned long long*)counters, k);
  SD_CHECK_LAUNCH("sd_greedy_verify");
  return SD_OK;
}

extern "C" int sd_argmax_rows(const float* logits, int64_t ld, int64_t* ids, int R, int V,
                              sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 1 && ld >= V && logits && ids, SD_ERR_INVALID,
             "sd_argmax_rows: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(sd_aligned16(logits) && ld % 4 == 0, SD_ERR_INVALID,
             "sd_argmax_rows: rows must be 16-byte aligned");
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(R), dim3(kSampleThreads), 0,
                     (hipStream_t)stream, logits, ld, ids, V);
  SD_CHECK_LAUNCH("sd_argmax_rows");
  return SD_OK;
}

extern "C" int sd_softmax_rows(const float* logits, int64_t ld, float temperature,
                               float* probs, int64_t ld_probs, int R, int V,
                               sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 1 && ld >= V && ld_probs >= V && logits && probs &&
                 temperature > 0.f,
             SD_ERR_INVALID, "sd_softmax_rows: bad arguments (R=%d V=%d T=%g)", R, V,
             (double)temperature);
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(R), dim3(kSampleThreads), 0,
                     (hipStream_t)stream, logits, ld, 1.0f / temperature, probs, ld_probs, V);
  SD_CHECK_LAUNCH("sd_softmax_rows");
  return SD_OK;
}

extern "C" int sd_sample_rows(const float* probs, int64_t ld, const float* uniforms,
                              int64_t* ids, int R, int V, sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 4 && ld >= V && probs && uniforms && ids, SD_ERR_INVALID,
             "sd_sample_rows: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(V % 4 == 0 && ld % 4 == 0 && sd_aligned16(probs), SD_ERR_UNSUPPORTED,
             "sd_sample_rows: V and ld must be multiples of 4, rows 16-byte aligned");
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(R), dim3(kSampleThreads), 0,
                     (hipStream_t)stream, probs, ld, uniforms, ids, V);
  SD_CHECK_LAUNCH("sd_sample_rows");
  return SD_OK;
}

extern "C" int sd_store_proposal(const int64_t* ids, int32_t* tok_buf, int k, int step,
                                 const int32_t* seq_lens, const int32_t* truth,
                                 int64_t truth_stride, float accept_prob, uint32_t seed,
                                 int B, sd_stream_t stream) {
  SD_REQUIRE(ids && tok_buf && seq_lens && B >= 0 && k >= 1 && step >= 0 && step < k,
             SD_ERR_INVALID, "sd_store_proposal: bad arguments (B=%d k=%d step=%d)", B, k, step);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(store_proposal_kernel, dim3((B + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, ids, tok_buf, k, step, seq_lens, truth,
                     truth_stride, accept_prob, seed, B);
  SD_CHECK_LAUNCH("sd_store_proposal");
  return SD_OK;
}

extern "C" int sd_gather_proposals(const int32_t* tok_buf, int64_t* draft_ids, int B, int k,
                                   sd_stream_t stream) {
  SD_REQUIRE(tok_buf && draft_ids && B >= 0 && k >= 1, SD_ERR_INVALID,
             "sd_gather_proposals: bad arguments");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(gather_proposals_kernel, dim3((B * k + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, tok_buf, draft_ids, B, k);
  SD_CHECK_LAUNCH("sd_gather_proposals");
  return SD_OK;
}

extern "C" int sd_advance_state(const int64_t* out_ids, int32_t* seq_lens, int32_t* tok_buf,
                                int32_t* gen_tokens, int64_t gen_stride, int gen_capacity,
                                int32_t* gen_counts, int32_t* step_out, int B, int k,
                                sd_stream_t stream) {
  SD_REQUIRE(out_ids && seq_lens && tok_buf && gen_counts && B >= 0 && k >= 0,
             SD_ERR_INVALID, "sd_advance_state: bad arguments");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(advance_state_kernel, dim3((B + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, out_ids, seq_lens, tok_buf, gen_tokens, gen_stride,
                     gen_capacity, gen_counts, step_out, B, k);
  SD_CHECK_LAUNCH("sd_advance_state");
  return SD_OK;
}

extern "C" int sd_argmax_partial(const float* logits, int64_t ld, float* part_val,
                                 int32_t* part_idx, int R, int V, sd_stream_t stream) {
  SD_REQUIRE(logits && part_val && part_idx && R >= 0 && V >= 1, SD_ERR_INVALID,
             "sd_argmax_partial: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(ld >= V && ld % 4 == 0 && sd_aligned16(logits), SD_ERR_INVALID,
             "sd_argmax_partial: rows must be 16-byte aligned");
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(argmax_partial_kernel, dim3(kArgmaxParts, R), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, part_val, part_idx, V);
  SD_CHECK_LAUNCH("sd_argmax_partial");
  return SD_OK;
}

extern "C" int sd_propose_from_partials(const float* part_val, const int32_t* part_idx,
                                        int row_offset, int row_step, int32_t* tok_buf, int k,
                                        int step, const int32_t* seq_lens, const int32_t* truth,
                                        int64_t truth_stride, float accept_prob, uint32_t seed,
                                        int B, sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx && tok_buf && B >= 0 && k >= 1 && step >= 0 && step < k &&
                 row_offset >= 0 && row_step >= 1, SD_ERR_INVALID,
             "sd_propose_from_partials: bad arguments (B=%d k=%d step=%d)", B, k, step);
  SD_REQUIRE(!truth || seq_lens, SD_ERR_INVALID, "sd_propose_from_partials: truth without seq_lens");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(propose_from_partials_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream,
                     part_val, part_idx, row_offset, row_step, tok_buf, k, step, seq_lens, truth,
                     truth_stride, accept_prob, seed);
  SD_CHECK_LAUNCH("sd_propose_from_partials");
  return SD_OK;
}

extern "C" int sd_verify_from_partials(const float* part_val, const int32_t* part_idx,
                                       int32_t* tok_buf, const int32_t* proposal_lens,
                                       int64_t* out_ids, int64_t* counters, int32_t* seq_lens,
                                       int32_t* gen_tokens, int64_t gen_stride, int gen_capacity,
                                       int32_t* gen_counts, int32_t* step_out, int B, int k,
                                       sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx && tok_buf && seq_lens && gen_counts && B >= 0 && k >= 0 &&
                 k <= 7, SD_ERR_INVALID, "sd_verify_from_partials: bad arguments (B=%d k=%d)", B, k);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(verify_from_partials_kernel, dim3(B), dim3(SD_WAVE * 8), 0,
                     (hipStream_t)stream, part_val, part_idx, tok_buf, proposal_lens, out_ids,
                     reinterpret_cast<unsigned long long*>(counters), seq_lens, gen_tokens,
                     gen_stride, gen_capacity, gen_counts, step_out, k);
  SD_CHECK_LAUNCH("sd_verify_from_partials");
  return SD_OK;
}
