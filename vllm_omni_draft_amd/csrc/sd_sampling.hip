// Acceptance and sampling kernels: rejection sampling over probability rows,
// greedy verification, row argmax / softmax / inverse-CDF sampling, and the
// per-sequence device state updates of a speculative step.
//
// The accept mask of a sequence is one wavefront ballot: lane i tests position
// i, lanes past the proposal length vote "accepted", and the first rejected
// position is the lowest clear bit of the 64-bit mask.
#include <climits>
#include <mutex>

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
// excl / n_excl: indices whose mass counts as zero (a short list, every thread sees the same).
__device__ int block_sample_residual(const float* __restrict__ p,
                                     const float* __restrict__ q, int V, float u,
                                     const int* excl = nullptr, int n_excl = 0) {
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
    float r[4] = {fmaxf(pv.x - qv.x, 0.f), fmaxf(pv.y - qv.y, 0.f),
                  fmaxf(pv.z - qv.z, 0.f), fmaxf(pv.w - qv.w, 0.f)};
    for (int e = 0; e < n_excl; ++e) {
      const unsigned d = (unsigned)(excl[e] - i);
      if (d < 4u) r[d] = 0.f;
    }
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
      for (int e = 0; e < n_excl; ++e) {
        const unsigned d = (unsigned)(excl[e] - i);
        if (d < 4u) r[d] = 0.f;
      }
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

// Typical acceptance: proposal x_i stands iff p_i(x_i) > min(cap, alpha * exp(-H(p_i))),
// H(p) = -sum p log(p + 1e-5); the first rejected position emits argmax p_j.  One workgroup per
// sequence walks the positions in order and stops at the first rejection, so rows past it are
// never read.  Every thread computes the same threshold: the walk is uniform.
__global__ __launch_bounds__(kSampleThreads) void typical_acceptance_kernel(
    const float* __restrict__ target_probs, const int64_t* __restrict__ bonus_ids,
    const int64_t* __restrict__ draft_ids, const int32_t* __restrict__ proposal_lens,
    float cap, float alpha, int64_t* __restrict__ out_ids,
    unsigned long long* __restrict__ counters, int k, int V) {
  __shared__ double s_d[kSampleWaves];
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  const int b = blockIdx.x, tid = threadIdx.x;
  int L = proposal_lens ? proposal_lens[b] : k;
  L = min(max(L, 0), k);
  const float* pb = target_probs + (size_t)b * (k + 1) * V;
  int j = 0;
  for (; j < L; ++j) {
    const float* row = pb + (size_t)j * V;
    double local = 0.0;
    for (int i = tid * 4; i < V; i += kSampleThreads * 4) {
      if (i + 3 < V) {
        const float4 v = *reinterpret_cast<const float4*>(row + i);
        local += (double)(v.x * logf(v.x + 1e-5f)) + (double)(v.y * logf(v.y + 1e-5f)) +
                 (double)(v.z * logf(v.z + 1e-5f)) + (double)(v.w * logf(v.w + 1e-5f));
      } else {
        for (int e = i; e < V; ++e) local += (double)(row[e] * logf(row[e] + 1e-5f));
      }
    }
    const double entropy = -block_sum_f64(local, s_d);
    const float thr = (float)fmin((double)cap, (double)alpha * exp(-entropy));
    const int64_t x = draft_ids[(size_t)b * k + j];
    const float cand = (x >= 0 && x < V) ? row[x] : -1.f;
    if (!(cand > thr)) break;
  }
  int64_t tok;
  if (j == L && L == k) tok = bonus_ids[b];
  else tok = block_argmax(pb + (size_t)j * V, V, s_f, s_i);
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

__global__ __launch_bounds__(SD_WAVE) void map_tokens_kernel(int32_t* __restrict__ tokens, int64_t stride,
                                                             int col, const int32_t* __restrict__ map,
                                                             int n_map, int B) {
  const int b = blockIdx.x * SD_WAVE + threadIdx.x;
  if (b >= B) return;
  int32_t* t = tokens + (size_t)b * stride + col;
  *t = map[min(max(*t, 0), n_map - 1)];
}

// ---- token trees ------------------------------------------------------------
__global__ __launch_bounds__(SD_WAVE) void tree_prepare_kernel(
    const int32_t* __restrict__ tok_buf, int64_t tok_stride, int tok_col, int32_t* __restrict__ tree_tok,
    const int32_t* __restrict__ tree_depth, const int32_t* __restrict__ seq_lens, int pos_offset,
    int32_t* __restrict__ rope_pos, int q_len) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) tree_tok[(size_t)b * q_len] = tok_buf[(size_t)b * tok_stride + tok_col];
  if (t < q_len)
    rope_pos[(size_t)b * q_len + t] = seq_lens[b] + pos_offset + tree_depth[(size_t)b * q_len + t];
}

// The b largest entries of a logits row as tokens (largest first; ties: lowest index), one
// workgroup per sequence and row: b sweeps over the row, each skipping the winners so far.
// Feeds the levels of a tree grown by the draft model.
__global__ __launch_bounds__(kSampleThreads) void topk_tokens_kernel(
    const float* __restrict__ logits, int64_t ld, int V, int row_offset, int row_step, int b,
    int32_t* __restrict__ out_tok, int64_t out_stride, int out_col) {
  row_offset += blockIdx.y;
  out_col += blockIdx.y * b;
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  __shared__ int s_win[SD_TOPK_TOKENS_MAX];
  const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (size_t)(row_offset + seq * row_step) * ld;
  for (int c = 0; c < b; ++c) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += kSampleThreads) {
      const float v = row[i];
      bool taken = false;
      for (int e = 0; e < c; ++e) taken |= s_win[e] == i;
      if (!taken && v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_f[wave] = best; s_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kSampleWaves; ++w)
        if (s_f[w] > best || (s_f[w] == best && s_i[w] < bi)) { best = s_f[w]; bi = s_i[w]; }
      // (a row with fewer than b finite entries repeats nothing: the free indices follow)
      if (bi == 0x7fffffff) {
        bi = 0;
        for (int e = 0; e < c; ++e) if (s_win[e] == bi) { ++bi; e = -1; }
      }
      s_win[c] = bi;
      out_tok[(size_t)seq * out_stride + out_col + c] = bi;
    }
    __syncthreads();
  }
}

// One wave per sequence: the walk is a chain of dependent steps, each step a wave-wide
// search for the child that carries the target's token.
__global__ __launch_bounds__(SD_WAVE) void tree_accept_greedy_kernel(
    const int64_t* __restrict__ target_ids, const int32_t* __restrict__ tree_tok,
    const int32_t* __restrict__ tree_parent, const int32_t* __restrict__ tree_len, int q_len, int k,
    int64_t* __restrict__ out_ids, int32_t* __restrict__ path, int32_t* __restrict__ n_path,
    unsigned long long* __restrict__ counters) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int len = min(max(tree_len[b], 1), q_len);
  const bool node = lane >= 1 && lane < len;
  const int my_tok = node ? tree_tok[(size_t)b * q_len + lane] : -1;
  const int my_par = node ? tree_parent[(size_t)b * q_len + lane] : -1;
  // depth of the tree (its longest chain): parents come first, so q_len sweeps settle it
  int depth = 0;
  for (int sweep = 0; sweep < q_len; ++sweep) {
    const int pd = __shfl(depth, my_par < 0 ? 0 : my_par, SD_WAVE);
    const int nd = node ? pd + 1 : 0;
    if (!__any(nd != depth)) break;
    depth = nd;
  }
  int max_depth = depth;
  for (int o = 32; o > 0; o >>= 1) max_depth = max(max_depth, __shfl_xor(max_depth, o, SD_WAVE));
  int cur = 0, j = 0;
  for (; j < k; ++j) {
    const int64_t want = target_ids[(size_t)b * q_len + cur];
    const unsigned long long hit = __ballot(node && my_par == cur && (int64_t)my_tok == want);
    if (!hit) break;
    cur = __ffsll((long long)hit) - 1;
    if (lane == 0) {
      out_ids[(size_t)b * (k + 1) + j] = want;
      path[(size_t)b * k + j] = cur;
    }
  }
  if (lane == 0) {
    out_ids[(size_t)b * (k + 1) + j] = target_ids[(size_t)b * q_len + cur];
    n_path[b] = j;
    if (counters) {
      atomicAdd(&counters[0], (unsigned long long)j);
      atomicAdd(&counters[1], (unsigned long long)(j + 1));
      atomicAdd(&counters[2], (unsigned long long)min(max_depth, k));
    }
  }
  if (lane > j && lane <= k) out_ids[(size_t)b * (k + 1) + lane] = -1;
  if (lane >= j && lane < k) path[(size_t)b * k + lane] = -1;
}

// Multi-candidate speculative sampling down a token tree whose candidates carry no draft
// distribution (point-mass proposals).  At a node with remaining target distribution r (the
// node's probability row, tokens of rejected children zeroed) the children are tried in row
// order: child x stands with probability r(x) / mass(r); a rejected child's token leaves r.
// When no child stands (or the depth limit is reached) the emitted token is drawn from r.
// Every thread walks the same scalar chain; only the final draw uses the whole workgroup.
__global__ __launch_bounds__(kSampleThreads) void tree_accept_sample_kernel(
    const float* __restrict__ probs, int64_t ld, const int32_t* __restrict__ tree_tok,
    const int32_t* __restrict__ tree_parent, const int32_t* __restrict__ tree_len, int q_len, int k,
    const float* __restrict__ u_accept, const float* __restrict__ u_final, int V,
    int64_t* __restrict__ out_ids, int32_t* __restrict__ path, int32_t* __restrict__ n_path,
    unsigned long long* __restrict__ counters) {
  __shared__ int s_excl[SD_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = min(max(tree_len[b], 1), q_len);
  const int32_t* tok = tree_tok + (size_t)b * q_len;
  const int32_t* par = tree_parent + (size_t)b * q_len;
  const float* ua = u_accept + (size_t)b * q_len;
  int cur = 0, j = 0, n_excl = 0;
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
        // a sibling that carries a token an earlier (rejected) sibling carried has nothing left
        // to stand on: that token's mass left r already
        bool dup = false;
        for (int e = 1; e < c; ++e) dup |= par[e] == cur && tok[e] == x;
        if (dup) continue;
        const double px = (double)row[x];
        if ((double)ua[c] * (1.0 - removed) < px) {
          next = c;
          break;
        }
        removed += px;
        if (tid == 0) s_excl[n_excl] = x;   // read after the barrier below
        ++n_excl;
      }
    }
    if (next < 0) break;
    if (tid == 0) {
      out_ids[(size_t)b * (k + 1) + j] = tok[next];
      path[(size_t)b * k + j] = next;
    }
    cur = next;
    ++j;
  }
  __syncthreads();
  const int last = block_sample_residual(probs + ((size_t)b * q_len + cur) * ld, nullptr, V, u_final[b],
                                         s_excl, n_excl);
  if (tid == 0) {
    out_ids[(size_t)b * (k + 1) + j] = last;
    n_path[b] = j;
    // the longest chain of the tree counts as what was proposed
    int max_depth = 0;
    for (int c = 1; c < len; ++c) {
      int d = 0;
      // (a parent is a smaller row; anything else ends the walk)
      for (int a = c; a > 0 && d <= q_len; a = (par[a] >= 0 && par[a] < a) ? par[a] : 0) ++d;
      max_depth = max(max_depth, d);
    }
    if (counters) {
      atomicAdd(&counters[0], (unsigned long long)j);
      atomicAdd(&counters[1], (unsigned long long)(j + 1));
      atomicAdd(&counters[2], (unsigned long long)min(max_depth, k));
    }
  }
  if (tid > j && tid <= k) out_ids[(size_t)b * (k + 1) + tid] = -1;
  if (tid >= j && tid < k) path[(size_t)b * k + tid] = -1;
}

__global__ __launch_bounds__(kSampleThreads) void argmax_rows_kernel(
    const float* __restrict__ logits, int64_t ld, int64_t* __restrict__ ids, int V) {
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  const int r = blockIdx.x;
  const int idx = block_argmax(logits + (size_t)r * ld, V, s_f, s_i);
  if (threadIdx.x == 0) ids[r] = idx;
}

// Temperature of row r: the scalar, or (temps != null) that of sequence r / rows_per_seq.  A
// temperature of 0 asks for the greedy distribution: all mass on the first maximum of the row.
struct RowTemps {
  const float* temps;
  int rows_per_seq;
  float inv_t;
};
__device__ __forceinline__ float row_inv_t(const RowTemps& t, int r, bool& greedy) {
  greedy = false;
  if (!t.temps) return t.inv_t;
  const float T = t.temps[r / t.rows_per_seq];
  greedy = !(T > 0.f);
  return greedy ? 1.f : 1.f / T;
}

__global__ __launch_bounds__(kSampleThreads) void softmax_rows_kernel(
    const float* __restrict__ logits, int64_t ld, RowTemps tm,
    float* __restrict__ probs, int64_t ld_probs, int V) {
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  __shared__ double s_d[kSampleWaves];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (size_t)r * ld;
  float* out = probs + (size_t)r * ld_probs;
  bool greedy;
  const float inv_t = row_inv_t(tm, r, greedy);
  if (greedy) {   // block-uniform
    const int best = block_argmax(row, V, s_f, s_i);
    for (int i = tid; i < V; i += blockDim.x) out[i] = i == best ? 1.f : 0.f;
    return;
  }
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

__global__ __launch_bounds__(256) void add_logit_bias_kernel(
    float* __restrict__ logits, int64_t ld, const float* __restrict__ bias, int64_t ld_bias,
    const int32_t* __restrict__ enabled, int rows_per_seq, int V) {
  const int r = blockIdx.y, seq = r / rows_per_seq;
  if (!enabled[seq]) return;   // block-uniform
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  float* row = logits + (size_t)r * ld;
  const float* add = bias + (size_t)seq * ld_bias;
  if (i + 3 < V) {
    float4 v = *reinterpret_cast<float4*>(row + i);
    const float4 a = *reinterpret_cast<const float4*>(add + i);
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    *reinterpret_cast<float4*>(row + i) = v;
  } else {
    for (int j = i; j < V; ++j) row[j] += add[j];
  }
}

// Log-probability and rank of one chosen token per row, and the row's n_top largest entries in
// (value descending, index ascending) order: each is the maximum of what comes after the previous
// one in that order, so n_top passes over a row that sits in L2 and no sort.
__global__ __launch_bounds__(kSampleThreads) void token_logprobs_kernel(
    const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ tokens, int V, int n_top,
    int32_t* __restrict__ out_ids, float* __restrict__ out_lp, int32_t* __restrict__ out_rank) {
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  __shared__ double s_d[kSampleWaves];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok = tokens[r];
  if (tok < 0 || tok >= V) return;  // block-uniform
  const float* row = logits + (size_t)r * ld;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += blockDim.x) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if (lane == 0) s_f[wave] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < kSampleWaves; ++w) mx = fmaxf(mx, s_f[w]);
  const float chosen = row[tok];
  double sum = 0.0, ge = 0.0;
  for (int i = tid; i < V; i += blockDim.x) {
    sum += (double)expf(row[i] - mx);
    ge += row[i] >= chosen ? 1.0 : 0.0;
  }
  const double total = block_sum_f64(sum, s_d);
  const double n_ge = block_sum_f64(ge, s_d);
  const float lse = mx + (float)log(total);
  const size_t o = (size_t)r * (n_top + 1);
  if (tid == 0) {
    out_ids[o] = tok;
    out_lp[o] = chosen - lse;
    out_rank[r] = (int)n_ge;
  }
  float pv = INFINITY;
  int pi = -1;
  for (int j = 0; j < n_top; ++j) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += blockDim.x) {
      const float v = row[i];
      if ((v < pv || (v == pv && i > pi)) && (v > best || (v == best && i < bi))) {
        best = v;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { s_f[wave] = best; s_i[wave] = bi; }
    __syncthreads();
    best = s_f[0];
    bi = s_i[0];
    for (int w = 1; w < kSampleWaves; ++w)
      if (s_f[w] > best || (s_f[w] == best && s_i[w] < bi)) { best = s_f[w]; bi = s_i[w]; }
    const bool none = bi == 0x7fffffff || best == -INFINITY;   // nothing finite is left
    if (tid == 0) {
      out_ids[o + 1 + j] = none ? -1 : bi;
      out_lp[o + 1 + j] = none ? -INFINITY : best - lse;
    }
    if (none) {
      pv = -INFINITY;
      pi = 0x7fffffff;
    } else {
      pv = best;
      pi = bi;
    }
  }
}

// ---------------------------------------------------------------------------------------
// top-k / top-p / min-p of probability rows, in place.  One workgroup per row finds the cut-off
// VALUE by bisection on the bit pattern of a positive float (which orders like the float): the
// k-th largest value is the largest t with count(p >= t) >= k; the nucleus bound is the largest
// t with mass(p >= t) >= top_p x the mass left by top-k and min-p.  Entries equal to the cut-off
// are kept (ties stay together).  The row lives in three tiers: kFilterRegs values per thread in
// registers, up to kFilterLdsMax more per thread in LDS, the rest (vocabularies beyond ~123k
// entries) re-read from global memory (L2) in every pass.
struct FilterArgs {
  const int32_t* top_k;   // [n_seq] or null; <= 0: off
  const float* top_p;     // [n_seq] or null; >= 1: off
  const float* min_p;     // [n_seq] or null; <= 0: off
  int rows_per_seq;
};
constexpr int kFilterRegs = 84;
constexpr int kFilterLdsMax = 36;   // x 1024 threads x 4 B = 144 KiB

__global__ __launch_bounds__(kSampleThreads) void filter_probs_kernel(float* __restrict__ probs, int64_t ld,
                                                                      FilterArgs f, int V, int n_lds) {
  extern __shared__ float s_row[];   // [n_lds][kSampleThreads]
  __shared__ float s_f[kSampleWaves];
  __shared__ int s_i[kSampleWaves];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = r / f.rows_per_seq;
  const int k = f.top_k ? f.top_k[seq] : 0;
  const float P = f.top_p ? f.top_p[seq] : 1.f;
  const float mp = f.min_p ? f.min_p[seq] : 0.f;
  const bool use_k = k > 0 && k < V, use_p = P < 1.f, use_m = mp > 0.f;
  if (!use_k && !use_p && !use_m) return;   // block-uniform
  float* row = probs + (size_t)r * ld;
  const int vpt = (V + kSampleThreads - 1) / kSampleThreads;
  float v[kFilterRegs];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < kFilterRegs; ++j) {
    const int i = j * kSampleThreads + tid;
    v[j] = i < V ? row[i] : 0.f;
    mx = fmaxf(mx, v[j]);
  }
  for (int j = 0; j < n_lds; ++j) {
    const int i = (kFilterRegs + j) * kSampleThreads + tid;
    const float x = i < V ? row[i] : 0.f;
    s_row[j * kSampleThreads + tid] = x;   // read back by this thread only
    mx = fmaxf(mx, x);
  }
  for (int j = kFilterRegs + n_lds; j < vpt; ++j) {
    const int i = j * kSampleThreads + tid;
    if (i < V) mx = fmaxf(mx, row[i]);
  }
  // (count, mass) of the entries >= t, over the whole row
  auto stats_ge = [&](float t, int& count, float& mass) {
    int c = 0;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < kFilterRegs; ++j) {
      const bool in = v[j] >= t;
      c += in ? 1 : 0;
      m += in ? v[j] : 0.f;
    }
    for (int j = 0; j < n_lds; ++j) {
      const float x = s_row[j * kSampleThreads + tid];
      const bool in = x >= t;
      c += in ? 1 : 0;
      m += in ? x : 0.f;
    }
    for (int j = kFilterRegs + n_lds; j < vpt; ++j) {
      const int i = j * kSampleThreads + tid;
      const float x = i < V ? row[i] : 0.f;
      const bool in = x >= t;
      c += in ? 1 : 0;
      m += in ? x : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    m = wave_sum(m);
    __syncthreads();
    if (lane == 0) {
      s_i[wave] = c;
      s_f[wave] = m;
    }
    __syncthreads();
    count = 0;
    mass = 0.f;
    for (int w = 0; w < kSampleWaves; ++w) {
      count += s_i[w];
      mass += s_f[w];
    }
  };
  mx = wave_max(mx);
  if (lane == 0) s_f[wave] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < kSampleWaves; ++w) mx = fmaxf(mx, s_f[w]);
  if (!(mx > 0.f)) return;
  const unsigned mx_bits = __float_as_uint(mx);
  int cnt;
  float mass;
  float tau = 0.f;
  if (use_k) {
    unsigned cand = 0;
    for (int bit = 30; bit >= 0; --bit) {
      const unsigned t = cand | (1u << bit);
      if (t > mx_bits) continue;   // block-uniform
      stats_ge(__uint_as_float(t), cnt, mass);
      if (cnt >= k) cand = t;
    }
    tau = __uint_as_float(cand);
  }
  if (use_m) tau = fmaxf(tau, mp * mx);
  // the smallest positive float stands for "every entry with mass" when no other filter is on
  const float base = fmaxf(tau, __uint_as_float(1u));
  if (use_p) {
    stats_ge(base, cnt, mass);
    const float target = P * mass;
    unsigned cand = 0;
    for (int bit = 30; bit >= 0; --bit) {
      const unsigned t = cand | (1u << bit);
      if (t > mx_bits) continue;
      stats_ge(fmaxf(__uint_as_float(t), base), cnt, mass);
      if (mass >= target) cand = t;
    }
    tau = fmaxf(base, __uint_as_float(cand));
  } else {
    tau = base;
  }
  stats_ge(tau, cnt, mass);
  const float inv = mass > 0.f ? 1.f / mass : 0.f;
#pragma unroll
  for (int j = 0; j < kFilterRegs; ++j) {
    const int i = j * kSampleThreads + tid;
    if (i < V) row[i] = v[j] >= tau ? v[j] * inv : 0.f;
  }
  for (int j = 0; j < n_lds; ++j) {
    const int i = (kFilterRegs + j) * kSampleThreads + tid;
    const float x = s_row[j * kSampleThreads + tid];
    if (i < V) row[i] = x >= tau ? x * inv : 0.f;
  }
  for (int j = kFilterRegs + n_lds; j < vpt; ++j) {
    const int i = j * kSampleThreads + tid;
    if (i < V) {
      const float x = row[i];
      row[i] = x >= tau ? x * inv : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Split forms: a vocabulary row is cut into kRowParts slices, one workgroup each, so that a
// handful of rows still spreads over the chip; a per-row finishing step combines the slices.
// Scratch per row: kRowParts x {f64 mass, i32 last index with mass, f32 max, f32 sum}.
constexpr int kRowParts = 64;
constexpr int kPartThreads = 256;

struct RowScratch {
  double* mass;   // [R][kRowParts]
  int32_t* last;  // [R][kRowParts]
  float* pmax;    // [R][kRowParts]
  float* psum;    // [R][kRowParts]
};

__host__ __device__ inline size_t row_scratch_bytes(int R) {
  return (size_t)R * kRowParts * (sizeof(double) + sizeof(int32_t) + 2 * sizeof(float));
}

__host__ __device__ inline RowScratch row_scratch(void* base, int R) {
  RowScratch r;
  r.mass = reinterpret_cast<double*>(base);
  r.last = reinterpret_cast<int32_t*>(r.mass + (size_t)R * kRowParts);
  r.pmax = reinterpret_cast<float*>(r.last + (size_t)R * kRowParts);
  r.psum = r.pmax + (size_t)R * kRowParts;
  return r;
}

__device__ __forceinline__ void part_range(int part, int V, int& lo, int& hi) {
  const int chunk = ((V + kRowParts - 1) / kRowParts + 3) & ~3;
  lo = min(part * chunk, V);
  hi = min(lo + chunk, V);
}

// Block-wide sum of doubles / max of floats / max of ints for kPartThreads threads.
__device__ __forceinline__ double part_block_sum(double v, double* smem) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < kPartThreads / 64; ++w) t += smem[w];
  return t;
}

__global__ __launch_bounds__(kPartThreads) void softmax_partial_kernel(
    const float* __restrict__ logits, int64_t ld, RowTemps tm, RowScratch sc, int V) {
  __shared__ float s_f[kPartThreads / 64];
  __shared__ double s_d[kPartThreads / 64];
  const int part = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  int lo, hi;
  part_range(part, V, lo, hi);
  const float* row = logits + (size_t)r * ld;
  bool greedy;
  const float inv_t = row_inv_t(tm, r, greedy);   // greedy rows: slice maxima of the raw logits
  float mx = -INFINITY;
  for (int i = lo + tid; i < hi; i += kPartThreads) mx = fmaxf(mx, row[i] * inv_t);
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_f[tid >> 6] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < kPartThreads / 64; ++w) mx = fmaxf(mx, s_f[w]);
  double sum = 0.0;
  if (mx != -INFINITY)
    for (int i = lo + tid; i < hi; i += kPartThreads) sum += (double)expf(row[i] * inv_t - mx);
  const double total = part_block_sum(sum, s_d);
  if (tid == 0) {
    sc.pmax[(size_t)r * kRowParts + part] = mx;
    sc.psum[(size_t)r * kRowParts + part] = (float)total;
  }
}

__global__ __launch_bounds__(kPartThreads) void softmax_finish_kernel(
    const float* __restrict__ logits, int64_t ld, RowTemps tm, RowScratch sc,
    float* __restrict__ probs, int64_t ld_probs, int V) {
  const int part = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  bool greedy;
  const float inv_t = row_inv_t(tm, r, greedy);
  // every thread folds the row's 64 slice statistics the same way
  float mx = -INFINITY;
  for (int p = 0; p < kRowParts; ++p) mx = fmaxf(mx, sc.pmax[(size_t)r * kRowParts + p]);
  if (greedy) {   // all mass on the first maximum: the lowest slice that holds it, lowest index
    __shared__ int s_first[kPartThreads / 64];
    int first_part = 0;
    while (first_part < kRowParts - 1 && sc.pmax[(size_t)r * kRowParts + first_part] != mx) ++first_part;
    int lo, hi;
    part_range(part, V, lo, hi);
    const float* row = logits + (size_t)r * ld;
    float* out = probs + (size_t)r * ld_probs;
    int mine = INT_MAX;
    if (part == first_part)
      for (int i = lo + tid; i < hi; i += kPartThreads)
        if (row[i] == mx) mine = min(mine, i);
    for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
    if ((tid & 63) == 0) s_first[tid >> 6] = mine;
    __syncthreads();
    int best = s_first[0];
    for (int w = 1; w < kPartThreads / 64; ++w) best = min(best, s_first[w]);
    for (int i = lo + tid; i < hi; i += kPartThreads) out[i] = i == best ? 1.f : 0.f;
    return;
  }
  double total = 0.0;
  for (int p = 0; p < kRowParts; ++p) {
    const float pm = sc.pmax[(size_t)r * kRowParts + p];
    if (pm != -INFINITY) total += (double)sc.psum[(size_t)r * kRowParts + p] * (double)expf(pm - mx);
  }
  const float inv = (float)(1.0 / total);
  int lo, hi;
  part_range(part, V, lo, hi);
  const float* row = logits + (size_t)r * ld;
  float* out = probs + (size_t)r * ld_probs;
  for (int i = lo + tid; i < hi; i += kPartThreads) out[i] = expf(row[i] * inv_t - mx) * inv;
}

// First rejected position of sequence b (or L when every proposal is accepted).
__device__ __forceinline__ int first_reject(const float* __restrict__ pb, const float* __restrict__ qb,
                                            const int64_t* __restrict__ draft_ids,
                                            const float* __restrict__ u_accept, int b, int k, int L,
                                            int V) {
  int j = L;
  for (int i = L - 1; i >= 0; --i) {
    const int64_t x = draft_ids[(size_t)b * k + i];
    bool ok = false;
    if (x >= 0 && x < V) {
      const float uq = __fmul_rn(u_accept[(size_t)b * k + i], qb[(size_t)i * V + x]);
      ok = uq < pb[(size_t)i * V + x];
    }
    if (!ok) j = i;
  }
  return j;
}

struct ResidualRows {  // which rows the residual max(p - q, 0) of "row r" is taken from
  const float* p;      // p row r at p + r * ld_p
  int64_t ld_p;
  // rejection sampling (draft_ids != nullptr): r is a sequence; the rows are the target / draft
  // rows of its first rejected position
  const float* q;
  const int64_t* draft_ids;
  const float* u_accept;
  const int32_t* proposal_lens;
  int k;
};

__device__ __forceinline__ bool residual_rows(const ResidualRows& a, int r, int V, const float*& p,
                                              const float*& q, int& j, int& L) {
  if (!a.draft_ids) {
    p = a.p + (size_t)r * a.ld_p;
    q = nullptr;
    j = 0;
    L = 0;
    return true;
  }
  L = a.proposal_lens ? a.proposal_lens[r] : a.k;
  L = min(max(L, 0), a.k);
  const float* pb = a.p + (size_t)r * (a.k + 1) * V;
  const float* qb = a.q + (size_t)r * a.k * V;
  j = first_reject(pb, qb, a.draft_ids, a.u_accept, r, a.k, L, V);
  p = pb + (size_t)j * V;
  q = j < L ? qb + (size_t)j * V : nullptr;
  return !(j == L && L == a.k);  // all accepted: the bonus token, no residual draw
}

__global__ __launch_bounds__(kPartThreads) void residual_mass_kernel(ResidualRows a, RowScratch sc,
                                                                    int V) {
  __shared__ double s_d[kPartThreads / 64];
  __shared__ int s_i[kPartThreads / 64];
  const int part = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const float *p, *q;
  int j, L;
  if (!residual_rows(a, r, V, p, q, j, L)) return;
  int lo, hi;
  part_range(part, V, lo, hi);
  double local = 0.0;
  int last_pos = -1;
  for (int i = lo + tid * 4; i < hi; i += kPartThreads * 4) {
    const float4 pv = *reinterpret_cast<const float4*>(p + i);
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q) qv = *reinterpret_cast<const float4*>(q + i);
    const float e[4] = {fmaxf(pv.x - qv.x, 0.f), fmaxf(pv.y - qv.y, 0.f), fmaxf(pv.z - qv.z, 0.f),
                        fmaxf(pv.w - qv.w, 0.f)};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      local += (double)e[jj];
      if (e[jj] > 0.f) last_pos = i + jj;
    }
  }
  const double total = part_block_sum(local, s_d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last_pos = max(last_pos, __shfl_xor(last_pos, o, 64));
  if ((tid & 63) == 0) s_i[tid >> 6] = last_pos;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kPartThreads / 64; ++w) last_pos = max(last_pos, s_i[w]);
    sc.mass[(size_t)r * kRowParts + part] = total;
    sc.last[(size_t)r * kRowParts + part] = last_pos;
  }
}

// Inverse-CDF pick for row r from the slice masses: the slice whose running sum first exceeds
// u * total, then a scan inside it (same rule as block_sample_residual).  Whole block.
__device__ int pick_from_parts(const float* __restrict__ p, const float* __restrict__ q, int V,
                               float u, const RowScratch& sc, int r) {
  __shared__ double s_prefix[kRowParts + 1];
  __shared__ double s_w[kPartThreads / 64];
  __shared__ int s_found, s_part;
  __shared__ float s_f[kPartThreads / 64];
  __shared__ int s_i[kPartThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    double c = 0.0;
    for (int pp = 0; pp < kRowParts; ++pp) {
      s_prefix[pp] = c;
      c += sc.mass[(size_t)r * kRowParts + pp];
    }
    s_prefix[kRowParts] = c;
    s_found = 0x7fffffff;
  }
  __syncthreads();
  const double total = s_prefix[kRowParts];
  if (!(total > 0.0)) {  // no mass: argmax(p), whole row
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += kPartThreads)
      if (p[i] > best) { best = p[i]; bi = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_f[wave] = best; s_i[wave] = bi; }
    __syncthreads();
    best = s_f[0];
    bi = s_i[0];
    for (int w = 1; w < kPartThreads / 64; ++w)
      if (s_f[w] > best || (s_f[w] == best && s_i[w] < bi)) { best = s_f[w]; bi = s_i[w]; }
    return bi == 0x7fffffff ? 0 : bi;
  }
  const double thr = (double)u * total;
  if (tid == 0) {
    int part = -1;
    for (int pp = 0; pp < kRowParts; ++pp)
      if (s_prefix[pp] <= thr && s_prefix[pp + 1] > thr) { part = pp; break; }
    s_part = part;
  }
  __syncthreads();
  const int part = s_part;
  if (part >= 0) {
    int lo, hi;
    part_range(part, V, lo, hi);
    double carry = s_prefix[part];
    for (int base = lo; base < hi; base += kPartThreads * 4) {
      const int i = base + tid * 4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (i < hi) {
        const float4 pv = *reinterpret_cast<const float4*>(p + i);
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q) qv = *reinterpret_cast<const float4*>(q + i);
        e[0] = fmaxf(pv.x - qv.x, 0.f);
        e[1] = fmaxf(pv.y - qv.y, 0.f);
        e[2] = fmaxf(pv.z - qv.z, 0.f);
        e[3] = fmaxf(pv.w - qv.w, 0.f);
      }
      const double s4 = (double)e[0] + (double)e[1] + (double)e[2] + (double)e[3];
      double inc = s4;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      __syncthreads();
      if (lane == 63) s_w[wave] = inc;
      __syncthreads();
      double wave_prefix = 0.0, tile_total = 0.0;
      for (int w = 0; w < kPartThreads / 64; ++w) {
        if (w < wave) wave_prefix += s_w[w];
        tile_total += s_w[w];
      }
      const double excl = carry + wave_prefix + (inc - s4);
      if (i < hi && excl <= thr && excl + s4 > thr) {
        double c = excl;
        int idx = i + 3;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          c += (double)e[jj];
          if (c > thr) { idx = i + jj; break; }
        }
        atomicMin(&s_found, idx);
      }
      carry += tile_total;
      __syncthreads();
      if (s_found != 0x7fffffff) break;
    }
  }
  const int found = s_found;
  if (found != 0x7fffffff) return found;
  int last_pos = -1;  // rounding left no index: the last one with mass
  for (int pp = 0; pp < kRowParts; ++pp) last_pos = max(last_pos, sc.last[(size_t)r * kRowParts + pp]);
  return max(last_pos, 0);
}

__global__ __launch_bounds__(kPartThreads) void sample_finish_kernel(
    ResidualRows a, RowScratch sc, const float* __restrict__ uniforms, int64_t* __restrict__ ids,
    int V) {
  const int r = blockIdx.x;
  const int idx = pick_from_parts(a.p + (size_t)r * a.ld_p, nullptr, V, uniforms[r], sc, r);
  if (threadIdx.x == 0) ids[r] = idx;
}

__global__ __launch_bounds__(kPartThreads) void rejection_finish_kernel(
    ResidualRows a, RowScratch sc, const int64_t* __restrict__ bonus_ids,
    const float* __restrict__ u_recover, int64_t* __restrict__ out_ids,
    unsigned long long* __restrict__ counters, int V) {
  const int b = blockIdx.x, tid = threadIdx.x, k = a.k;
  const float *p, *q;
  int j, L;
  int64_t tok;
  if (residual_rows(a, b, V, p, q, j, L))
    tok = pick_from_parts(p, q, V, u_recover[(size_t)b * (k + 1) + j], sc, b);
  else
    tok = bonus_ids[b];
  if (tid <= k) {
    int64_t v = -1;
    if (tid < j) v = a.draft_ids[(size_t)b * k + tid];
    else if (tid == j) v = tok;
    out_ids[(size_t)b * (k + 1) + tid] = v;
  }
  if (tid == 0 && counters) {
    atomicAdd(&counters[0], (unsigned long long)j);
    atomicAdd(&counters[1], (unsigned long long)(j + 1));
    atomicAdd(&counters[2], (unsigned long long)L);
  }
}

// u[i] = 24-bit uniform in [0, 1) from (seed, *counter, i); one workgroup, which then advances
// the counter, so a replayed graph draws fresh numbers every time.
__global__ __launch_bounds__(1024) void fill_uniforms_kernel(float* __restrict__ out, int n,
                                                            unsigned long long seed,
                                                            unsigned long long* __restrict__ counter) {
  const unsigned long long c = *counter;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    unsigned long long z = seed ^ (c * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)i << 32 | (unsigned)i);
    z += 0x9e3779b97f4a7c15ull;   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    out[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) *counter = c + 1;
}

__device__ __forceinline__ float uniform24(unsigned long long seed, unsigned long long c,
                                          unsigned long long i) {
  unsigned long long z = seed ^ (c * 0x9e3779b97f4a7c15ull) ^ (i << 32 | (unsigned)i);
  z += 0x9e3779b97f4a7c15ull;   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

struct UniformRegions {
  sd_uniform_region r[SD_UNIFORM_REGIONS_MAX];
  int n;
};

// One wave per sequence: lane j writes the sequence's j-th, (j+64)-th, ... number of the step.
__global__ __launch_bounds__(SD_WAVE) void fill_uniforms_seq_kernel(
    float* __restrict__ out, UniformRegions regs, const unsigned long long* __restrict__ seeds,
    unsigned long long* __restrict__ counters) {
  const int b = blockIdx.x;
  const unsigned long long seed = seeds[b], c = counters[b];
  int j0 = 0;
  for (int r = 0; r < regs.n; ++r) {
    const sd_uniform_region& g = regs.r[r];
    for (int i = threadIdx.x; i < g.n; i += SD_WAVE)
      out[g.offset + (int64_t)i * g.stride_j + (int64_t)b * g.stride_b] =
          uniform24(seed, c, (unsigned long long)(j0 + i));
    j0 += g.n;
  }
  if (threadIdx.x == 0) counters[b] = c + 1;
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

__global__ void store_proposal_kernel(const int64_t* __restrict__ ids,
                                      int32_t* __restrict__ tok_buf, int k, int step, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  tok_buf[(size_t)b * (k + 2) + 2 + step] = (int32_t)ids[b];
}

// Benchmark control (sd_debug_force_proposals): overwrites the k proposals of sequence b so
// that exactly a = pattern[step_counter[b] % len] of them will be accepted: proposals 0..a-1
// become the true tokens of their positions, proposal a (if a < k) a token that is certainly
// not the true one; later proposals keep what the draft model made.
struct ForcePattern {
  uint8_t n_accept[SD_DEBUG_PATTERN_MAX];
  int len;
};
__global__ void force_proposals_kernel(int32_t* __restrict__ tok_buf, int k,
                                       const int32_t* __restrict__ seq_lens,
                                       const int32_t* __restrict__ truth, int64_t truth_stride,
                                       ForcePattern pat, int32_t* __restrict__ step_counter,
                                       int vocab, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int t = step_counter[b];
  step_counter[b] = t + 1;
  const int a = min((int)pat.n_accept[t % pat.len], k);
  const int n = seq_lens[b];  // position of proposal 0
  for (int i = 0; i <= a && i < k; ++i) {
    if (n + i >= truth_stride) break;
    const int32_t tok = truth[(size_t)b * truth_stride + n + i];
    tok_buf[(size_t)b * (k + 2) + 2 + i] = i < a ? tok : (tok + 1) % vocab;
  }
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

__global__ void set_proposals_kernel(const int32_t* __restrict__ props,
                                     int32_t* __restrict__ tok_buf, int B, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * k) return;
  const int b = i / k, j = i % k;
  tok_buf[(size_t)b * (k + 2) + 2 + j] = props[i];
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

// Stage 1 of a split row argmax: wave (part, row) scans one of the row's kArgmaxParts slices.
// (The LM head of a greedy pass can leave the same slots itself: sd_norm_linear_argmax, one slot
// per workgroup of its launch, which is why there are as many slots as that launch has
// workgroups at most.)
__global__ __launch_bounds__(SD_WAVE) void argmax_partial_kernel(
    const float* __restrict__ logits, int64_t ld, float* __restrict__ part_val,
    int32_t* __restrict__ part_idx, int V) {
  const int part = blockIdx.x, r = blockIdx.y;
  const int chunk = ((V + kArgmaxParts - 1) / kArgmaxParts + 3) & ~3;
  const int lo = part * chunk, hi = min(lo + chunk, V);
  const float* row = logits + (size_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x * 4; i < hi; i += SD_WAVE * 4) {
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
  if (threadIdx.x == 0) {
    part_val[(size_t)r * kArgmaxParts + part] = best;
    part_idx[(size_t)r * kArgmaxParts + part] = bi;
  }
}

// Stage 2, one wave per row: the smallest index among the row's maxima (slots in any order).
__device__ __forceinline__ int wave_argmax_finish(const float* __restrict__ part_val,
                                                  const int32_t* __restrict__ part_idx, int row,
                                                  int lane) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < kArgmaxParts / SD_WAVE; ++j) {
    const float v = part_val[(size_t)row * kArgmaxParts + j * SD_WAVE + lane];
    const int id = part_idx[(size_t)row * kArgmaxParts + j * SD_WAVE + lane];
    if (v > best || (v == best && id < bi)) { best = v; bi = id; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  return bi == 0x7fffffff ? 0 : bi;
}

// Slice maxima of n_ranks vocabulary shards (each rank's indices local to its shard) -> slice
// maxima over the whole vocabulary: slot (row, part) keeps the largest value, ties to the lowest
// global index, so a finish over the combined slots is the first argmax of the full row.
__global__ __launch_bounds__(256) void argmax_combine_kernel(
    const float* __restrict__ vals, const int32_t* __restrict__ idx, int n_ranks, int n,
    int shard_vocab, int64_t rank_stride, float* __restrict__ out_val,
    int32_t* __restrict__ out_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int r = 0; r < n_ranks; ++r) {
    const float v = vals[(size_t)r * rank_stride + i];
    int id = idx[(size_t)r * rank_stride + i];
    if (id != 0x7fffffff) id += r * shard_vocab;
    if (v > best || (v == best && id < bi)) { best = v; bi = id; }
  }
  out_val[i] = best;
  out_idx[i] = bi;
}

__global__ __launch_bounds__(SD_WAVE) void argmax_from_partials_kernel(
    const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
    int64_t* __restrict__ ids) {
  const int id = wave_argmax_finish(part_val, part_idx, blockIdx.x, threadIdx.x);
  if (threadIdx.x == 0) ids[blockIdx.x] = id;
}

__global__ __launch_bounds__(SD_WAVE) void propose_from_partials_kernel(
    const float* __restrict__ part_val, const int32_t* __restrict__ part_idx, int row_offset,
    int row_step, int32_t* __restrict__ tok_buf, int k, int step) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int id = wave_argmax_finish(part_val, part_idx, row_offset + b * row_step, lane);
  if (lane == 0) tok_buf[(size_t)b * (k + 2) + 2 + step] = id;
}

// Greedy verification straight from the target's argmax partials: wave j finishes row j
// of the sequence, wave 0 then matches the proposals in tok_buf and applies the result.
__global__ __launch_bounds__(SD_WAVE * 16) void verify_from_partials_kernel(
    const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
    int32_t* __restrict__ tok_buf, const int32_t* __restrict__ proposal_lens,
    int64_t* __restrict__ out_ids, unsigned long long* __restrict__ counters,
    int32_t* __restrict__ seq_lens, int32_t* __restrict__ gen_tokens, int64_t gen_stride,
    int gen_capacity, int32_t* __restrict__ gen_counts, int32_t* __restrict__ step_out, int k) {
  __shared__ int s_target[16];
  __shared__ int64_t s_out[16];
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

extern "C" int sd_typical_acceptance(const float* target_probs, const int64_t* bonus_ids,
                                     const int64_t* draft_ids, const int32_t* proposal_lens,
                                     float posterior_threshold, float posterior_alpha,
                                     int64_t* out_ids, int64_t* counters, int B, int k, int V,
                                     sd_stream_t stream) {
  SD_REQUIRE(B >= 0 && k >= 1 && k <= 63 && V >= 1, SD_ERR_INVALID,
             "sd_typical_acceptance: need B>=0, 1<=k<=63, V>=1 (B=%d k=%d V=%d)", B, k, V);
  SD_REQUIRE(V % 4 == 0, SD_ERR_UNSUPPORTED, "sd_typical_acceptance: V %% 4 != 0 (V=%d)", V);
  SD_REQUIRE(target_probs && bonus_ids && draft_ids && out_ids, SD_ERR_INVALID,
             "sd_typical_acceptance: null pointer");
  SD_REQUIRE(sd_aligned16(target_probs), SD_ERR_INVALID,
             "sd_typical_acceptance: probability rows must be 16-byte aligned");
  SD_REQUIRE(posterior_threshold >= 0.f && posterior_alpha >= 0.f, SD_ERR_INVALID,
             "sd_typical_acceptance: negative threshold or alpha");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(typical_acceptance_kernel, dim3(B), dim3(kSampleThreads), 0,
                     (hipStream_t)stream, target_probs, bonus_ids, draft_ids, proposal_lens,
                     posterior_threshold, posterior_alpha, out_ids,
                     (unsigned long long*)counters, k, V);
  SD_CHECK_LAUNCH("sd_typical_acceptance");
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

extern "C" int sd_map_tokens(int32_t* tokens, int64_t stride, int col, const int32_t* map, int n_map,
                             int B, sd_stream_t stream) {
  SD_REQUIRE(tokens && map && n_map >= 1 && B >= 0 && col >= 0 && col < stride, SD_ERR_INVALID,
             "sd_map_tokens: bad arguments (n_map=%d col=%d)", n_map, col);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(map_tokens_kernel, dim3((B + SD_WAVE - 1) / SD_WAVE), dim3(SD_WAVE), 0,
                     (hipStream_t)stream, tokens, stride, col, map, n_map, B);
  SD_CHECK_LAUNCH("sd_map_tokens");
  return SD_OK;
}

extern "C" int sd_tree_prepare(const int32_t* tok_buf, int64_t tok_stride, int tok_col,
                               int32_t* tree_tok, const int32_t* tree_depth,
                               const int32_t* seq_lens, int pos_offset, int32_t* rope_pos, int B,
                               int q_len, sd_stream_t stream) {
  SD_REQUIRE(tok_buf && tree_tok && tree_depth && seq_lens && rope_pos, SD_ERR_INVALID,
             "sd_tree_prepare: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && q_len <= SD_WAVE && tok_col >= 0 && tok_col < tok_stride,
             SD_ERR_INVALID, "sd_tree_prepare: bad shape (q_len=%d)", q_len);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(tree_prepare_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream, tok_buf,
                     tok_stride, tok_col, tree_tok, tree_depth, seq_lens, pos_offset, rope_pos, q_len);
  SD_CHECK_LAUNCH("sd_tree_prepare");
  return SD_OK;
}

extern "C" int sd_tree_accept_greedy(const int64_t* target_ids, const int32_t* tree_tok,
                                     const int32_t* tree_parent, const int32_t* tree_len,
                                     int q_len, int k, int64_t* out_ids, int32_t* path,
                                     int32_t* n_path, int64_t* counters, int B,
                                     sd_stream_t stream) {
  SD_REQUIRE(target_ids && tree_tok && tree_parent && tree_len && out_ids && path && n_path,
             SD_ERR_INVALID, "sd_tree_accept_greedy: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && q_len <= SD_WAVE && k >= 1 && k < SD_WAVE, SD_ERR_INVALID,
             "sd_tree_accept_greedy: bad shape (q_len=%d k=%d)", q_len, k);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(tree_accept_greedy_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream,
                     target_ids, tree_tok, tree_parent, tree_len, q_len, k, out_ids, path, n_path,
                     (unsigned long long*)counters);
  SD_CHECK_LAUNCH("sd_tree_accept_greedy");
  return SD_OK;
}

extern "C" int sd_tree_accept_sample(const float* target_probs, int64_t ld, const int32_t* tree_tok,
                                     const int32_t* tree_parent, const int32_t* tree_len, int q_len,
                                     int k, const float* u_accept, const float* u_final,
                                     int64_t* out_ids, int32_t* path, int32_t* n_path,
                                     int64_t* counters, int B, int V, sd_stream_t stream) {
  SD_REQUIRE(target_probs && tree_tok && tree_parent && tree_len && u_accept && u_final && out_ids &&
                 path && n_path,
             SD_ERR_INVALID, "sd_tree_accept_sample: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && q_len <= SD_WAVE && k >= 1 && k < SD_WAVE && V >= 4 && ld >= V,
             SD_ERR_INVALID, "sd_tree_accept_sample: bad shape (q_len=%d k=%d V=%d)", q_len, k, V);
  SD_REQUIRE(V % 4 == 0 && ld % 4 == 0 && sd_aligned16(target_probs), SD_ERR_UNSUPPORTED,
             "sd_tree_accept_sample: V and ld must be multiples of 4, rows 16-byte aligned");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(tree_accept_sample_kernel, dim3(B), dim3(kSampleThreads), 0, (hipStream_t)stream,
                     target_probs, ld, tree_tok, tree_parent, tree_len, q_len, k, u_accept, u_final, V,
                     out_ids, path, n_path, (unsigned long long*)counters);
  SD_CHECK_LAUNCH("sd_tree_accept_sample");
  return SD_OK;
}

extern "C" int sd_topk_tokens(const float* logits, int64_t ld, int V, int row_offset, int row_step,
                              int rows, int b, int32_t* out_tok, int64_t out_stride, int out_col,
                              int B, sd_stream_t stream) {
  SD_REQUIRE(logits && out_tok && V >= 1 && ld >= V && row_offset >= 0 && row_step >= 1 && B >= 0 &&
                 rows >= 1 && rows <= row_step,
             SD_ERR_INVALID, "sd_topk_tokens: bad arguments (B=%d V=%d rows=%d)", B, V, rows);
  SD_REQUIRE(b >= 1 && b <= SD_TOPK_TOKENS_MAX && b <= V && out_col >= 0 &&
                 out_stride >= out_col + (int64_t)rows * b,
             SD_ERR_INVALID, "sd_topk_tokens: b=%d outside 1..%d or the output row is too short", b,
             SD_TOPK_TOKENS_MAX);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(topk_tokens_kernel, dim3(B, rows), dim3(kSampleThreads), 0, (hipStream_t)stream,
                     logits, ld, V, row_offset, row_step, b, out_tok, out_stride, out_col);
  SD_CHECK_LAUNCH("sd_topk_tokens");
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
                     (hipStream_t)stream, logits, ld, RowTemps{nullptr, 1, 1.0f / temperature}, probs,
                     ld_probs, V);
  SD_CHECK_LAUNCH("sd_softmax_rows");
  return SD_OK;
}

extern "C" int sd_softmax_rows_seq(const float* logits, int64_t ld, const float* temperatures,
                                   int rows_per_seq, float* probs, int64_t ld_probs, int R, int V,
                                   void* scratch, sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 1 && ld >= V && ld_probs >= V && logits && probs && temperatures &&
                 rows_per_seq >= 1,
             SD_ERR_INVALID, "sd_softmax_rows_seq: bad arguments (R=%d V=%d rows_per_seq=%d)", R, V,
             rows_per_seq);
  if (R == 0) return SD_OK;
  const RowTemps tm{temperatures, rows_per_seq, 1.f};
  if (!scratch) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(R), dim3(kSampleThreads), 0, (hipStream_t)stream,
                       logits, ld, tm, probs, ld_probs, V);
    SD_CHECK_LAUNCH("sd_softmax_rows_seq");
    return SD_OK;
  }
  const RowScratch sc = row_scratch(scratch, R);
  hipLaunchKernelGGL(softmax_partial_kernel, dim3(kRowParts, R), dim3(kPartThreads), 0,
                     (hipStream_t)stream, logits, ld, tm, sc, V);
  SD_CHECK_LAUNCH("sd_softmax_rows_seq");
  hipLaunchKernelGGL(softmax_finish_kernel, dim3(kRowParts, R), dim3(kPartThreads), 0,
                     (hipStream_t)stream, logits, ld, tm, sc, probs, ld_probs, V);
  SD_CHECK_LAUNCH("sd_softmax_rows_seq");
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

extern "C" int sd_filter_probs_seq(float* probs, int64_t ld, const int32_t* top_k, const float* top_p,
                                   const float* min_p, int rows_per_seq, int R, int V,
                                   sd_stream_t stream) {
  SD_REQUIRE(probs && R >= 0 && V >= 1 && ld >= V && rows_per_seq >= 1, SD_ERR_INVALID,
             "sd_filter_probs_seq: bad arguments (R=%d V=%d rows_per_seq=%d)", R, V, rows_per_seq);
  if (R == 0 || (!top_k && !top_p && !min_p)) return SD_OK;
  const FilterArgs f{top_k, top_p, min_p, rows_per_seq};
  const int vpt = (V + kSampleThreads - 1) / kSampleThreads;
  int n_lds = vpt - kFilterRegs;
  n_lds = n_lds < 0 ? 0 : (n_lds > kFilterLdsMax ? kFilterLdsMax : n_lds);
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(filter_probs_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kFilterLdsMax * kSampleThreads * (int)sizeof(float));
  });
  SD_REQUIRE(attr_err == hipSuccess, SD_ERR_HIP, "sd_filter_probs_seq: hipFuncSetAttribute: %s",
             hipGetErrorString(attr_err));
  hipLaunchKernelGGL(filter_probs_kernel, dim3(R), dim3(kSampleThreads),
                     (size_t)n_lds * kSampleThreads * sizeof(float), (hipStream_t)stream, probs, ld, f, V,
                     n_lds);
  SD_CHECK_LAUNCH("sd_filter_probs_seq");
  return SD_OK;
}

extern "C" int sd_add_logit_bias(float* logits, int64_t ld, const float* bias, int64_t ld_bias,
                                 const int32_t* enabled, int rows_per_seq, int R, int V,
                                 sd_stream_t stream) {
  SD_REQUIRE(logits && bias && enabled && R >= 0 && V >= 1 && ld >= V && ld_bias >= V && rows_per_seq >= 1,
             SD_ERR_INVALID, "sd_add_logit_bias: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(sd_aligned16(logits) && sd_aligned16(bias) && ld % 4 == 0 && ld_bias % 4 == 0,
             SD_ERR_UNSUPPORTED, "sd_add_logit_bias: rows must be 16-byte aligned");
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(add_logit_bias_kernel, dim3((V + 1023) / 1024, R), dim3(256), 0, (hipStream_t)stream,
                     logits, ld, bias, ld_bias, enabled, rows_per_seq, V);
  SD_CHECK_LAUNCH("sd_add_logit_bias");
  return SD_OK;
}

extern "C" int sd_token_logprobs(const float* logits, int64_t ld, const int32_t* tokens, int R, int V,
                                 int n_top, int32_t* out_ids, float* out_logprobs, int32_t* out_ranks,
                                 sd_stream_t stream) {
  SD_REQUIRE(logits && tokens && out_ids && out_logprobs && out_ranks && R >= 0 && V >= 1 && ld >= V,
             SD_ERR_INVALID, "sd_token_logprobs: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(n_top >= 0 && n_top <= SD_LOGPROBS_TOP_MAX, SD_ERR_INVALID,
             "sd_token_logprobs: n_top %d outside 0..%d", n_top, SD_LOGPROBS_TOP_MAX);
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(token_logprobs_kernel, dim3(R), dim3(kSampleThreads), 0, (hipStream_t)stream,
                     logits, ld, tokens, V, n_top, out_ids, out_logprobs, out_ranks);
  SD_CHECK_LAUNCH("sd_token_logprobs");
  return SD_OK;
}

extern "C" size_t sd_sampling_scratch_bytes(int R) {
  return R > 0 ? (row_scratch_bytes(R) + 255) & ~(size_t)255 : 0;
}

extern "C" int sd_softmax_rows_split(const float* logits, int64_t ld, float temperature,
                                     float* probs, int64_t ld_probs, int R, int V, void* scratch,
                                     sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 1 && ld >= V && ld_probs >= V && logits && probs && scratch &&
                 temperature > 0.f,
             SD_ERR_INVALID, "sd_softmax_rows_split: bad arguments (R=%d V=%d T=%g)", R, V,
             (double)temperature);
  if (R == 0) return SD_OK;
  const RowScratch sc = row_scratch(scratch, R);
  const RowTemps tm{nullptr, 1, 1.0f / temperature};
  hipLaunchKernelGGL(softmax_partial_kernel, dim3(kRowParts, R), dim3(kPartThreads), 0,
                     (hipStream_t)stream, logits, ld, tm, sc, V);
  SD_CHECK_LAUNCH("sd_softmax_rows_split");
  hipLaunchKernelGGL(softmax_finish_kernel, dim3(kRowParts, R), dim3(kPartThreads), 0,
                     (hipStream_t)stream, logits, ld, tm, sc, probs, ld_probs, V);
  SD_CHECK_LAUNCH("sd_softmax_rows_split");
  return SD_OK;
}

extern "C" int sd_sample_rows_split(const float* probs, int64_t ld, const float* uniforms,
                                    int64_t* ids, int R, int V, void* scratch,
                                    sd_stream_t stream) {
  SD_REQUIRE(R >= 0 && V >= 4 && ld >= V && probs && uniforms && ids && scratch, SD_ERR_INVALID,
             "sd_sample_rows_split: bad arguments (R=%d V=%d)", R, V);
  SD_REQUIRE(V % 4 == 0 && ld % 4 == 0 && sd_aligned16(probs), SD_ERR_UNSUPPORTED,
             "sd_sample_rows_split: V and ld must be multiples of 4, rows 16-byte aligned");
  if (R == 0) return SD_OK;
  const RowScratch sc = row_scratch(scratch, R);
  const ResidualRows a{probs, ld, nullptr, nullptr, nullptr, nullptr, 0};
  hipLaunchKernelGGL(residual_mass_kernel, dim3(kRowParts, R), dim3(kPartThreads), 0,
                     (hipStream_t)stream, a, sc, V);
  SD_CHECK_LAUNCH("sd_sample_rows_split");
  hipLaunchKernelGGL(sample_finish_kernel, dim3(R), dim3(kPartThreads), 0, (hipStream_t)stream, a,
                     sc, uniforms, ids, V);
  SD_CHECK_LAUNCH("sd_sample_rows_split");
  return SD_OK;
}

extern "C" int sd_rejection_sample_split(const float* target_probs, const int64_t* bonus_ids,
                                         const float* draft_probs, const int64_t* draft_ids,
                                         const float* u_accept, const float* u_recover,
                                         const int32_t* proposal_lens, int64_t* out_ids,
                                         int64_t* counters, int B, int k, int V, void* scratch,
                                         sd_stream_t stream) {
  SD_REQUIRE(B >= 0 && k >= 1 && k <= 63 && V >= 4, SD_ERR_INVALID,
             "sd_rejection_sample_split: need B>=0, 1<=k<=63, V>=4 (B=%d k=%d V=%d)", B, k, V);
  SD_REQUIRE(V % 4 == 0, SD_ERR_UNSUPPORTED, "sd_rejection_sample_split: V %% 4 != 0 (V=%d)", V);
  SD_REQUIRE(target_probs && bonus_ids && draft_probs && draft_ids && u_accept && u_recover &&
                 out_ids && scratch, SD_ERR_INVALID, "sd_rejection_sample_split: null pointer");
  SD_REQUIRE(sd_aligned16(target_probs) && sd_aligned16(draft_probs), SD_ERR_INVALID,
             "sd_rejection_sample_split: probability rows must be 16-byte aligned");
  if (B == 0) return SD_OK;
  const RowScratch sc = row_scratch(scratch, B);
  const ResidualRows a{target_probs, 0, draft_probs, draft_ids, u_accept, proposal_lens, k};
  hipLaunchKernelGGL(residual_mass_kernel, dim3(kRowParts, B), dim3(kPartThreads), 0,
                     (hipStream_t)stream, a, sc, V);
  SD_CHECK_LAUNCH("sd_rejection_sample_split");
  hipLaunchKernelGGL(rejection_finish_kernel, dim3(B), dim3(kPartThreads), 0, (hipStream_t)stream,
                     a, sc, bonus_ids, u_recover, out_ids,
                     reinterpret_cast<unsigned long long*>(counters), V);
  SD_CHECK_LAUNCH("sd_rejection_sample_split");
  return SD_OK;
}

extern "C" int sd_fill_uniforms(float* out, int n, uint64_t seed, uint64_t* counter,
                                sd_stream_t stream) {
  SD_REQUIRE(out && counter && n >= 0, SD_ERR_INVALID, "sd_fill_uniforms: bad arguments");
  if (n == 0) return SD_OK;
  hipLaunchKernelGGL(fill_uniforms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, n,
                     (unsigned long long)seed, reinterpret_cast<unsigned long long*>(counter));
  SD_CHECK_LAUNCH("sd_fill_uniforms");
  return SD_OK;
}

extern "C" int sd_fill_uniforms_seq(float* out, const sd_uniform_region* regions, int n_regions,
                                    const uint64_t* seeds, uint64_t* counters, int B,
                                    sd_stream_t stream) {
  SD_REQUIRE(out && regions && seeds && counters && B >= 0 && n_regions >= 1 &&
                 n_regions <= SD_UNIFORM_REGIONS_MAX,
             SD_ERR_INVALID, "sd_fill_uniforms_seq: bad arguments (regions=%d)", n_regions);
  UniformRegions regs;
  regs.n = n_regions;
  for (int r = 0; r < n_regions; ++r) {
    SD_REQUIRE(regions[r].n >= 0 && regions[r].offset >= 0, SD_ERR_INVALID,
               "sd_fill_uniforms_seq: region %d has a negative size or offset", r);
    regs.r[r] = regions[r];
  }
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(fill_uniforms_seq_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream, out,
                     regs, reinterpret_cast<const unsigned long long*>(seeds),
                     reinterpret_cast<unsigned long long*>(counters));
  SD_CHECK_LAUNCH("sd_fill_uniforms_seq");
  return SD_OK;
}

extern "C" int sd_store_proposal(const int64_t* ids, int32_t* tok_buf, int k, int step, int B,
                                 sd_stream_t stream) {
  SD_REQUIRE(ids && tok_buf && B >= 0 && k >= 1 && step >= 0 && step < k,
             SD_ERR_INVALID, "sd_store_proposal: bad arguments (B=%d k=%d step=%d)", B, k, step);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(store_proposal_kernel, dim3((B + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, ids, tok_buf, k, step, B);
  SD_CHECK_LAUNCH("sd_store_proposal");
  return SD_OK;
}

extern "C" int sd_debug_force_proposals(int32_t* tok_buf, int k, const int32_t* seq_lens,
                                        const int32_t* truth, int64_t truth_stride,
                                        const uint8_t* pattern, int pattern_len,
                                        int32_t* step_counter, int vocab, int B,
                                        sd_stream_t stream) {
  SD_REQUIRE(tok_buf && seq_lens && truth && pattern && step_counter && B >= 0 && k >= 1 &&
                 vocab >= 2 && truth_stride >= 1,
             SD_ERR_INVALID, "sd_debug_force_proposals: bad arguments (B=%d k=%d)", B, k);
  SD_REQUIRE(pattern_len >= 1 && pattern_len <= SD_DEBUG_PATTERN_MAX, SD_ERR_INVALID,
             "sd_debug_force_proposals: pattern length %d outside 1..%d", pattern_len,
             SD_DEBUG_PATTERN_MAX);
  if (B == 0) return SD_OK;
  ForcePattern pat;
  for (int i = 0; i < SD_DEBUG_PATTERN_MAX; ++i) pat.n_accept[i] = i < pattern_len ? pattern[i] : 0;
  pat.len = pattern_len;
  hipLaunchKernelGGL(force_proposals_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     tok_buf, k, seq_lens, truth, truth_stride, pat, step_counter, vocab, B);
  SD_CHECK_LAUNCH("sd_debug_force_proposals");
  return SD_OK;
}

extern "C" int sd_set_proposals(const int32_t* proposals, int32_t* tok_buf, int B, int k,
                                sd_stream_t stream) {
  SD_REQUIRE(proposals && tok_buf && B >= 0 && k >= 1, SD_ERR_INVALID,
             "sd_set_proposals: bad arguments");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(set_proposals_kernel, dim3((B * k + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, proposals, tok_buf, B, k);
  SD_CHECK_LAUNCH("sd_set_proposals");
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
  hipLaunchKernelGGL(argmax_partial_kernel, dim3(kArgmaxParts, R), dim3(SD_WAVE), 0,
                     (hipStream_t)stream, logits, ld, part_val, part_idx, V);
  SD_CHECK_LAUNCH("sd_argmax_partial");
  return SD_OK;
}

extern "C" int sd_argmax_combine(const float* part_val, const int32_t* part_idx, int n_ranks,
                                 int R, int shard_vocab, int64_t rank_stride, float* out_val,
                                 int32_t* out_idx, sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx && out_val && out_idx, SD_ERR_INVALID,
             "sd_argmax_combine: null pointer");
  SD_REQUIRE(n_ranks >= 1 && R >= 0 && shard_vocab >= 1 &&
                 rank_stride >= (int64_t)R * kArgmaxParts, SD_ERR_INVALID,
             "sd_argmax_combine: bad arguments (ranks=%d R=%d shard=%d)", n_ranks, R, shard_vocab);
  if (R == 0) return SD_OK;
  const int n = R * kArgmaxParts;
  hipLaunchKernelGGL(argmax_combine_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, part_val, part_idx, n_ranks, n, shard_vocab,
                     rank_stride, out_val, out_idx);
  SD_CHECK_LAUNCH("sd_argmax_combine");
  return SD_OK;
}

extern "C" int sd_argmax_from_partials(const float* part_val, const int32_t* part_idx,
                                       int64_t* ids, int R, sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx && ids && R >= 0, SD_ERR_INVALID,
             "sd_argmax_from_partials: bad arguments");
  if (R == 0) return SD_OK;
  hipLaunchKernelGGL(argmax_from_partials_kernel, dim3(R), dim3(SD_WAVE), 0, (hipStream_t)stream,
                     part_val, part_idx, ids);
  SD_CHECK_LAUNCH("sd_argmax_from_partials");
  return SD_OK;
}

extern "C" int sd_propose_from_partials(const float* part_val, const int32_t* part_idx,
                                        int row_offset, int row_step, int32_t* tok_buf, int k,
                                        int step, int B, sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx && tok_buf && B >= 0 && k >= 1 && step >= 0 && step < k &&
                 row_offset >= 0 && row_step >= 1, SD_ERR_INVALID,
             "sd_propose_from_partials: bad arguments (B=%d k=%d step=%d)", B, k, step);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(propose_from_partials_kernel, dim3(B), dim3(SD_WAVE), 0, (hipStream_t)stream,
                     part_val, part_idx, row_offset, row_step, tok_buf, k, step);
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
                 k <= 15, SD_ERR_INVALID, "sd_verify_from_partials: bad arguments (B=%d k=%d)", B, k);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(verify_from_partials_kernel, dim3(B), dim3(SD_WAVE * (k < 8 ? 8 : 16)), 0,
                     (hipStream_t)stream, part_val, part_idx, tok_buf, proposal_lens, out_ids,
                     reinterpret_cast<unsigned long long*>(counters), seq_lens, gen_tokens,
                     gen_stride, gen_capacity, gen_counts, step_out, k);
  SD_CHECK_LAUNCH("sd_verify_from_partials");
  return SD_OK;
}
