// A decoder layer of the DRAFT model as three launches instead of five, for passes of a few
// token rows (one or two per sequence):
//
//   1. RMSNorm + QKV + rotary + cache append        (sd_norm_qkv_rope_adds, sd_linear.hip)
//   2. attention + output projection                 (draft_attn_oproj_kernel, here)
//   3. RMSNorm + gate/up + SiLU*mul + down           (draft_mlp_kernel, here)
//
// What makes the cuts possible is that a linear layer whose INPUT is produced distributed over
// workgroups can be computed by the producers themselves: a workgroup that holds a slice of
// the input multiplies it with the matching columns of the weight matrix and contributes a
// partial output row.  Launch 2: one workgroup attends the WHOLE context of one KV head (no key
// splits across workgroups, so no merge) and multiplies the head group's normalised output
// with its columns of the projection: Hkv partial rows, kept apart in slabs and added by the
// next launch while it stages its input.  Launch 3: a workgroup owns 32 ffn columns, turns
// them into activations and multiplies those with its 32 columns of the down projection; the
// ffn/32 partial rows are added into an f32 accumulator with float atomics.
//
// The residual stream is carried as (h, pending sums): every launch that normalises computes
// h' = bf16(h + sums) itself and one workgroup keeps h' for the launches behind it.
//
// The sums run in another order than the five-launch layer's (and the atomics' order is not
// fixed), so the rows are NOT bit-identical to it: this path is for the draft model only,
// whose rounding shows in what is proposed, never in what is emitted.
#include <mutex>

#include "sd_common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / SD_WAVE;
constexpr int kXPad = 8;

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

#ifdef SD_DRAFT_STAMPS
__device__ unsigned long long sd_draft_stamps[2][16];
#define DSTAMP(k, i) \
  if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0) sd_draft_stamps[k][i] = wall_clock64()
#else
#define DSTAMP(k, i)
#endif

// ------------------------------------------------------------------ launch 2
struct AttnProjArgs {
  const bf16_t* q;
  int64_t ldq;
  const bf16_t* kc;
  const bf16_t* vc;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset;
  int q_len, Hq, Hkv, H;
  float scale_log2;
  int n_pages;        // block-table entries a workgroup stages: every page up to the context bound
  const bf16_t* Wo;   // [H/16][Hq*D/32][64][8], tiled PLAIN
  float* slabs;       // [Hkv][rows_total][H] f32: one partial projection row per KV head
  int rows_total;
  float* zero;        // words this launch clears for the launch behind it (or NULL)
  int zero_n;
};

// One workgroup = 8 waves = (KV head, 8 tiles of 16 projection rows, sequence).  Every workgroup
// of a KV head attends that head's whole context (the redundant K/V reads hit the L2: the
// workgroups of a head have equal blockIdx.x % Hkv, and Hkv is a multiple of 8 or divides it);
// 32-key tile j goes to wave j % 8, the waves are merged in index order through LDS.
// KO: k-steps of the projection per KV head (G*D/32), rounded up to 8 or 16.
template <int D, int NBUF, int KO>
__global__ __launch_bounds__(kThreads) void draft_attn_oproj_kernel(AttnProjArgs a) {
  constexpr int KS = D / 16;  // k-steps of K.Q^T
  constexpr int DT = D / 32;  // 32-row tiles of O^T
  constexpr int LDM = 33;
  constexpr int kMaxKO = KO;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm_o = reinterpret_cast<float*>(smem_raw);             // [8][D][LDM]
  float* sm_m = sm_o + kWaves * D * LDM;                         // [8][32]
  float* sm_l = sm_m + kWaves * 32;                              // [8][32]
  float* sm_w = sm_l + kWaves * 32;                              // [32][8] wave weights
  float* sm_lt = sm_w + 32 * kWaves;                             // [32] row sums
  int32_t* sm_bt = reinterpret_cast<int32_t*>(sm_lt + 32);       // [n_pages]
  const int G = a.Hq / a.Hkv;
  const int GD = G * D;                                          // channels of the head group
  bf16_t* xs = reinterpret_cast<bf16_t*>(sm_bt + ((a.n_pages + 3) & ~3));   // [q_len + 1][GD + kXPad]

  DSTAMP(0, 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int kvh = blockIdx.x % a.Hkv, part = blockIdx.x / a.Hkv, b = blockIdx.y;
  const int q_len = a.q_len;
  const int row0 = b * q_len;
  const int m_rows = q_len * G;

  // clear the next launch's accumulator (a slice per workgroup)
  if (a.zero) {
    const int n_wg = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = me * kThreads + tid; i < a.zero_n; i += n_wg * kThreads) a.zero[i] = 0.f;
  }
  // Everything whose address does not depend on the sequence length goes out first: the block
  // table, the Q^T fragments, the length.
  const int32_t* bt = a.block_tables + (size_t)b * a.max_blocks;
  const int32_t bt_reg = tid < a.n_pages ? bt[min(tid, a.max_blocks - 1)] : 0;
  bf16x8 qf[KS];
  {
    const bool valid = r < m_rows;
    const int t = valid ? r / G : 0, g = valid ? r % G : 0;
    const bf16_t* qrow = a.q + (size_t)(row0 + t) * a.ldq + (size_t)(kvh * G + g) * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) v = *reinterpret_cast<const bf16x8*>(qrow + 16 * s + 8 * h);
      qf[s] = v;
    }
  }
  const int pos0 = a.seq_lens[b] + a.pos_offset;
  const int kv_len = pos0 + q_len;
  // This wave's tile of the projection: rows 16*nt.., the head group's k-steps.  Requested now,
  // used at the very end.
  const int KSO = a.Hq * D / 32, n_ko = GD / 32;
  const int nt = part * kWaves + w;
  bf16x8 wo[kMaxKO];
  {
    const bf16_t* wp = a.Wo + ((size_t)nt * KSO + (size_t)kvh * n_ko) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < kMaxKO; ++j) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (j < n_ko) v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (size_t)j * 512));
      wo[j] = v;
    }
  }
  if (tid < a.n_pages) sm_bt[tid] = bt_reg;
  // zero row of xs: B-operand lanes of token rows >= q_len read it
  for (int c = tid; c < GD + kXPad; c += kThreads) xs[(size_t)q_len * (GD + kXPad) + c] = (bf16_t)0.f;
  const int kvlen_m = r < m_rows ? pos0 + r / G + 1 : 0;
  __syncthreads();
  DSTAMP(0, 1);

  f32x16 o_acc[DT];
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[dt][i] = 0.f;

  const size_t page_elems = (size_t)D * 16;
  const int last_page = max(kv_len - 1, 0) >> 4;
  bf16x8 kf[NBUF][KS], vf[NBUF][2][DT];
  const bf16x8* dummy = reinterpret_cast<const bf16x8*>(a.kc);
  auto page_of = [&](int tile, int second) {
    const int pg = tile * 2 + (second && tile * 2 + 1 <= last_page ? 1 : 0);
    return sm_bt[min(pg, a.n_pages - 1)];
  };
  auto load_k = [&](bf16x8(&dst)[KS], int tile) {
    const bool in = tile * 32 < kv_len;
    const int blk = in ? page_of(tile, r >> 4) : 0;
    const bf16_t* kpage = a.kc + ((size_t)blk * a.Hkv + kvh) * page_elems;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8* src = reinterpret_cast<const bf16x8*>(kpage + ((2 * s + h) * 16 + (r & 15)) * 8);
      dst[s] = *(in ? src : dummy);
    }
  };
  auto load_v = [&](bf16x8(&dst)[2][DT], int tile) {
    const bool in = tile * 32 < kv_len;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int blk = in ? page_of(tile, s) : 0;
      const bf16_t* vpage = a.vc + ((size_t)blk * a.Hkv + kvh) * page_elems;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(vpage + (dt * 32 + r) * 16 + 8 * h);
        dst[s][dt] = *(in ? src : dummy);
      }
    }
  };
#pragma unroll
  for (int u = 0; u < NBUF; ++u) {
    load_k(kf[u], w + kWaves * u);
    load_v(vf[u], w + kWaves * u);
  }
  for (int base = w; base * 32 < kv_len; base += kWaves * NBUF) {
#pragma unroll
    for (int u = 0; u < NBUF; ++u) {
      const int tile = base + kWaves * u;
      const int key0 = tile * 32;
      const bool live = key0 < kv_len;
      bf16x8 pf[2];
      float alpha = 1.f;
      if (live) {
        f32x16 s_acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) s_acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[u][s], qf[s], s_acc, 0, 0, 0);
        float sv[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          sv[i] = key < kvlen_m ? s_acc[i] * a.scale_log2 : -INFINITY;
          tmax = fmaxf(tmax, sv[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(sv[i] - m_use);
          psum += p;
          pf[i >> 3][i & 7] = (bf16_t)p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
      }
      load_k(kf[u], tile + kWaves * NBUF);
      if (live) {
        if (__any(alpha != 1.f)) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[dt][i] *= alpha;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u][s][dt], pf[s], o_acc[dt], 0, 0, 0);
      }
      load_v(vf[u], tile + kWaves * NBUF);
    }
  }

  DSTAMP(0, 2);
  // the 8 waves' (m, l, O^T) through LDS, merged in wave order
  {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (h == 0) {
      sm_m[w * 32 + r] = m_run;
      sm_l[w * 32 + r] = l_tot;
    }
    float* so = sm_o + (size_t)w * D * LDM;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = dt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        so[d * LDM + r] = o_acc[dt][i];
      }
  }
  __syncthreads();
  if (tid < 32) {
    float wgt[kWaves];
    float mx = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) {
      wgt[ww] = sm_m[ww * 32 + tid];
      mx = fmaxf(mx, wgt[ww]);
    }
    float l = 0.f;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) {
      wgt[ww] = (mx == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(wgt[ww] - mx);
      l += wgt[ww] * sm_l[ww * 32 + tid];
      sm_w[tid * kWaves + ww] = wgt[ww];
    }
    sm_lt[tid] = l;
  }
  __syncthreads();
  DSTAMP(0, 3);
  // normalised output rows of the head group, bf16, as the projection's input
  for (int it = tid; it < m_rows * D; it += kThreads) {
    const int m = it / D, d = it - m * D;
    float o = 0.f;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) o += sm_w[m * kWaves + ww] * sm_o[((size_t)ww * D + d) * LDM + m];
    const float l = sm_lt[m];
    const int t = m / G, g = m - t * G;
    xs[(size_t)t * (GD + kXPad) + g * D + d] = (bf16_t)(l > 0.f ? o / l : 0.f);
  }
  __syncthreads();
  DSTAMP(0, 4);
  // projection: this wave's 16 rows x the head group's channels
  {
    const int xc = lane & 15, xq = lane >> 4;
    const bf16_t* xrow = xs + (size_t)(xc < q_len ? xc : q_len) * (GD + kXPad) + 8 * xq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kMaxKO; ++j) {
      if (j < n_ko) {
        const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(xrow + j * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[j], bfrag, acc, 0, 0, 0);
      }
    }
    if (xc < q_len) {
      float* dst = a.slabs + ((size_t)kvh * a.rows_total + row0 + xc) * a.H + 16 * nt + 4 * xq;
      *reinterpret_cast<f32x4*>(dst) = acc;
    }
  }
  DSTAMP(0, 5);
}

// ------------------------------------------------------------------ launch 3
struct MlpArgs {
  const bf16_t* h;
  int64_t ldh;
  const float* add;     // [n_add][M][H] f32 partial rows added to h (or NULL)
  int n_add;
  bf16_t* h_out;        // [M][ldh]: h + adds, kept by workgroup 0 (or NULL)
  const bf16_t* norm_w;
  float eps;
  const bf16_t* Wup;    // [F/8][H/32][64][8], tiled GATED
  const bf16_t* Wdown;  // [F/32][H/16][64][8]: the PLAIN tiles of [H, F] ordered k-step first
  float* acc;           // [M][H] f32, added to with float atomics
  int M, H, F;
};

// A workgroup owns 32 ffn columns (4 gated tiles): gate/up rows x the normalised input, SiLU *
// mul, then its 32 columns of the down projection; K of the first matrix is cut into segments
// of 8 k-steps over the 8 waves (segment j -> wave j % 8), the output tiles of the second go
// round the waves.
template <int TPW>   // down-projection tiles per wave = H / 128
__global__ __launch_bounds__(kThreads) void draft_mlp_kernel(MlpArgs a) {
  constexpr int kSeg = 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int M = a.M, H = a.H;
  const int LDX = H + kXPad;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem_raw);                         // [M + 1][LDX]
  float* red = reinterpret_cast<float*>(smem_raw + (size_t)(M + 1) * LDX * 2);  // [8][4][256]
  bf16_t* acts = reinterpret_cast<bf16_t*>(red + kWaves * 4 * 256);          // [17][32 + 8]
  float* outs = reinterpret_cast<float*>(acts + 17 * 40);                    // [M][H]

  DSTAMP(1, 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = H / 32, n_seg = (KS + kSeg - 1) / kSeg;
  const int tile0 = blockIdx.x * 4;

  // ---- staging first (its loads must not queue behind weight requests): thread `it` takes 4
  // columns of a row: h, every partial row and the norm weight go out together
  const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = tid * 8; c < LDX; c += kThreads * 8) *reinterpret_cast<bf16x8*>(xs + (size_t)M * LDX + c) = z8;
  for (int c = tid * 8; c + 8 <= 17 * 40; c += kThreads * 8) *reinterpret_cast<bf16x8*>(acts + c) = z8;
  constexpr int kMaxAdd = 8;
  const int per_row = H / 4;
  float s2[SD_DRAFT_LAYER_MAX_ROWS] = {0.f, 0.f, 0.f, 0.f};
  // (the norm weight of the thread's first item travels with the staging loads: a load behind
  // the barrier would be one more dependent round trip)
  const bf16x4 g_first = tid < M * per_row ? *reinterpret_cast<const bf16x4*>(a.norm_w + (tid % per_row) * 4)
                                           : bf16x4{0, 0, 0, 0};
  for (int it = tid; it < M * per_row; it += kThreads) {
    const int m = it / per_row, c = (it - m * per_row) * 4;
    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(a.h + (size_t)m * a.ldh + c);
    f32x4 p[kMaxAdd];
#pragma unroll
    for (int s = 0; s < kMaxAdd; ++s) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (s < a.n_add) v = *reinterpret_cast<const f32x4*>(a.add + ((size_t)s * M + m) * H + c);
      p[s] = v;
    }
    bf16x4 o;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < kMaxAdd; ++s) sum += p[s][j];      // slabs in index order
      o[j] = (bf16_t)((float)hv[j] + sum);
      const float f = (float)o[j];
      ss += f * f;
    }
    if (m == 0) s2[0] += ss;
    if (m == 1) s2[1] += ss;
    if (m == 2) s2[2] += ss;
    if (m == 3) s2[3] += ss;
    *reinterpret_cast<bf16x4*>(xs + (size_t)m * LDX + c) = o;
    if (a.h_out && blockIdx.x == 0) *reinterpret_cast<bf16x4*>(a.h_out + (size_t)m * a.ldh + c) = o;
  }
  float* sred = reinterpret_cast<float*>(outs);     // [8 waves][4 rows], free until the end
#pragma unroll
  for (int m = 0; m < SD_DRAFT_LAYER_MAX_ROWS; ++m) {
    const float v = wave_sum(s2[m]);
    if (lane == 0) sred[wave * 4 + m] = v;
  }

  // ---- weights: this wave's segments of the 4 gated tiles, then its down tiles
  bf16x8 nxt[kSeg];
  auto load_up = [&](int tile, int seg) {
    const bf16_t* p = a.Wup + ((size_t)tile * KS + (size_t)seg * kSeg) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < kSeg; ++j) {
      const int jj = seg * kSeg + j < KS ? j : 0;
      nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p + (size_t)jj * 512));
    }
  };
  const int my_segs = wave < n_seg ? (n_seg - wave + kWaves - 1) / kWaves : 0;
  const int n_units = 4 * my_segs;                   // (tile, segment) pairs of this wave
  if (n_units > 0) load_up(tile0, wave);
  __syncthreads();                                   // raw rows and the row sums are in LDS
  DSTAMP(1, 1);
  {
    float rs[SD_DRAFT_LAYER_MAX_ROWS];
#pragma unroll
    for (int m = 0; m < SD_DRAFT_LAYER_MAX_ROWS; ++m) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < kWaves; ++ww) t += sred[ww * 4 + m];
      rs[m] = rsqrtf(t / (float)H + a.eps);
    }
    for (int it = tid; it < M * per_row; it += kThreads) {
      const int m = it / per_row, c = (it - m * per_row) * 4;
      const bf16x4 g = it == tid ? g_first : *reinterpret_cast<const bf16x4*>(a.norm_w + c);
      bf16_t* xl = xs + (size_t)m * LDX + c;
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(xl);
      const float r_m = m == 0 ? rs[0] : (m == 1 ? rs[1] : (m == 2 ? rs[2] : rs[3]));
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(round_bf16((float)v[j] * r_m) * (float)g[j]);
      *reinterpret_cast<bf16x4*>(xl) = o;
    }
  }
  __syncthreads();

  DSTAMP(1, 2);
  const int xc = lane & 15, xq = lane >> 4;
  const bf16_t* xrow = xs + (size_t)(xc < M ? xc : M) * LDX + 8 * xq;
  bf16x8 wd[TPW];
  constexpr int kEarly = TPW < 8 ? TPW : 8;
  const bf16_t* dp = a.Wdown + ((size_t)blockIdx.x * (H / 16) + (size_t)wave * TPW) * 512 + lane * 8;
  f32x4 acc[4];
  int u = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int si = 0; si < my_segs; ++si, ++u) {
      const int seg = wave + si * kWaves;
      bf16x8 cur[kSeg];
#pragma unroll
      for (int j = 0; j < kSeg; ++j) cur[j] = nxt[j];
      if (u + 1 < n_units) {
        const int t1 = (u + 1) / my_segs;
        load_up(tile0 + t1, wave + (u + 1 - t1 * my_segs) * kWaves);
      } else {
        // the last unit: the down projection's tiles follow in the stream (the first ones
        // now, the rest when the first matrix's registers are free)
#pragma unroll
        for (int i = 0; i < kEarly; ++i)
          wd[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(dp + (size_t)i * 512));
      }
      const bf16_t* xp = xrow + (size_t)seg * kSeg * 32;
#pragma unroll
      for (int j = 0; j < kSeg; ++j) {
        if (seg * kSeg + j < KS) {
          const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(xp + j * 32);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j], bfrag, acc[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i)
    if (i >= kEarly || n_units == 0)   // (a wave without a segment of the first matrix loads all here)
      wd[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(dp + (size_t)i * 512));
  DSTAMP(1, 3);
#pragma unroll
  for (int t = 0; t < 4; ++t)
    *reinterpret_cast<f32x4*>(red + ((size_t)wave * 4 + t) * 256 + lane * 4) = acc[t];
  __syncthreads();
  DSTAMP(1, 4);
  {
    // thread (t, e): element e of gated tile t; rows 0..7 of a tile are gate rows, 8..15 the
    // matching up rows (element e + 128)
    const int t = tid >> 7, e = tid & 127;
    const int nl = 4 * (e >> 6) + (e & 3), m = (e >> 2) & 15;
    if (m < M) {
      float g = 0.f, uu = 0.f;
#pragma unroll
      for (int ww = 0; ww < kWaves; ++ww) {
        g += red[((size_t)ww * 4 + t) * 256 + e];
        uu += red[((size_t)ww * 4 + t) * 256 + e + 128];
      }
      g = round_bf16(silu_f(round_bf16(g)));
      acts[m * 40 + t * 8 + nl] = (bf16_t)(g * round_bf16(uu));
    }
  }
  __syncthreads();
  DSTAMP(1, 5);
  {
    const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(acts + (xc < M ? xc : 16) * 40 + 8 * xq);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[i], bfrag, o, 0, 0, 0);
      if (xc < M) *reinterpret_cast<f32x4*>(outs + (size_t)xc * H + 16 * (wave * TPW + i) + 4 * xq) = o;
    }
  }
  __syncthreads();
  DSTAMP(1, 6);
  for (int i = tid; i < M * H; i += kThreads) atomicAdd(a.acc + i, outs[i]);
  DSTAMP(1, 7);
}

template <int D, int NBUF, int KO>
int launch_attn_oproj(const AttnProjArgs& a, int B, int parts, hipStream_t st) {
  const int GD = (a.Hq / a.Hkv) * D;
  const size_t lds = ((size_t)kWaves * D * 33 + 2 * kWaves * 32 + 32 * kWaves + 32) * 4 +
                     (size_t)((a.n_pages + 3) & ~3) * 4 + (size_t)(a.q_len + 1) * (GD + kXPad) * 2;
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(draft_attn_oproj_kernel<D, NBUF, KO>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  SD_REQUIRE(attr_err == hipSuccess, SD_ERR_HIP, "sd_draft_attn_oproj: hipFuncSetAttribute failed: %s",
             hipGetErrorString(attr_err));
  SD_REQUIRE(lds <= 160 * 1024, SD_ERR_UNSUPPORTED, "sd_draft_attn_oproj: %zu bytes of LDS", lds);
  hipLaunchKernelGGL((draft_attn_oproj_kernel<D, NBUF, KO>), dim3(a.Hkv * parts, B), dim3(kThreads), lds, st, a);
  SD_CHECK_LAUNCH("sd_draft_attn_oproj");
  return SD_OK;
}

template <int TPW>
int launch_mlp(const MlpArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(a.M + 1) * (a.H + kXPad) * 2 + (size_t)kWaves * 4 * 256 * 4 + 17 * 40 * 2 +
                     (size_t)(a.M > 1 ? a.M : 1) * a.H * 4 + 64;
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(draft_mlp_kernel<TPW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  SD_REQUIRE(attr_err == hipSuccess, SD_ERR_HIP, "sd_draft_mlp: hipFuncSetAttribute failed: %s",
             hipGetErrorString(attr_err));
  SD_REQUIRE(lds <= 160 * 1024, SD_ERR_UNSUPPORTED, "sd_draft_mlp: %zu bytes of LDS", lds);
  hipLaunchKernelGGL((draft_mlp_kernel<TPW>), dim3(a.F / 32), dim3(kThreads), lds, st, a);
  SD_CHECK_LAUNCH("sd_draft_mlp");
  return SD_OK;
}

}  // namespace

#ifdef SD_DRAFT_STAMPS
extern "C" int sd_debug_read_draft_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_draft_stamps), sizeof(sd_draft_stamps));
}
#endif

extern "C" int sd_draft_layer_supported(int rows, int q_len, int hidden, int n_heads, int n_kv_heads,
                                        int head_dim, int ffn, int max_ctx) {
  if (rows < 1 || rows > SD_DRAFT_LAYER_MAX_ROWS || q_len < 1 || q_len > 2 || rows % q_len) return 0;
  if (head_dim != 64 && head_dim != 128) return 0;
  if (n_kv_heads < 1 || n_heads % n_kv_heads) return 0;
  const int G = n_heads / n_kv_heads, GD = G * head_dim;
  if (q_len * G > 32 || GD % 32 || GD / 32 > 16) return 0;
  if (hidden % 128 || hidden > 4096 || hidden / 128 > 32 || ffn % 32) return 0;
  if (max_ctx < 1 || max_ctx > SD_DRAFT_LAYER_MAX_CTX) return 0;
  return 1;
}

extern "C" int sd_draft_attn_oproj(const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                                   const int32_t* block_tables, int max_blocks_per_seq,
                                   const int32_t* seq_lens, int pos_offset, const void* o_w,
                                   float* slabs, float* zero, int zero_n, int B, int q_len, int Hq,
                                   int Hkv, int D, int H, float scale, int max_ctx, sd_stream_t stream) {
  SD_REQUIRE(q && k_cache && v_cache && block_tables && seq_lens && o_w && slabs, SD_ERR_INVALID,
             "sd_draft_attn_oproj: null pointer");
  SD_REQUIRE(B >= 0 && zero_n >= 0 && (zero || zero_n == 0), SD_ERR_INVALID, "sd_draft_attn_oproj: bad sizes");
  SD_REQUIRE(B == 0 || sd_draft_layer_supported(B * q_len <= SD_DRAFT_LAYER_MAX_ROWS ? B * q_len : 0, q_len, H,
                                                Hq, Hkv, D, 32, max_ctx),
             SD_ERR_UNSUPPORTED,
             "sd_draft_attn_oproj: shape outside sd_draft_layer_supported (B=%d q_len=%d H=%d Hq=%d Hkv=%d "
             "D=%d max_ctx=%d)", B, q_len, H, Hq, Hkv, D, max_ctx);
  SD_REQUIRE(max_blocks_per_seq * 16 >= max_ctx, SD_ERR_INVALID,
             "sd_draft_attn_oproj: max_ctx %d exceeds the block table (%d pages)", max_ctx, max_blocks_per_seq);
  SD_REQUIRE(ldq % 8 == 0 && sd_aligned16(q) && sd_aligned16(o_w) && sd_aligned16(slabs), SD_ERR_INVALID,
             "sd_draft_attn_oproj: unaligned buffer");
  if (B == 0) return SD_OK;
  AttnProjArgs a;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.kc = (const bf16_t*)k_cache; a.vc = (const bf16_t*)v_cache;
  a.block_tables = block_tables; a.max_blocks = max_blocks_per_seq; a.seq_lens = seq_lens;
  a.pos_offset = pos_offset; a.q_len = q_len; a.Hq = Hq; a.Hkv = Hkv; a.H = H;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.n_pages = (max_ctx + 15) / 16;
  a.Wo = (const bf16_t*)o_w; a.slabs = slabs; a.rows_total = B * q_len; a.zero = zero; a.zero_n = zero_n;
  const int parts = H / 128;
  hipStream_t st = (hipStream_t)stream;
  const bool ko8 = (Hq / Hkv) * D / 32 <= 8;
  if (D == 64) return ko8 ? launch_attn_oproj<64, 2, 8>(a, B, parts, st) : launch_attn_oproj<64, 2, 16>(a, B, parts, st);
  return ko8 ? launch_attn_oproj<128, 1, 8>(a, B, parts, st) : launch_attn_oproj<128, 1, 16>(a, B, parts, st);
}

extern "C" int sd_draft_mlp(const void* h, int64_t ldh, const float* add, int n_add, void* h_out,
                            const void* norm_w, float eps, const void* up_w, const void* down_w_km,
                            float* acc, int M, int H, int F, sd_stream_t stream) {
  SD_REQUIRE(h && norm_w && up_w && down_w_km && acc, SD_ERR_INVALID, "sd_draft_mlp: null pointer");
  SD_REQUIRE(M >= 0 && M <= SD_DRAFT_LAYER_MAX_ROWS && n_add >= 0 && (add || n_add == 0), SD_ERR_INVALID,
             "sd_draft_mlp: bad sizes (M=%d n_add=%d)", M, n_add);
  SD_REQUIRE(H % 128 == 0 && H >= 128 && H <= 4096 && F % 32 == 0 && F >= 32, SD_ERR_UNSUPPORTED,
             "sd_draft_mlp: H=%d must be a multiple of 128 up to 4096 and F=%d a multiple of 32", H, F);
  SD_REQUIRE(ldh % 8 == 0 && ldh >= H && sd_aligned16(h) && sd_aligned16(add) && sd_aligned16(h_out) &&
                 sd_aligned16(up_w) && sd_aligned16(down_w_km) && sd_aligned16(acc),
             SD_ERR_INVALID, "sd_draft_mlp: unaligned buffer");
  if (M == 0) return SD_OK;
  MlpArgs a;
  a.h = (const bf16_t*)h; a.ldh = ldh; a.add = add; a.n_add = n_add; a.h_out = (bf16_t*)h_out;
  a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.Wup = (const bf16_t*)up_w;
  a.Wdown = (const bf16_t*)down_w_km; a.acc = acc; a.M = M; a.H = H; a.F = F;
  hipStream_t st = (hipStream_t)stream;
  switch (H / 128) {
    case 1: return launch_mlp<1>(a, st);
    case 2: return launch_mlp<2>(a, st);
    case 4: return launch_mlp<4>(a, st);
    case 7: return launch_mlp<7>(a, st);
    case 8: return launch_mlp<8>(a, st);
    case 12: return launch_mlp<12>(a, st);
    case 16: return launch_mlp<16>(a, st);
    case 24: return launch_mlp<24>(a, st);
    case 32: return launch_mlp<32>(a, st);
  }
  sd_set_error("sd_draft_mlp: no instantiation for H=%d", H);
  return SD_ERR_UNSUPPORTED;
}
