// Weight-streaming dense ops of a decoder layer at <= 16 token rows: the
// linear layers (one pass over the weight matrix at HBM rate, every token row
// riding along), with RMSNorm / LayerNorm fused in front and rotary embedding +
// KV-cache append fused behind; the weight re-tiling pass, stand-alone norms
// and the embedding gather.
//
// Weights are stored PRE-TILED for the 16x16x32 bf16 MFMA: tile (nt, ks) holds
// rows 16nt..16nt+15, columns 32ks..32ks+31 as 64 lanes x 8 elements in the A
// operand's lane order, so one wave load is 1 KiB of contiguous memory and is
// an MFMA operand as it lands (no LDS round trip, no cross-lane reduction, no
// VALU multiply work: the kernel is a pure HBM stream).
//
// linear_kernel: a workgroup of 8 waves walks a strided list of 16-row output
// tiles; with a fused norm it first stages the normalised activations x[M][K] in
// LDS, otherwise its lanes read their activation fragments from global memory.  K is cut into segments of 8 k-steps (256 columns); segment j
// always belongs to wave j % 8, which chains its segments in increasing order
// into one accumulator tile; the 8 wave tiles are then added in wave order
// through LDS.  That order is a function of K alone, so a token row gets the
// same bits whether it runs alone (decode) or beside k other rows (verify).
#include <mutex>

#include "sd_common.h"

namespace {

// What a workgroup stages in LDS as its activations: nothing (lanes read x from global
// memory), RMS- or LayerNorm-ed x, or the attention output merged from per-split partials.
// NORM_RMS_ADD: RMSNorm of x + add (an f32 row of pending partial sums per token row); one
// workgroup also keeps x + add for the launches behind (the draft model's three-launch layer).
enum { NORM_NONE = 0, NORM_RMS = 1, NORM_LAYER = 2, STAGE_ATTN_MERGE = 3, NORM_RMS_ADD = 4 };
constexpr int kMergeMaxSplits = 8;
constexpr int EPI_QKV_ROPE = 100;  // internal: sd_norm_qkv_rope
constexpr int EPI_ARGMAX = 101;    // internal: sd_norm_linear_argmax (slice maxima instead of logits)
constexpr int kLinThreads = 512;
constexpr int kLinWaves = kLinThreads / SD_WAVE;
constexpr int kSeg = 8;            // k-steps per segment
constexpr int kXPad = 8;           // LDS row padding (elements): conflict-free b128 reads

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
  RopePos mp;  // multimodal rotary positions (all null: the cache position)
};

// Attention partials (the workspace layout of sd_paged_attention): per (sequence, KV
// head, split) m_pad rows of D un-normalised outputs, and (row max, row sum) pairs.
struct MergeArgs {
  const float* part_o;
  const float* part_ml;
  const int32_t* seq_lens;
  int pos_offset, q_len, G, Hkv, D, n_splits, split_tokens, m_pad;
};

struct LinArgs {
  const bf16_t* W;         // tiled [n_tiles][K/32][64][8] bf16, or [n_tiles][K/64][64][16] fp8
  const float* w_scale;    // fp8 weights: f32 scale per output row (indexed like bias); else null
  const bf16_t* x;
  int64_t ldx;
  const bf16_t* bias;
  const bf16_t* residual;
  void* out;
  int64_t ldo;
  int M, N, K, KC;         // N: output columns; KC: K padded to whole 256-column segments
  int n_tiles;             // 16-row weight tiles
  const bf16_t* norm_w;    // fused norm parameters
  const bf16_t* norm_b;
  float eps;
  RopeArgs rope;
  MergeArgs merge;
  const float* add;        // NORM_RMS_ADD: [M][ld_add] f32
  int64_t ld_add;
  bf16_t* x_out;           // NORM_RMS_ADD: [M][ldx] bf16 = x + add, or NULL
  int32_t* part_idx;       // EPI_ARGMAX: [M][SD_ARGMAX_PARTS] beside out = part_val (f32)
};

// HALF: a work unit is 8 of a tile's 16 rows (lanes of the other 8 rows load nothing and
// feed zeros), so a matrix with fewer than 256 tiles still spreads over every CU.  A
// row's arithmetic is the same as in a whole-tile unit.
#ifdef SD_LIN_STAMPS
__device__ unsigned long long sd_lin_stamps[16];
#define LSTAMP(i) \
  if (blockIdx.x == 0 && threadIdx.x == 0) sd_lin_stamps[i] = wall_clock64()
#else
#define LSTAMP(i)
#endif

// W8: the weights are e4m3 bytes, two k-steps per 16-byte lane load, dequantised in registers
// (the layer is bound by the weight stream, so the conversion is free and activations keep
// bf16); the f32 scale of an output row multiplies its finished sum.
template <int EPI, int NORM, bool HALF, bool W8>
__global__ __launch_bounds__(kLinThreads) void linear_kernel(LinArgs a) {
  LSTAMP(0);
  constexpr bool GATED = EPI == SD_EPI_SILU_MUL;
  constexpr bool QKV = EPI == EPI_QKV_ROPE;
  // Without a norm the activations need no staging: each lane loads its B-operand
  // pieces straight from global memory (they sit in L2) beside the weight loads.
  constexpr bool DIRECT = NORM == NORM_NONE;
  constexpr bool ADD = NORM == NORM_RMS_ADD;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int LDX = a.KC + kXPad;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem_raw);                               // [M+1][LDX]
  float* red = reinterpret_cast<float*>(
      smem_raw + (DIRECT ? 0 : (size_t)(a.M + 1) * LDX * 2));                     // [2][waves][256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M;
  const int KS = a.K / 32;                         // k-steps in total
  const int n_seg = (KS + kSeg - 1) / kSeg;        // segments in total

  // zero row: B-operand lanes of token rows >= M read it
  if (!DIRECT)
    for (int c = tid * 8; c < LDX; c += kLinThreads * 8)
      *reinterpret_cast<bf16x8*>(xs + (size_t)M * LDX + c) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  const int xc = lane & 15, xq = lane >> 4;
  const bf16_t* xrow = xs + (size_t)(xc < M ? xc : M) * LDX + 8 * xq;           // LDS
  const bf16_t* xglob = a.x + (size_t)(xc < M ? xc : 0) * a.ldx + 8 * xq;        // DIRECT
  // the grid is even in HALF mode, so a workgroup keeps one row half for all its units
  const int row_half = HALF ? (int)(blockIdx.x & 1) : 0;
  // Lanes whose output nobody reads (HALF: the other 8 rows of the tile; DIRECT: token
  // columns >= M) ask for the address of a lane that is read, so every load of a segment is
  // unconditional: a row of A only reaches its own output row, a column of B its own
  // column, and a doubled address costs no memory traffic.
  const int wlane = (HALF && ((lane >> 3) & 1) != row_half) ? lane ^ 8 : lane;

  // A fragments (and, DIRECT, B fragments) of this wave's next segment
  constexpr int kWLoads = W8 ? kSeg / 2 : kSeg;   // 16-byte loads per segment
  constexpr int kWSteps = W8 ? 2 : 1;             // k-steps per load
  bf16x8 nxt[kWLoads];
  bf16x8 nxt_x[DIRECT ? kSeg : 1];
  auto seg_ptr = [&](int tile, int seg) {
    // 512 bf16 (1 KiB) per wave load either way
    return a.W + ((size_t)tile * (KS / kWSteps) + (size_t)seg * kWLoads) * 512 + wlane * 8;
  };
  auto load_whole_seg = [&](int tile, int seg) {  // all of the segment's k-steps exist
    const bf16_t* p = seg_ptr(tile, seg);
    const bf16_t* q = xglob + (size_t)seg * kSeg * 32;
#pragma unroll
    for (int j = 0; j < kWLoads; ++j)
      nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p + (size_t)j * 512));
    if (DIRECT) {
#pragma unroll
      for (int j = 0; j < kSeg; ++j)
        nxt_x[DIRECT ? j : 0] = *reinterpret_cast<const bf16x8*>(q + (size_t)j * 32);
    }
  };
  auto load_seg = [&](int tile, int seg) {
    if (seg * kSeg + kSeg <= KS) {  // the usual case: nothing to check
      load_whole_seg(tile, seg);
    } else {
      const bf16_t* p = seg_ptr(tile, seg);
      const bf16_t* q = xglob + (size_t)seg * kSeg * 32;
#pragma unroll
      for (int j = 0; j < kWLoads; ++j) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (seg * kSeg + j * kWSteps < KS)
          v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p + (size_t)j * 512));
        nxt[j] = v;
      }
      if (DIRECT) {
#pragma unroll
        for (int j = 0; j < kSeg; ++j) {
          bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (seg * kSeg + j < KS) v = *reinterpret_cast<const bf16x8*>(q + (size_t)j * 32);
          nxt_x[DIRECT ? j : 0] = v;
        }
      }
    }
  };
  // Fused norm, part 1 (before any weight load is issued: loads return in issue order, so
  // x asked for behind a weight segment would wait for it).  All of the staging's global
  // loads go out together: this thread's pieces of the norm weight (and bias), and the
  // wave's row in 4096-column pieces, which is summed and parked raw in LDS.  The norm
  // parameters sit in the cross-wave sum buffer, unused until the first tile ends.
  bf16_t* gs = reinterpret_cast<bf16_t*>(red);
  bf16_t* bs = gs + a.K;  // LayerNorm only
  float rs_m[2] = {0.f, 0.f}, mean_m[2] = {0.f, 0.f};  // rows wave, wave + 8
  const int n_units = HALF ? 2 * a.n_tiles : a.n_tiles;
  auto tile_of = [&](int unit) { return HALF ? unit >> 1 : unit; };
  int unit = blockIdx.x;
  // the wave's first segment: in flight while x is staged (a wave with no segment is never
  // asked for its fragments)
  const int first_tile = tile_of(min(unit, n_units - 1));
  const int first_seg = max(min(wave, n_seg - 1), 0);
  // With at most one row per wave and one 4096-column piece per row (and registers to
  // spare: not LayerNorm, no additive input), every load of the staging is asked for before
  // the wave's first weight segment and the segment before anything waits on the row: no
  // wave's weights wait for a round trip.  Otherwise the segment follows part 1.
  constexpr bool kStaged = !DIRECT && NORM != STAGE_ATTN_MERGE;
  const bool early = kStaged && !ADD && NORM != NORM_LAYER && M <= kLinWaves && a.K <= 4096 &&
                     KS % kSeg == 0 && KS > 0;
  if (NORM == STAGE_ATTN_MERGE) {
    // x[m][c] = (sum_s f_s o_s) / (sum_s f_s l_s), f_s = exp2(m_s - max m), splits in index
    // order: the arithmetic of the stand-alone merge kernel.  Every split's partial is
    // requested at once (those past the sequence's length are read and ignored).
    const MergeArgs& g = a.merge;
    for (int it = tid; it < M * (a.K / 4); it += kLinThreads) {
      const int m = it / (a.K / 4), c = (it - m * (a.K / 4)) * 4;
      const int b = m / g.q_len, t = m - b * g.q_len;
      const int head = c / g.D, d = c - head * g.D;
      const int kvh = head / g.G, mrow = t * g.G + (head - kvh * g.G);
      const size_t row0 = (size_t)(b * g.Hkv + kvh) * g.n_splits * g.m_pad + mrow;
      f32x4 ov[kMergeMaxSplits];
      float mv[kMergeMaxSplits], lv[kMergeMaxSplits];
#pragma unroll
      for (int sp = 0; sp < kMergeMaxSplits; ++sp) {
        if (sp < g.n_splits) {
          const size_t row = row0 + (size_t)sp * g.m_pad;
          ov[sp] = *reinterpret_cast<const f32x4*>(g.part_o + row * g.D + d);
          mv[sp] = g.part_ml[row * 2];
          lv[sp] = g.part_ml[row * 2 + 1];
        }
      }
      const int kv_len = g.seq_lens[b] + g.pos_offset + g.q_len;
      const int n_valid = min((kv_len + g.split_tokens - 1) / g.split_tokens, g.n_splits);
      float mx = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < kMergeMaxSplits; ++sp)
        if (sp < n_valid) mx = fmaxf(mx, mv[sp]);
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      float l = 0.f;
      if (mx != -INFINITY) {
#pragma unroll
        for (int sp = 0; sp < kMergeMaxSplits; ++sp) {
          if (sp < n_valid) {
            const float f = __builtin_amdgcn_exp2f(mv[sp] - mx);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += f * ov[sp][j];
            l += f * lv[sp];
          }
        }
      }
      bf16_t* xl = xs + (size_t)m * LDX + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) xl[j] = (bf16_t)(l > 0.f ? o[j] / l : 0.f);
    }
  } else if (!DIRECT) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 g_pre[2], b_pre[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid * 8 + i * kLinThreads * 8;
      g_pre[i] = c < a.K ? *reinterpret_cast<const bf16x8*>(a.norm_w + c) : z;
      b_pre[i] = (NORM == NORM_LAYER && c < a.K && a.norm_b)
                     ? *reinterpret_cast<const bf16x8*>(a.norm_b + c) : z;
    }
    // sums one 4096-column piece of row m into (s1, s2) and parks it raw in LDS
    auto sum_park = [&](int m, int c0, bf16x8* v, f32x4 (*p)[2], float& s1, float& s2) {
      bf16_t* xl = xs + (size_t)m * LDX;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (ADD && c < a.K) v[i][j] = (bf16_t)((float)v[i][j] + p[ADD ? i : 0][j >> 2][j & 3]);
          const float f = (float)v[i][j];
          s1 += f;
          s2 += f * f;
        }
        if (c < a.K) {
          *reinterpret_cast<bf16x8*>(xl + c) = v[i];
          if (ADD && a.x_out && blockIdx.x == 0)
            *reinterpret_cast<bf16x8*>(a.x_out + (size_t)m * a.ldx + c) = v[i];
        }
      }
    };
    // the row's scale (and mean) from its sums
    auto row_stats = [&](int rr, int m, float s1, float s2) {
      if (NORM == NORM_LAYER) {
        // variance around the mean, from the parked values (same-wave LDS round trip)
        const bf16_t* xl = xs + (size_t)m * LDX;
        const float mean = wave_sum(s1) / (float)a.K;
        float ss = 0.f;
        for (int c = lane * 8; c < a.K; c += 512) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(xl + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j] - mean;
            ss += f * f;
          }
        }
        mean_m[rr] = mean;
        rs_m[rr] = rsqrtf(wave_sum(ss) / (float)a.K + a.eps);
      } else {
        rs_m[rr] = rsqrtf(wave_sum(s2) / (float)a.K + a.eps);
      }
    };
    auto park_params = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid * 8 + i * kLinThreads * 8;
        if (c < a.K) {
          *reinterpret_cast<bf16x8*>(gs + c) = g_pre[i];
          if (NORM == NORM_LAYER) *reinterpret_cast<bf16x8*>(bs + c) = b_pre[i];
        }
      }
    };
    if (early) {
      // Straight-line, so that each wait below names exactly the loads it needs (they return
      // in issue order): every wave asks for a row (waves without one for row 0, and pieces
      // past K for the row's start: read and dropped).
      bf16x8 v0[8];
      const bf16_t* xr = a.x + (size_t)(wave < M ? wave : 0) * a.ldx;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = i * 512 + lane * 8;
        v0[i] = *reinterpret_cast<const bf16x8*>(xr + (c < a.K ? c : 0));
      }
      load_whole_seg(first_tile, first_seg);
      park_params();
      if (wave < M) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i * 512 + lane * 8 >= a.K) v0[i] = z;
        float s1 = 0.f, s2 = 0.f;
        sum_park(wave, 0, v0, nullptr, s1, s2);
        row_stats(0, wave, s1, s2);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int m = wave + rr * kLinWaves;
        if (m < M) {
          const bf16_t* xr = a.x + (size_t)m * a.ldx;
          float s1 = 0.f, s2 = 0.f;
          for (int c0 = 0; c0 < a.K; c0 += 4096) {
            bf16x8 v[8];
            f32x4 p[ADD ? 8 : 1][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int c = c0 + i * 512 + lane * 8;
              v[i] = c < a.K ? *reinterpret_cast<const bf16x8*>(xr + c) : z;
              if (ADD) {
                const float* ar = a.add + (size_t)m * a.ld_add + (c < a.K ? c : 0);
                p[ADD ? i : 0][0] = *reinterpret_cast<const f32x4*>(ar);
                p[ADD ? i : 0][1] = *reinterpret_cast<const f32x4*>(ar + 4);
              }
            }
            sum_park(m, c0, v, p, s1, s2);
          }
          row_stats(rr, m, s1, s2);
        }
      }
      park_params();
    }
  }
  if (!early) load_seg(first_tile, first_seg);
  LSTAMP(1);
  // Part 2, while the first weight segment is in flight: scale the parked rows in place.
  auto finish_norm = [&]() {
    __syncthreads();  // norm parameters are in LDS
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int m = wave + rr * kLinWaves;
      if (NORM != STAGE_ATTN_MERGE && m < M) {
        bf16_t* xl = xs + (size_t)m * LDX;
        const float rs = rs_m[rr], mean = mean_m[rr];
        for (int c = lane * 8; c < a.K; c += 512) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(xl + c);
          const bf16x8 g = *reinterpret_cast<const bf16x8*>(gs + c);
          bf16x8 bb = {0, 0, 0, 0, 0, 0, 0, 0};
          if (NORM == NORM_LAYER) bb = *reinterpret_cast<const bf16x8*>(bs + c);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (NORM == NORM_RMS || ADD)
              o[j] = (bf16_t)(round_bf16((float)v[j] * rs) * (float)g[j]);
            else
              o[j] = (bf16_t)(((float)v[j] - mean) * rs * (float)g[j] + (float)bb[j]);
          }
          *reinterpret_cast<bf16x8*>(xl + c) = o;
        }
      }
    }
    __syncthreads();
  };

  if (!DIRECT) finish_norm();
  LSTAMP(2);

  int parity = 0;
  // Epilogue role of this thread (waves 0..3): element e = tid of the summed tile is
  // output (row nl = 4*(l>>4) + reg of the tile, token m = l & 15), l = tid >> 2.
  const int nl = 4 * (tid >> 6) + (tid & 3), m = (tid >> 2) & 15;
  const bool epi_live = tid < 256 && m < M && (!HALF || (nl >> 3) == row_half);
  // Inputs of the epilogue are fetched while the weights stream, not after the sum.
  // EPI_ARGMAX: this thread's largest output so far and its column (columns come in increasing
  // order, so the first maximum wins)
  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  int q_pos = -1, q_page = 0, q_b = 0;
  if (QKV && epi_live) {
    const RopeArgs& rp = a.rope;
    q_b = m / rp.q_len;
    q_pos = rp.seq_lens[q_b] + rp.pos_offset + (m - q_b * rp.q_len);
    if (q_pos >= 0) q_page = rp.block_tables[(size_t)q_b * rp.max_blocks + (q_pos >> 4)];
  }
  for (; unit < n_units; unit += gridDim.x, parity ^= 1) {
    const int tile = tile_of(unit);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int next_unit = unit + (int)gridDim.x;
    float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f, pre3 = 0.f;  // bias / residual / cos, sin
    float sc0 = 1.f, sc1 = 1.f;                            // W8: scales of the thread's rows
    if (epi_live) {
      if (QKV) {
        if (nl < 8) {
          const RopeArgs& rp = a.rope;
          const int D = rp.D, half = D / 2, tiles_per_head = D / 16;
          const int head = tile / tiles_per_head, i = (tile % tiles_per_head) * 8 + nl;
          if (a.bias) {
            pre0 = (float)a.bias[head * D + i];
            pre1 = (float)a.bias[head * D + i + half];
          }
          if (W8) {
            sc0 = a.w_scale[head * D + i];
            sc1 = a.w_scale[head * D + i + half];
          }
          if (q_pos >= 0 && head < rp.Hq + rp.Hkv && rp.cs) {
            const int r_pos = max(sd_rope_position(rp.mp, q_pos, q_b, m, i), 0);
            const float* cs = rp.cs + (size_t)min(r_pos, rp.max_pos - 1) * D;
            pre2 = cs[i];
            pre3 = cs[half + i];
          }
        }
      } else if (GATED) {
        const int n = tile * 8 + nl;
        if (nl < 8 && n < a.N) {
          if (a.bias) {
            pre0 = (float)a.bias[n];
            pre1 = (float)a.bias[a.N + n];
          }
          if (W8) {
            sc0 = a.w_scale[n];
            sc1 = a.w_scale[a.N + n];
          }
        }
      } else {
        const int n = tile * 16 + nl;
        if (n < a.N) {
          if (a.bias) pre0 = (float)a.bias[n];
          if (EPI == SD_EPI_RESIDUAL) pre1 = (float)a.residual[(size_t)m * a.ldo + n];
          if (W8) sc0 = a.w_scale[n];
        }
      }
    }
    // this wave's segments: the j with j % 8 == wave, in increasing order
    for (int seg = wave; seg < n_seg; seg += kLinWaves) {
      bf16x8 cur[kWLoads];
#pragma unroll
      for (int j = 0; j < kWLoads; ++j) cur[j] = nxt[j];
      bf16x8 cur_x[DIRECT ? kSeg : 1];
      if (DIRECT) {
#pragma unroll
        for (int j = 0; j < kSeg; ++j) cur_x[DIRECT ? j : 0] = nxt_x[DIRECT ? j : 0];
      }
      if (seg + kLinWaves < n_seg) load_seg(tile, seg + kLinWaves);
      else if (next_unit < n_units) load_seg(tile_of(next_unit), wave);
      const bf16_t* xp = xrow + (size_t)seg * kSeg * 32;
      auto step = [&](int j) {
        const bf16x8 b = DIRECT ? cur_x[DIRECT ? j : 0]
                                : *reinterpret_cast<const bf16x8*>(xp + j * 32);
        bf16x8 w;
        if constexpr (W8) {
          const u32x4 raw = __builtin_bit_cast(u32x4, cur[j >> 1]);
          w = fp8x8_to_bf16(raw[2 * (j & 1)], raw[2 * (j & 1) + 1]);
        } else {
          w = cur[j];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc, 0, 0, 0);
      };
      if (seg * kSeg + kSeg <= KS) {
#pragma unroll
        for (int j = 0; j < kSeg; ++j) step(j);
      } else {
#pragma unroll
        for (int j = 0; j < kSeg; ++j)
          if (seg * kSeg + j < KS) step(j);
      }
    }

    LSTAMP(3);
    // add the 8 wave tiles in wave order
    float* rbuf = red + parity * kLinWaves * 256;
    *reinterpret_cast<f32x4*>(rbuf + wave * 256 + lane * 4) = acc;
    __syncthreads();
    if (epi_live) {
      auto total = [&](int e) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kLinWaves; ++w) s += rbuf[w * 256 + e];
        return s;
      };
      if (QKV) {
        // tile rows 0..7 are dims i0..i0+7 of a head, rows 8..15 their rotation partners
        // (row nl + 8 is element e + 128)
        if (nl < 8 && q_pos >= 0) {
          const RopeArgs& rp = a.rope;
          const int D = rp.D, half = D / 2;
          const int tiles_per_head = D / 16;
          const int head = tile / tiles_per_head, i = (tile % tiles_per_head) * 8 + nl;
          float lo = round_bf16(total(tid) * sc0 + pre0), hi = round_bf16(total(tid + 128) * sc1 + pre1);
          if (head < rp.Hq + rp.Hkv && rp.cs) {
            const float nlo = lo * pre2 - hi * pre3, nhi = hi * pre2 + lo * pre3;
            lo = nlo;
            hi = nhi;
          }
          if (head < rp.Hq) {
            bf16_t* q = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + (size_t)head * D;
            q[i] = (bf16_t)lo;
            q[i + half] = (bf16_t)hi;
          } else {
            const int slot = q_pos & 15;
            if (head < rp.Hq + rp.Hkv) {
              bf16_t* kp = rp.kc + ((size_t)q_page * rp.Hkv + (head - rp.Hq)) * D * 16;
              const int i2 = i + half;
              kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = (bf16_t)lo;
              kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = (bf16_t)hi;
            } else {
              bf16_t* vp = rp.vc + ((size_t)q_page * rp.Hkv + (head - rp.Hq - rp.Hkv)) * D * 16;
              const int sl = sd_v_slot(slot);
              vp[i * 16 + sl] = (bf16_t)lo;
              vp[(i + half) * 16 + sl] = (bf16_t)hi;
            }
          }
        }
      } else if (GATED) {
        // tile rows 0..7 are gate rows 8*tile.., rows 8..15 the matching up rows
        const int n = tile * 8 + nl;
        if (nl < 8 && n < a.N) {
          float g = total(tid) * sc0 + pre0, u = total(tid + 128) * sc1 + pre1;
          g = round_bf16(silu_f(round_bf16(g)));
          reinterpret_cast<bf16_t*>(a.out)[(size_t)m * a.ldo + n] = (bf16_t)(g * round_bf16(u));
        }
      } else {
        const int n = tile * 16 + nl;
        if (n < a.N) {
          const float y = total(tid) * sc0 + pre0;
          const size_t o = (size_t)m * a.ldo + n;
          if (EPI == EPI_ARGMAX) {
            if (y > best_v) {
              best_v = y;
              best_i = n;
            }
          } else if (EPI == SD_EPI_F32) {
            reinterpret_cast<float*>(a.out)[o] = y;
          } else {
            bf16_t* ob = reinterpret_cast<bf16_t*>(a.out);
            if (EPI == SD_EPI_STORE) ob[o] = (bf16_t)y;
            else if (EPI == SD_EPI_RESIDUAL) ob[o] = (bf16_t)(round_bf16(y) + pre1);
            else if (EPI == SD_EPI_RELU) ob[o] = (bf16_t)fmaxf(round_bf16(y), 0.f);
            else ob[o] = (bf16_t)gelu_tanh_f(round_bf16(y));
          }
        }
      }
    }
    LSTAMP(4);
    // No barrier here: the next tile writes the other half of `red`, and no wave
    // can get two tiles ahead of a reader because of the barrier above.
  }
  if (EPI == EPI_ARGMAX) {
    // The workgroup's maximum per token row -> slot blockIdx.x of the row's slice maxima (values
    // are the f32 logits the F32 epilogue would have stored; ties go to the lowest column).
    // A token's 16 tile rows sit in lanes 4m..4m+3 of waves 0..3.
    auto take = [&](float ov, int oi) {
      if (ov > best_v || (ov == best_v && oi < best_i)) {
        best_v = ov;
        best_i = oi;
      }
    };
    take(__shfl_xor(best_v, 1, 64), __shfl_xor(best_i, 1, 64));
    take(__shfl_xor(best_v, 2, 64), __shfl_xor(best_i, 2, 64));
    __syncthreads();  // the last tile's sums have been read
    float* s_v = red;                                      // [4][16]
    int* s_i = reinterpret_cast<int*>(red + 64);           // [4][16]
    if (tid < 256 && (tid & 3) == 0) {
      s_v[(tid >> 6) * 16 + m] = best_v;
      s_i[(tid >> 6) * 16 + m] = best_i;
    }
    __syncthreads();
    if (tid < 16 && tid < M) {
      float bv = s_v[tid];
      int bi = s_i[tid];
      for (int w = 1; w < 4; ++w) {
        const float ov = s_v[w * 16 + tid];
        const int oi = s_i[w * 16 + tid];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      float* pv = reinterpret_cast<float*>(a.out) + (size_t)tid * SD_ARGMAX_PARTS;
      int32_t* pi = a.part_idx + (size_t)tid * SD_ARGMAX_PARTS;
      pv[blockIdx.x] = bv;
      pi[blockIdx.x] = bi;
      for (int slot = blockIdx.x + gridDim.x; slot < SD_ARGMAX_PARTS; slot += gridDim.x) {
        pv[slot] = -INFINITY;     // slots no workgroup owns
        pi[slot] = 0x7fffffff;
      }
    }
  }
}

// ------------------------------------------------------------------ launch
constexpr size_t kXBudgetNorm = 144 * 1024;  // a fused norm needs whole rows

constexpr size_t kRedBytes = (size_t)2 * kLinWaves * 256 * 4;  // cross-wave sum buffer

size_t lin_lds_bytes(int M, int KC, int norm) {
  return (norm == NORM_NONE ? 0 : (size_t)(M + 1) * (KC + kXPad) * 2) + kRedBytes;  // x + zero row
}

// K padded to whole 256-column segments if the rows of a fused norm fit in LDS, else 0.
int choose_kc(int M, int K, int norm) {
  const int kpad = (K + 255) / 256 * 256;
  const size_t whole = (size_t)(M + 1) * (kpad + kXPad) * 2;
  // the norm weight (and LayerNorm bias) are staged in the sum buffer
  const size_t params =
      norm == STAGE_ATTN_MERGE ? 0 : (size_t)K * 2 * (norm == NORM_LAYER ? 2 : 1);   // (RMS_ADD: as RMS)
  return whole <= kXBudgetNorm && params <= kRedBytes ? kpad : 0;
}

template <int EPI, int NORM, bool HALF, bool W8>
int launch_w(LinArgs& a, hipStream_t st) {
  const size_t lds = lin_lds_bytes(a.M, a.KC, NORM);
  // once per instantiation and process (host threads may race here)
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_kernel<EPI, NORM, HALF, W8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess) {
    sd_set_error("sd_linear: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    return SD_ERR_HIP;
  }
  // Up to two workgroups per CU (LDS permitting), each walking a strided tile list.
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;  // 16 waves per CU
  const int n_units = HALF ? 2 * a.n_tiles : a.n_tiles;
  const int blocks = n_units < 256 * per_cu ? n_units : 256 * per_cu;
  hipLaunchKernelGGL((linear_kernel<EPI, NORM, HALF, W8>), dim3(blocks), dim3(kLinThreads), lds,
                     st, a);
  SD_CHECK_LAUNCH("sd_linear");
  return SD_OK;
}

template <int EPI, int NORM, bool HALF>
int launch_half(LinArgs& a, hipStream_t st) {
  return a.w_scale ? launch_w<EPI, NORM, HALF, true>(a, st) : launch_w<EPI, NORM, HALF, false>(a, st);
}

template <int EPI, int NORM>
int launch_one(LinArgs& a, hipStream_t st) {
  // Row-half units where whole tiles would leave CUs idle (plain row-per-output tiles only:
  // gated and rotary tiles pair row r with row r + 8).
  constexpr bool kCanHalve = (NORM == NORM_NONE || NORM == STAGE_ATTN_MERGE) &&
                             EPI != SD_EPI_SILU_MUL && EPI != EPI_QKV_ROPE;
  if constexpr (kCanHalve) {
    if (a.n_tiles < 256) return launch_half<EPI, NORM, true>(a, st);
  }
  return launch_half<EPI, NORM, false>(a, st);
}

template <int NORM>
int launch_n(LinArgs& a, int epi, hipStream_t st) {
  if constexpr (NORM == NORM_RMS_ADD) {
    if (epi == SD_EPI_F32) return launch_one<SD_EPI_F32, NORM>(a, st);
    if (epi == EPI_QKV_ROPE) return launch_one<EPI_QKV_ROPE, NORM>(a, st);
    sd_set_error("sd_linear: epilogue %d not available with an additive input", epi);
    return SD_ERR_UNSUPPORTED;
  }
  if constexpr (NORM == STAGE_ATTN_MERGE) {
    if (epi == SD_EPI_STORE) return launch_one<SD_EPI_STORE, NORM>(a, st);
    if (epi == SD_EPI_RESIDUAL) return launch_one<SD_EPI_RESIDUAL, NORM>(a, st);
    sd_set_error("sd_linear_attn_merge: epilogue %d not available", epi);
    return SD_ERR_UNSUPPORTED;
  }
  if (epi == EPI_ARGMAX) {
    if constexpr (NORM == NORM_RMS || NORM == NORM_LAYER) return launch_one<EPI_ARGMAX, NORM>(a, st);
    sd_set_error("sd_norm_linear_argmax: needs a fused norm");
    return SD_ERR_UNSUPPORTED;
  }
  switch (epi) {
    case SD_EPI_STORE: return launch_one<SD_EPI_STORE, NORM>(a, st);
    case SD_EPI_SILU_MUL: return launch_one<SD_EPI_SILU_MUL, NORM>(a, st);
    case SD_EPI_RELU: return launch_one<SD_EPI_RELU, NORM>(a, st);
    case SD_EPI_F32: return launch_one<SD_EPI_F32, NORM>(a, st);
    case EPI_QKV_ROPE: return launch_one<EPI_QKV_ROPE, NORM>(a, st);
    case SD_EPI_RESIDUAL:
      if (NORM == NORM_NONE) return launch_one<SD_EPI_RESIDUAL, NORM_NONE>(a, st);
      break;
    case SD_EPI_GELU_TANH:
      if (NORM == NORM_NONE) return launch_one<SD_EPI_GELU_TANH, NORM_NONE>(a, st);
      break;
  }
  sd_set_error("sd_linear: epilogue %d not available with norm mode %d", epi, NORM);
  return SD_ERR_UNSUPPORTED;
}

}  // namespace

// Mirrors launch_one / launch_w: which units the launch's workgroups ask for first.
sd_pf_range sd_linear_pf_range(const void* W, int N, int K, int epi, int fp8, int fused_norm,
                               size_t max_bytes) {
  const bool paired = epi == SD_EPI_SILU_MUL || epi == EPI_QKV_ROPE;
  const int n_tiles = paired ? (N + 7) / 8 : (N + 15) / 16;
  const bool half = !paired && !fused_norm && n_tiles < 256;  // as launch_one decides
  const unsigned tile_bytes = (unsigned)(16 * (size_t)K * (fp8 ? 1 : 2));
  const unsigned n_units = half ? 2 * n_tiles : n_tiles;
  const unsigned unit_bytes = half ? tile_bytes / 2 : tile_bytes;
  sd_pf_range r;
  r.base = W;
  r.tile_bytes = tile_bytes;
  r.half = half;
  r.consumer_grid = n_units < 512 ? n_units : 512;
  const size_t cap = max_bytes / unit_bytes;
  r.n_units = (unsigned)(cap < n_units ? cap : n_units);
  return r;
}

namespace {
int linear_common(LinArgs& a, int epi, int norm, hipStream_t st) {
  SD_REQUIRE(a.W && a.x && a.out, SD_ERR_INVALID, "sd_linear: null pointer");
  SD_REQUIRE(a.M >= 0 && a.M <= 16, SD_ERR_UNSUPPORTED, "sd_linear: M=%d outside 0..16", a.M);
  SD_REQUIRE(a.N >= 1 && a.K >= 32 && a.K % 32 == 0, SD_ERR_INVALID,
             "sd_linear: need N>=1 and K a positive multiple of 32 (N=%d K=%d)", a.N, a.K);
  SD_REQUIRE(a.ldx % 8 == 0 && a.ldx >= a.K && sd_aligned16(a.W) && sd_aligned16(a.x),
             SD_ERR_INVALID, "sd_linear: rows must be 16-byte aligned (ldx=%lld)", (long long)a.ldx);
  SD_REQUIRE(epi != SD_EPI_RESIDUAL || a.residual, SD_ERR_INVALID,
             "sd_linear: residual epilogue without residual pointer");
  SD_REQUIRE(norm == NORM_NONE || norm == STAGE_ATTN_MERGE || a.norm_w, SD_ERR_INVALID,   // (RMS_ADD too)
             "sd_linear: fused norm without weight");
  SD_REQUIRE(!a.w_scale || a.K % 64 == 0, SD_ERR_UNSUPPORTED,
             "sd_linear: fp8 weights need K %% 64 == 0 (K=%d)", a.K);
  if (a.M == 0) return SD_OK;
  a.KC = norm == NORM_NONE ? (a.K + 255) / 256 * 256 : choose_kc(a.M, a.K, norm);
  SD_REQUIRE(a.KC > 0, SD_ERR_UNSUPPORTED,
             "sd_linear: fused norm needs (M+1)*K*2 <= %zu bytes of LDS and K <= 8192 (LayerNorm: 4096) (M=%d K=%d)", kXBudgetNorm,
             a.M, a.K);
  switch (norm) {
    case NORM_NONE: return launch_n<NORM_NONE>(a, epi, st);
    case NORM_RMS: return launch_n<NORM_RMS>(a, epi, st);
    case NORM_LAYER: return launch_n<NORM_LAYER>(a, epi, st);
    case STAGE_ATTN_MERGE: return launch_n<STAGE_ATTN_MERGE>(a, epi, st);
    case NORM_RMS_ADD: return launch_n<NORM_RMS_ADD>(a, epi, st);
  }
  sd_set_error("sd_linear: unknown norm mode %d", norm);
  return SD_ERR_INVALID;
}

// dst tile (nt, ks), lane l, element j  <-  src[row(nt, l & 15)][32 ks + 8 (l >> 4) + j]
__global__ __launch_bounds__(256) void tile_weights_kernel(const bf16_t* __restrict__ src,
                                                           bf16_t* __restrict__ dst, int rows_src,
                                                           int K, int n_tiles, int mode, int p0) {
  const int KS = K / 32;
  const size_t total = (size_t)n_tiles * KS * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const size_t t = i >> 6;
    const int ks = (int)(t % KS), nt = (int)(t / KS);
    const int r = l & 15, q = l >> 4;
    int row;
    if (mode == SD_TILE_PLAIN) {
      row = nt * 16 + r;
    } else if (mode == SD_TILE_GATED) {  // p0 = F: gate rows 8nt.., then up rows F + 8nt..
      const int n = nt * 8 + (r & 7);
      row = n < p0 ? (r < 8 ? n : p0 + n) : rows_src;  // past the end: zeros
    } else {  // SD_TILE_ROTARY, p0 = D: dims i0..i0+7 and their partners i0+D/2..
      const int tph = p0 / 16, head = nt / tph, i0 = (nt % tph) * 8;
      row = head * p0 + ((r < 8) ? i0 + r : i0 + p0 / 2 + (r - 8));
    }
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < rows_src) v = *reinterpret_cast<const bf16x8*>(src + (size_t)row * K + ks * 32 + q * 8);
    *reinterpret_cast<bf16x8*>(dst + i * 8) = v;
  }
}

// fp8 form: dst (nt, ks2), lane l, byte 8h + j  <-  src[row(nt, l & 15)][64 ks2 + 32 h + 8 (l >> 4) + j]
__global__ __launch_bounds__(256) void tile_weights_fp8_kernel(const uint8_t* __restrict__ src,
                                                               uint8_t* __restrict__ dst, int rows_src,
                                                               int K, int n_tiles, int mode, int p0) {
  const int KS2 = K / 64;
  const size_t total = (size_t)n_tiles * KS2 * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const size_t t = i >> 6;
    const int ks2 = (int)(t % KS2), nt = (int)(t / KS2);
    const int r = l & 15, q = l >> 4;
    int row;
    if (mode == SD_TILE_PLAIN) {
      row = nt * 16 + r;
    } else if (mode == SD_TILE_GATED) {
      const int n = nt * 8 + (r & 7);
      row = n < p0 ? (r < 8 ? n : p0 + n) : rows_src;
    } else {
      const int tph = p0 / 16, head = nt / tph, i0 = (nt % tph) * 8;
      row = head * p0 + ((r < 8) ? i0 + r : i0 + p0 / 2 + (r - 8));
    }
    uint2 lo = {0u, 0u}, hi = {0u, 0u};
    if (row < rows_src) {
      const uint8_t* sp = src + (size_t)row * K + ks2 * 64 + q * 8;
      lo = *reinterpret_cast<const uint2*>(sp);
      hi = *reinterpret_cast<const uint2*>(sp + 32);
    }
    *reinterpret_cast<uint4*>(dst + i * 16) = uint4{lo.x, lo.y, hi.x, hi.y};
  }
}

// Stand-alone norms: one wave per row (four rows per workgroup), with exactly the arithmetic of
// the norm fused into linear_kernel (each lane sums its 8-element pieces 512 columns apart, then
// the wave butterfly), so a layer gives a row the same bits whether its norm was fused or not:
// which of the two a pass takes depends on how many rows it carries.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                      const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int64_t ldo,
                                                      int rows, int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx;
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  float s2 = 0.f;
  for (int c0 = 0; c0 < H; c0 += 4096) {  // eight pieces in flight, summed in column order
    bf16x8 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 512 + lane * 8;
      v[i] = c < H ? *reinterpret_cast<const bf16x8*>(xr + c) : z;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[i][j];
        s2 += f * f;
      }
  }
  const float rs = rsqrtf(wave_sum(s2) / (float)H + eps);
  bf16_t* orow = out + (size_t)row * ldo;
  for (int c0 = 0; c0 < H; c0 += 4096) {
    bf16x8 v[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 512 + lane * 8;
      v[i] = c < H ? *reinterpret_cast<const bf16x8*>(xr + c) : z;
      g[i] = c < H ? *reinterpret_cast<const bf16x8*>(w + c) : z;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 512 + lane * 8;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(round_bf16((float)v[i][j] * rs) * (float)g[i][j]);
      if (c < H) *reinterpret_cast<bf16x8*>(orow + c) = o;
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                        const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bparam,
                                                        bf16_t* __restrict__ out, int64_t ldo,
                                                        int rows, int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx;
  float s1 = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s1 += (float)v[j];
  }
  const float mean = wave_sum(s1) / (float)H;
  float ss = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j] - mean;
      ss += f * f;
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)H + eps);
  bf16_t* orow = out + (size_t)row * ldo;
  for (int c = lane * 8; c < H; c += 512) {
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

// table_tiled != 0: the table is in the tiled layout of sd_tile_weights(PLAIN).
__global__ __launch_bounds__(256) void embed_kernel(const int32_t* __restrict__ ids,
                                                    int64_t id_stride_seq, int64_t id_offset,
                                                    const bf16_t* __restrict__ table, int table_tiled,
                                                    const bf16_t* __restrict__ pos_table,
                                                    int pos_bias, const int32_t* __restrict__ seq_lens,
                                                    int pos_offset, bf16_t* __restrict__ out,
                                                    int q_len, int H, int vocab) {
  const int row = blockIdx.x;
  const int b = row / q_len, t = row % q_len;
  int id = ids[(size_t)b * id_stride_seq + id_offset + t];
  id = min(max(id, 0), vocab - 1);
  const bf16_t* psrc = nullptr;
  if (pos_table) {
    const int pos = max(seq_lens[b] + pos_offset + t, 0);
    psrc = pos_table + (size_t)(pos + pos_bias) * H;
  }
  bf16_t* dst = out + (size_t)row * H;
  const int KS = H / 32;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    const bf16_t* src = table_tiled
        ? table + (((size_t)(id >> 4) * KS + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (id & 15)) * 8
        : table + (size_t)id * H + c;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
    if (psrc) {
      const bf16x8 p = *reinterpret_cast<const bf16x8*>(psrc + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)p[j]);
    }
    *reinterpret_cast<bf16x8*>(dst + c) = v;
  }
}

int tiles_for(int N, int epi) { return epi == SD_EPI_SILU_MUL ? (N + 7) / 8 : (N + 15) / 16; }

}  // namespace

bool sd_fused_norm_fits(int M, int K, bool layer_norm) {
  return choose_kc(M, K, layer_norm ? NORM_LAYER : NORM_RMS) > 0;
}

#ifdef SD_LIN_STAMPS
extern "C" int sd_debug_read_lin_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_lin_stamps), sizeof(sd_lin_stamps));
}
#endif

extern "C" size_t sd_tiled_weight_elems(int N, int K, int mode) {
  if (N < 1 || K < 32 || K % 32) return 0;
  const int tiles = mode == SD_TILE_GATED ? (N + 7) / 8 : (N + 15) / 16;
  return (size_t)tiles * 16 * K;
}

namespace {
int tile_weights_common(const void* src, void* dst, int N, int K, int mode, int param, bool fp8,
                        hipStream_t stream) {
  SD_REQUIRE(src && dst && N >= 1 && K >= 32 && K % 32 == 0, SD_ERR_INVALID,
             "sd_tile_weights: bad arguments (N=%d K=%d)", N, K);
  SD_REQUIRE(!fp8 || K % 64 == 0, SD_ERR_UNSUPPORTED,
             "sd_tile_weights_fp8: K %% 64 != 0 (K=%d)", K);
  SD_REQUIRE(sd_aligned16(src) && sd_aligned16(dst), SD_ERR_INVALID, "sd_tile_weights: unaligned");
  int tiles, rows_src, p0 = 0;
  if (mode == SD_TILE_PLAIN) {
    tiles = (N + 15) / 16;
    rows_src = N;
  } else if (mode == SD_TILE_GATED) {  // N = F outputs; src has 2F rows (gate, then up)
    tiles = (N + 7) / 8;
    rows_src = 2 * N;
    p0 = N;
  } else if (mode == SD_TILE_ROTARY) {  // N rows = heads * D; param = D
    SD_REQUIRE(param >= 16 && param % 16 == 0 && N % param == 0, SD_ERR_INVALID,
               "sd_tile_weights: rotary mode needs head_dim %% 16 == 0 and N %% head_dim == 0");
    tiles = N / 16;
    rows_src = N;
    p0 = param;
  } else {
    sd_set_error("sd_tile_weights: unknown mode %d", mode);
    return SD_ERR_INVALID;
  }
  if (fp8)
    hipLaunchKernelGGL(tile_weights_fp8_kernel, dim3(2048), dim3(256), 0, stream,
                       (const uint8_t*)src, (uint8_t*)dst, rows_src, K, tiles, mode, p0);
  else
    hipLaunchKernelGGL(tile_weights_kernel, dim3(2048), dim3(256), 0, stream,
                       (const bf16_t*)src, (bf16_t*)dst, rows_src, K, tiles, mode, p0);
  SD_CHECK_LAUNCH("sd_tile_weights");
  return SD_OK;
}
}  // namespace

extern "C" int sd_tile_weights(const void* src, void* dst, int N, int K, int mode, int param,
                               sd_stream_t stream) {
  return tile_weights_common(src, dst, N, K, mode, param, false, (hipStream_t)stream);
}

extern "C" int sd_tile_weights_fp8(const void* src, void* dst, int N, int K, int mode, int param,
                                   sd_stream_t stream) {
  return tile_weights_common(src, dst, N, K, mode, param, true, (hipStream_t)stream);
}

extern "C" int sd_linear(const void* W, const float* w_scale, const void* x, int64_t ldx,
                         const void* bias, const void* residual, void* out, int64_t ldo, int M,
                         int N, int K, int epilogue, sd_stream_t stream) {
  SD_REQUIRE(epilogue >= SD_EPI_STORE && epilogue <= SD_EPI_GELU_TANH, SD_ERR_INVALID,
             "sd_linear: unknown epilogue %d", epilogue);
  SD_REQUIRE(ldo >= N, SD_ERR_INVALID, "sd_linear: ldo < N");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.residual = (const bf16_t*)residual; a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K;
  a.w_scale = w_scale;
  a.n_tiles = tiles_for(N, epilogue);
  return linear_common(a, epilogue, NORM_NONE, (hipStream_t)stream);
}

extern "C" int sd_linear_attn_merge(const void* W, const float* w_scale, const void* workspace,
                                    size_t workspace_bytes,
                                    const int32_t* seq_lens, int pos_offset, int B, int q_len,
                                    int Hq, int Hkv, int D, int max_ctx, int split_tokens,
                                    const void* bias, const void* residual, void* out,
                                    int64_t ldo, int N, int epilogue, sd_stream_t stream) {
  SD_REQUIRE(W && workspace && seq_lens && out, SD_ERR_INVALID,
             "sd_linear_attn_merge: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && (D == 64 || D == 128),
             SD_ERR_INVALID, "sd_linear_attn_merge: bad shape");
  SD_REQUIRE(split_tokens >= 128 && max_ctx >= 1, SD_ERR_INVALID,
             "sd_linear_attn_merge: bad split_tokens / max_ctx");
  SD_REQUIRE(ldo >= N, SD_ERR_INVALID, "sd_linear_attn_merge: ldo < N");
  const int G = Hq / Hkv;
  const int n_splits = (max_ctx + split_tokens - 1) / split_tokens;
  SD_REQUIRE(q_len * G <= 64 && n_splits >= 1 && n_splits <= kMergeMaxSplits, SD_ERR_UNSUPPORTED,
             "sd_linear_attn_merge: %d splits (at most %d)", n_splits, kMergeMaxSplits);
  const int m_pad = q_len * G <= 32 ? 32 : 64;
  const size_t rows = (size_t)B * Hkv * n_splits * m_pad;
  SD_REQUIRE(workspace_bytes >= rows * (D + 2) * sizeof(float), SD_ERR_WORKSPACE,
             "sd_linear_attn_merge: workspace smaller than the partials");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)workspace; a.ldx = (int64_t)Hq * D;
  a.bias = (const bf16_t*)bias; a.residual = (const bf16_t*)residual; a.out = out; a.ldo = ldo;
  a.M = B * q_len; a.N = N; a.K = Hq * D;
  a.w_scale = w_scale;
  a.n_tiles = tiles_for(N, epilogue);
  a.merge = MergeArgs{(const float*)workspace, (const float*)workspace + rows * D, seq_lens,
                      pos_offset, q_len, G, Hkv, D, n_splits, split_tokens, m_pad};
  return linear_common(a, epilogue, STAGE_ATTN_MERGE, (hipStream_t)stream);
}

namespace {
// adds != NULL with an add pointer: the staging normalises x + add and keeps it in adds->x_out.
int apply_adds(LinArgs& a, const sd_row_adds* adds, int* norm, const char* who) {
  if (!adds || !adds->add) return SD_OK;
  SD_REQUIRE(*norm == NORM_RMS, SD_ERR_UNSUPPORTED, "%s: an additive input needs a fused RMSNorm", who);
  SD_REQUIRE(!a.w_scale, SD_ERR_UNSUPPORTED, "%s: an additive input needs bf16 weights", who);
  SD_REQUIRE(adds->ld_add >= a.K && adds->ld_add % 4 == 0 && sd_aligned16(adds->add) &&
                 sd_aligned16(adds->x_out) && a.K % 8 == 0,
             SD_ERR_INVALID, "%s: additive rows must be 16-byte aligned (ld_add=%lld)", who,
             (long long)adds->ld_add);
  a.add = adds->add;
  a.ld_add = adds->ld_add;
  a.x_out = (bf16_t*)adds->x_out;
  *norm = NORM_RMS_ADD;
  return SD_OK;
}
}  // namespace

extern "C" int sd_norm_linear(const void* W, const float* w_scale, const void* x, int64_t ldx,
                              const void* norm_w,
                              const void* norm_b, float eps, int layer_norm, const void* bias,
                              void* out, int64_t ldo, int M, int N, int K, int epilogue,
                              sd_stream_t stream) {
  return sd_norm_linear_adds(W, w_scale, x, ldx, norm_w, norm_b, eps, layer_norm, bias, out, ldo, M, N,
                             K, epilogue, nullptr, stream);
}

extern "C" int sd_norm_linear_adds(const void* W, const float* w_scale, const void* x, int64_t ldx,
                                   const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                   const void* bias, void* out, int64_t ldo, int M, int N, int K,
                                   int epilogue, const sd_row_adds* adds, sd_stream_t stream) {
  SD_REQUIRE(epilogue == SD_EPI_STORE || epilogue == SD_EPI_SILU_MUL || epilogue == SD_EPI_RELU ||
                 epilogue == SD_EPI_F32, SD_ERR_UNSUPPORTED,
             "sd_norm_linear: epilogue %d not supported with a fused norm", epilogue);
  SD_REQUIRE(ldo >= N, SD_ERR_INVALID, "sd_norm_linear: ldo < N");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K;
  a.w_scale = w_scale;
  a.n_tiles = tiles_for(N, epilogue);
  a.norm_w = (const bf16_t*)norm_w; a.norm_b = (const bf16_t*)norm_b; a.eps = eps;
  int norm = layer_norm ? NORM_LAYER : NORM_RMS;
  const int rc = apply_adds(a, adds, &norm, "sd_norm_linear_adds");
  if (rc != SD_OK) return rc;
  return linear_common(a, epilogue, norm, (hipStream_t)stream);
}

extern "C" int sd_norm_linear_argmax(const void* W, const float* w_scale, const void* x, int64_t ldx,
                                     const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                     float* part_val, int32_t* part_idx, int M, int N, int K,
                                     sd_stream_t stream) {
  SD_REQUIRE(part_val && part_idx, SD_ERR_INVALID, "sd_norm_linear_argmax: null pointer");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx;
  a.out = part_val; a.part_idx = part_idx; a.ldo = N; a.M = M; a.N = N; a.K = K;
  a.w_scale = w_scale;
  a.n_tiles = tiles_for(N, EPI_ARGMAX);
  a.norm_w = (const bf16_t*)norm_w; a.norm_b = (const bf16_t*)norm_b; a.eps = eps;
  return linear_common(a, EPI_ARGMAX, layer_norm ? NORM_LAYER : NORM_RMS, (hipStream_t)stream);
}

extern "C" int sd_norm_qkv_rope(const void* W, const float* w_scale, const void* x, int64_t ldx,
                                const void* norm_w,
                                const void* norm_b, float eps, int layer_norm, const void* bias,
                                void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                                void* k_cache, void* v_cache, const int32_t* block_tables,
                                int max_blocks_per_seq, const int32_t* seq_lens, int pos_offset,
                                int B, int q_len, int Hq, int Hkv, int D, int K,
                                const sd_rope_positions* rope_pos, sd_stream_t stream) {
  return sd_norm_qkv_rope_adds(W, w_scale, x, ldx, norm_w, norm_b, eps, layer_norm, bias, q_out, ldq,
                               rope_cs, max_pos, k_cache, v_cache, block_tables, max_blocks_per_seq,
                               seq_lens, pos_offset, B, q_len, Hq, Hkv, D, K, rope_pos, nullptr, stream);
}

extern "C" int sd_norm_qkv_rope_adds(const void* W, const float* w_scale, const void* x, int64_t ldx,
                                     const void* norm_w, const void* norm_b, float eps, int layer_norm,
                                     const void* bias, void* q_out, int64_t ldq, const float* rope_cs,
                                     int max_pos, void* k_cache, void* v_cache,
                                     const int32_t* block_tables, int max_blocks_per_seq,
                                     const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                                     int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                                     const sd_row_adds* adds, sd_stream_t stream) {
  SD_REQUIRE(q_out && k_cache && v_cache && block_tables && seq_lens, SD_ERR_INVALID,
             "sd_norm_qkv_rope: null pointer");
  SD_REQUIRE(!rope_pos || (rope_pos->section_t >= 0 && rope_pos->section_h >= 0 &&
                           rope_pos->section_t + rope_pos->section_h <= D / 2),
             SD_ERR_INVALID, "sd_norm_qkv_rope: rotary sections exceed D/2");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && D % 16 == 0 && D >= 16,
             SD_ERR_INVALID, "sd_norm_qkv_rope: bad shape (D=%d)", D);
  SD_REQUIRE(!rope_cs || max_pos >= 1, SD_ERR_INVALID, "sd_norm_qkv_rope: max_pos < 1");
  SD_REQUIRE(ldq >= (int64_t)Hq * D, SD_ERR_INVALID, "sd_norm_qkv_rope: ldq too small");
  LinArgs a{};
  a.W = (const bf16_t*)W; a.x = (const bf16_t*)x; a.ldx = ldx; a.bias = (const bf16_t*)bias;
  a.out = q_out; a.ldo = ldq; a.M = B * q_len; a.N = (Hq + 2 * Hkv) * D; a.K = K;
  a.w_scale = w_scale;
  a.n_tiles = a.N / 16;
  a.norm_w = (const bf16_t*)norm_w; a.norm_b = (const bf16_t*)norm_b; a.eps = eps;
  a.rope = RopeArgs{rope_cs, max_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, block_tables,
                    max_blocks_per_seq, seq_lens, pos_offset, q_len, Hq, Hkv, D,
                    sd_make_rope_pos(rope_pos, B * q_len)};
  int norm = norm_w ? (layer_norm ? NORM_LAYER : NORM_RMS) : NORM_NONE;
  const int rc = apply_adds(a, adds, &norm, "sd_norm_qkv_rope_adds");
  if (rc != SD_OK) return rc;
  return linear_common(a, EPI_QKV_ROPE, norm, (hipStream_t)stream);
}

extern "C" int sd_rmsnorm(const void* x, int64_t ldx, const void* w, void* out, int64_t ldo,
                          int rows, int H, float eps, sd_stream_t stream) {
  SD_REQUIRE(x && w && out && rows >= 0 && H >= 8 && H % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0,
             SD_ERR_INVALID, "sd_rmsnorm: bad arguments (rows=%d H=%d)", rows, H);
  if (rows == 0) return SD_OK;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, rows, H, eps);
  SD_CHECK_LAUNCH("sd_rmsnorm");
  return SD_OK;
}

extern "C" int sd_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* out,
                            int64_t ldo, int rows, int H, float eps, sd_stream_t stream) {
  SD_REQUIRE(x && w && out && rows >= 0 && H >= 8 && H % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0,
             SD_ERR_INVALID, "sd_layernorm: bad arguments (rows=%d H=%d)", rows, H);
  if (rows == 0) return SD_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out,
                     ldo, rows, H, eps);
  SD_CHECK_LAUNCH("sd_layernorm");
  return SD_OK;
}

extern "C" int sd_embed(const int32_t* ids, int64_t id_stride_seq, int64_t id_offset,
                        const void* table, int table_tiled, const void* pos_table, int pos_bias,
                        const int32_t* seq_lens, int pos_offset, void* out, int B, int q_len,
                        int H, int vocab, sd_stream_t stream) {
  SD_REQUIRE(ids && table && out && B >= 0 && q_len >= 1 && H % 32 == 0 && vocab >= 1,
             SD_ERR_INVALID, "sd_embed: bad arguments");
  SD_REQUIRE(!pos_table || seq_lens, SD_ERR_INVALID, "sd_embed: pos_table needs seq_lens");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(embed_kernel, dim3(B * q_len), dim3(256), 0, (hipStream_t)stream, ids,
                     id_stride_seq, id_offset, (const bf16_t*)table, table_tiled,
                     (const bf16_t*)pos_table, pos_bias, seq_lens, pos_offset, (bf16_t*)out, q_len,
                     H, vocab);
  SD_CHECK_LAUNCH("sd_embed");
  return SD_OK;
}
