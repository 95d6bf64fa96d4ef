// Two dependent weight-streaming linear layers of a decoder layer in ONE launch, for passes of
// at most 4 token rows (a draft model's decode passes), where a layer is a chain of launches
// whose boundaries cost as much as the bytes they stream:
//
//   o_up      output projection (+ residual, attention split merge)  ->  RMSNorm + gate/up + SiLU*mul
//   down_qkv  down projection (+ residual)  ->  the NEXT layer's RMSNorm + QKV + rotary + cache append
//
// Inside a launch the second matrix depends on every output of the first.  The first phase's
// outputs travel as 8-byte {bf16 value, tag} words (one write-through store each; a reader
// polls the words it needs until they carry this launch's tag: no barrier, no flag, no fence),
// and the weight stream never stops: a workgroup's 8 streaming waves walk ONE list of weight
// segments over both phases through a ring of three segments in registers, with a bounded
// number of requests outstanding, so the second matrix is taken in while the first phase's
// outputs cross the chip.  Extra waves (one per token row) do all activation movement: they
// gather the tagged words, apply the norm and park the rows in LDS, so no poll ever queues
// behind a weight load of its own wave.  Measured on MI355X the pair of launches is slower than
// the four launches it replaces (36.4 against 33.0 us per Llama-3.2-1B layer): 256 CUs
// gathering the same 16 KB of coherent words takes 5..7 us under a full-rate stream, more than a
// launch boundary costs.  Callers opt in (sd_forward_args.chain_state).
//
// Arithmetic is that of linear_kernel (sd_linear.hip) operation for operation: K segments of
// 256 columns, segment j on wave j % 8 in increasing order, wave tiles added in wave order, the
// same norm sums, roundings and epilogues.  A row therefore has the same bits as through
// sd_linear_attn_merge / sd_linear + sd_norm_linear / sd_norm_qkv_rope.
//
// Every wait is bounded; a launch whose peers do not arrive sets the state's failure word and
// finishes with unspecified values (sd_chain_errors).  The grid is at most one workgroup per
// CU and every workgroup is resident at once: no workgroup waits for one that cannot start.
#include <mutex>

#include "sd_common.h"

namespace {

constexpr int kStreamWaves = 8;
constexpr int kStreamThreads = kStreamWaves * SD_WAVE;
constexpr int kMaxRows = SD_CHAIN_MAX_ROWS;
constexpr int kMaxThreads = kStreamThreads + kMaxRows * SD_WAVE;
constexpr int kSeg = 8;            // k-steps per segment
constexpr int kXPad = 8;           // LDS row padding (elements)
constexpr int kMergeMaxSplits = 8;
constexpr size_t kRedBytes = (size_t)2 * kStreamWaves * 256 * 4;
constexpr int kMaxUnits = 8;       // units of one phase a workgroup may own
constexpr size_t kEpiBytes = ((size_t)2 * kMaxRows * kMaxUnits * 16 + 4 * kMaxUnits * 16) * 4 +
                             kMaxRows * 2 * 4;
constexpr unsigned long long kPollTicks = 20000000ull;  // 0.2 s of the 100 MHz wall clock
#ifndef SD_CHAIN_IN_FLIGHT
#define SD_CHAIN_IN_FLIGHT 4
#endif
constexpr int kInFlight = SD_CHAIN_IN_FLIGHT;  // wave loads a streaming wave keeps outstanding

enum { KIND_O_UP = 0, KIND_DOWN_QKV = 1 };

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

struct ChainRope {
  const float* cs;
  int max_pos;
  bf16_t* kc;
  bf16_t* vc;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset, q_len, Hq, Hkv, D;
};

struct ChainMerge {
  const float* part_o;   // null: phase 0 reads x0
  const float* part_ml;
  const int32_t* seq_lens;
  int pos_offset, q_len, G, Hkv, D, n_splits, split_tokens, m_pad;
};

struct ChainPhase {
  const bf16_t* W;
  const float* w_scale;
  const bf16_t* bias;
  int K, n_units, half;
};

struct ChainArgs {
  ChainPhase ph[2];
  unsigned* state;              // [0] launch counter (the tag), [1] failure word
  unsigned long long* edge;     // [M][H] tagged words
  int M, H;
  float eps;
  const bf16_t* x0;             // phase 0 input rows [M][K0] (unless merged from partials)
  int64_t ldx0;
  ChainMerge merge;
  bf16_t* h;                    // [M][H]: residual in, phase 0 result out
  const bf16_t* norm_w;         // RMSNorm in front of phase 1
  bf16_t* out1;                 // o_up: act [M][ld1]; down_qkv: q rows [M][ld1]
  int64_t ld1;
  int N1;                       // o_up: ffn columns
  ChainRope rope;
};

#ifdef SD_CHAIN_STAMPS
__device__ unsigned long long sd_chain_stamps[32];
// slot i by the first streaming wave, slot 16 + i by the first row wave, of workgroup 0
#define CSTAMP(i) \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) sd_chain_stamps[(i) + (threadIdx.x >= 512 ? 16 : 0)] = wall_clock64()
// every workgroup: [0] start, [1] phase 0 published, [2] phase 1 input gathered, [3] end
__device__ unsigned long long sd_chain_wg_stamps[512][4];
#define WSTAMP(i) \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) sd_chain_wg_stamps[blockIdx.x][i] = wall_clock64()
#else
#define CSTAMP(i)
#define WSTAMP(i)
#endif

__device__ __forceinline__ unsigned long long pack_word(bf16_t v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned short, v);
}

template <int KIND, bool W8>
__global__ __launch_bounds__(kMaxThreads) void chain_kernel(ChainArgs a) {
  constexpr int kWLoads = W8 ? kSeg / 2 : kSeg;
  constexpr int kWSteps = W8 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = a.M, H = a.H, K0 = a.ph[0].K;
  const int LDX0 = K0 + kXPad, LDX1 = H + kXPad;
  bf16_t* xs0 = reinterpret_cast<bf16_t*>(smem);                 // [M+1][LDX0]
  bf16_t* xs1 = xs0 + (size_t)(M + 1) * LDX0;                    // [M+1][LDX1]
  bf16_t* gs = xs1 + (size_t)(M + 1) * LDX1;                     // [H] norm weight
  float* red = reinterpret_cast<float*>(gs + H);                 // [2][8][256]
  // Epilogue inputs, parked by the row waves so that the streaming waves issue no global
  // load but their weight segments (a wait for anything else would drain their ring).
  float* e_res = red + 2 * kStreamWaves * 256;                   // [rows][units][16] residual
  float* e_b0 = e_res + kMaxRows * kMaxUnits * 16;               // [units][16] bias, phase 0
  float* e_s0 = e_b0 + kMaxUnits * 16;                           // [units][16] fp8 row scale
  float* e_b1 = e_s0 + kMaxUnits * 16;                           // [units][16]: rows r, r + 8
  float* e_s1 = e_b1 + kMaxUnits * 16;
  float* e_cs = e_s1 + kMaxUnits * 16;                           // [rows][units][16] cos | sin
  int* e_qp = reinterpret_cast<int*>(e_cs + kMaxRows * kMaxUnits * 16);  // [rows][2] position, page

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x, wg = (int)blockIdx.x;
  const unsigned tag = __builtin_amdgcn_readfirstlane(a.state[0]);

  // this workgroup's units and per-wave items (unit, segment) of both phases
  const int spw0 = K0 / (32 * kSeg * kStreamWaves), spw1 = H / (32 * kSeg * kStreamWaves);
  const int nu0 = wg < a.ph[0].n_units ? (a.ph[0].n_units - 1 - wg) / G + 1 : 0;
  const int nu1 = wg < a.ph[1].n_units ? (a.ph[1].n_units - 1 - wg) / G + 1 : 0;
  const int n0 = nu0 * spw0, n_total = n0 + nu1 * spw1;

  // zero rows: B-operand lanes of token rows >= M read them
  for (int c = tid * 8; c < LDX0; c += (int)blockDim.x * 8)
    *reinterpret_cast<bf16x8*>(xs0 + (size_t)M * LDX0 + c) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = tid * 8; c < LDX1; c += (int)blockDim.x * 8)
    *reinterpret_cast<bf16x8*>(xs1 + (size_t)M * LDX1 + c) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

  // Phase 0 input.  Attention partials are merged by every thread (one round trip before the
  // weights are asked for); plain rows are copied by the row waves alone.
  if (KIND == KIND_O_UP && a.merge.part_o) {
    const ChainMerge& g = a.merge;
    for (int it = tid; it < M * (K0 / 4); it += (int)blockDim.x) {
      const int m = it / (K0 / 4), c = (it - m * (K0 / 4)) * 4;
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
      bf16_t* xl = xs0 + (size_t)m * LDX0 + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) xl[j] = (bf16_t)(l > 0.f ? o[j] / l : 0.f);
    }
  }

  if (wave >= kStreamWaves) {
    // ------------------------------------------------------------ row wave: activations
    const int m = wave - kStreamWaves;  // < M by launch geometry
    // Every load of this wave that depends on nothing is asked for BEFORE the streaming
    // waves ask for weights (barrier below): a CU serves its requests in order, and a load
    // queued behind the ring's 128 KiB would wait microseconds for it.
    const bool copy_x0 = !(KIND == KIND_O_UP && a.merge.part_o);
    const bf16_t* xr = a.x0 + (size_t)m * a.ldx0;
    bf16_t* xl0 = xs0 + (size_t)m * LDX0;
    bf16x8 xv[16], gv[4];
    // unconditional loads from clamped addresses (a branch around a load makes the compiler
    // wait for everything in flight at the join); values are selected after the barrier
    if (copy_x0) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        xv[i] = *reinterpret_cast<const bf16x8*>(xr + lane * 8 + min(i * 512, K0 - 512));
    }
    const int r16 = lane & 15;
    unsigned short res_raw[2], b0_raw[2];
    float s0_v[2] = {1.f, 1.f};
    int n_epi[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const ChainPhase& ph = a.ph[0];
      const int ui = min((lane >> 4) + 4 * k, nu0 - 1);
      const int unit = wg + ui * G;
      n_epi[k] = (ph.half ? unit >> 1 : unit) * 16 + r16;
      const int n = min(n_epi[k], H - 1);
      const unsigned short* hp = reinterpret_cast<const unsigned short*>(a.h);
      res_raw[k] = hp[(size_t)m * H + n];
      // no bias: any valid address, the value is dropped below
      b0_raw[k] = (ph.bias ? reinterpret_cast<const unsigned short*>(ph.bias) : hp)[n];
      if (W8) s0_v[k] = ph.w_scale[n];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      gv[i] = *reinterpret_cast<const bf16x8*>(a.norm_w + lane * 8 + min(i * 512, H - 512));
    int sl = 0;
    if (KIND == KIND_DOWN_QKV) sl = a.rope.seq_lens[m / a.rope.q_len];
    __syncthreads();                                   // the streaming waves may ask for weights
    if (copy_x0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = i * 512 + lane * 8;
        if (c < K0) *reinterpret_cast<bf16x8*>(xl0 + c) = xv[i];
      }
      for (int c0 = 8192; c0 < K0; c0 += 8192) {       // rows wider than one batch
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = c0 + i * 512 + lane * 8;
          if (c < K0) xv[i] = *reinterpret_cast<const bf16x8*>(xr + c);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = c0 + i * 512 + lane * 8;
          if (c < K0) *reinterpret_cast<bf16x8*>(xl0 + c) = xv[i];
        }
      }
    }
    __syncthreads();                                   // phase 0 input staged
    CSTAMP(1);
    // Epilogue inputs of phase 0: lane -> (unit index lane >> 4 (+ 4), tile row lane & 15)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ui = (lane >> 4) + 4 * k;
      if (ui < nu0) {
        const bool in = n_epi[k] < H;
        e_res[(m * kMaxUnits + ui) * 16 + r16] =
            in ? (float)__builtin_bit_cast(bf16_t, res_raw[k]) : 0.f;
        if (m == 0) {
          e_b0[ui * 16 + r16] =
              (in && a.ph[0].bias) ? (float)__builtin_bit_cast(bf16_t, b0_raw[k]) : 0.f;
          e_s0[ui * 16 + r16] = in ? s0_v[k] : 1.f;
        }
      }
    }
    if (nu1 > 0) {
      const int r = lane & 15;
      int q_pos = -1, q_page = 0;
      if (KIND == KIND_DOWN_QKV) {
        const ChainRope& rp = a.rope;
        const int b = m / rp.q_len;
        q_pos = sl + rp.pos_offset + (m - b * rp.q_len);
        if (q_pos >= 0) q_page = rp.block_tables[(size_t)b * rp.max_blocks + (q_pos >> 4)];
        if (lane == 0) {
          e_qp[m * 2] = q_pos;
          e_qp[m * 2 + 1] = q_page;
        }
      }
      for (int ui = lane >> 4; ui < nu1; ui += 4) {
        const ChainPhase& ph = a.ph[1];
        const int unit = wg + ui * G;
        int idx, lim;  // source row of tile row r, and the row count
        if (KIND == KIND_O_UP) {
          const int n = unit * 8 + (r & 7);
          idx = n < a.N1 ? (r < 8 ? n : a.N1 + n) : -1;
          lim = 2 * a.N1;
        } else {
          const ChainRope& rp = a.rope;
          const int D = rp.D, tph = D / 16;
          const int head = unit / tph, i = (unit % tph) * 8 + (r & 7);
          idx = head * D + i + (r < 8 ? 0 : D / 2);
          lim = (rp.Hq + 2 * rp.Hkv) * D;
          float cs = 0.f;
          if (q_pos >= 0 && head < rp.Hq + rp.Hkv && rp.cs)
            cs = rp.cs[(size_t)min(q_pos, rp.max_pos - 1) * D + (r < 8 ? i : D / 2 + i)];
          e_cs[(m * kMaxUnits + ui) * 16 + r] = cs;
        }
        if (m == 0) {
          const bool in = idx >= 0 && idx < lim;
          e_b1[ui * 16 + r] = (ph.bias && in) ? (float)ph.bias[idx] : 0.f;
          e_s1[ui * 16 + r] = (W8 && in) ? ph.w_scale[idx] : 1.f;
        }
      }
      // the norm weight of phase 1 (row wave 0 parks it for all row waves: they read it after
      // phase 0's unit-end barriers below)
      if (m == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = i * 512 + lane * 8;
          if (c < H) *reinterpret_cast<bf16x8*>(gs + c) = gv[i];
        }
        for (int c = 2048 + lane * 8; c < H; c += 512)
          *reinterpret_cast<bf16x8*>(gs + c) = *reinterpret_cast<const bf16x8*>(a.norm_w + c);
      }
    }
    for (int u = 0; u < nu0; ++u) __syncthreads();     // phase 0 unit ends
    CSTAMP(2);
    if (nu1 > 0) {
      // Gather row m of phase 0's result: words [m][c], c = i*512 + lane*8 + j, in the
      // column order of the fused norm's sums.
      const unsigned long long* er = a.edge + (size_t)m * H;
      bf16_t* xl = xs1 + (size_t)m * LDX1;
      float s2 = 0.f;
      bool ok = true;
      const unsigned long long t0 = wall_clock64();
      for (int c0 = 0; c0 < H; c0 += 1024) {
        unsigned long long w[16];
        for (;;) {
          bool all = true;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int c = c0 + (i >> 3) * 512 + lane * 8 + (i & 7);
            w[i] = __hip_atomic_load(er + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) all &= (unsigned)(w[i] >> 32) == tag;
          if (__all(all)) break;
          if (wall_clock64() - t0 > kPollTicks ||
              __hip_atomic_load(a.state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          bf16x8 v;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = __builtin_bit_cast(bf16_t, (unsigned short)w[i2 * 8 + j]);
            const float f = (float)v[j];
            s2 += f * f;
          }
          *reinterpret_cast<bf16x8*>(xl + c0 + i2 * 512 + lane * 8) = v;
        }
      }
      CSTAMP(3);
      if (m == 0) { WSTAMP(2); }
      if (!ok && lane == 0)
        __hip_atomic_store(a.state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float rs = rsqrtf(wave_sum(s2) / (float)H + a.eps);
      for (int c = lane * 8; c < H; c += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xl + c);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gs + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(round_bf16((float)v[j] * rs) * (float)g[j]);
        *reinterpret_cast<bf16x8*>(xl + c) = o;
      }
      __syncthreads();                                 // phase 1 input staged
      CSTAMP(4);
      for (int u = 0; u < nu1; ++u) __syncthreads();   // phase 1 unit ends
    }
    return;
  }

  // -------------------------------------------------------------- streaming waves
  // item gi of this wave: phase, unit, segment
  auto decode = [&](int gi, int& p, int& ui, int& unit, int& seg, int& s) {
    p = gi >= n0;
    const int i = p ? gi - n0 : gi, spw = p ? spw1 : spw0;
    ui = i / spw;
    s = i - ui * spw;
    unit = wg + ui * G;
    seg = wave + kStreamWaves * s;
  };
  // Three ring slots of one segment each: deep enough to take in the second matrix while the
  // first phase's outputs cross the chip.  Requests are THROTTLED: a wave keeps at most
  // kInFlight wave loads (1 KiB each; twice as many for half-tile items, whose loads carry
  // half the bytes) outstanding.  16..32 KiB in flight per CU already stream at the full HBM
  // rate (experiments/inflight_latency.hip: 6.2 TB/s at 16 KiB, 5.5 TB/s at 128 KiB), and a
  // small load of another wave of the CU then returns in 0.6..1 us instead of 8.
  bf16x8 r0[kWLoads], r1[kWLoads], r2[kWLoads];
  // lanes of a half-tile unit's other row half read their partner lane's bytes (same cache
  // line) and are zeroed when used
  auto issue = [&](bf16x8(&r)[kWLoads], int gi) {
    if (gi >= n_total) return;
    int p, ui, unit, seg, s;
    decode(gi, p, ui, unit, seg, s);
    const ChainPhase& ph = a.ph[p];
    const int KS = ph.K / 32;
    const int tile = ph.half ? unit >> 1 : unit;
    const bool live = !ph.half || ((lane >> 3) & 1) == (unit & 1);
    const size_t off = ((size_t)tile * (KS / kWSteps) + (size_t)seg * kWLoads) * 512;
    const bf16_t* w = ph.W + off + (live ? lane : lane ^ 8) * 8;
    if (ph.half) {
#pragma unroll
      for (int j = 0; j < kWLoads; ++j) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kInFlight - 1) : "memory");
        r[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w + (size_t)j * 512));
      }
    } else {
#pragma unroll
      for (int j = 0; j < kWLoads; ++j) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight - 1) : "memory");
        r[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w + (size_t)j * 512));
      }
    }
  };

  // epilogue role of this thread (waves 0..3): row nl of the tile, token m
  const int nl = 4 * (tid >> 6) + (tid & 3), m = (tid >> 2) & 15;
  const bool epi_row = tid < 256 && m < M;
  CSTAMP(0);
  if (wave == 0) { WSTAMP(0); }
  __syncthreads();  // the row waves have asked for their activations
  issue(r0, 0);
  __syncthreads();  // phase 0 input staged
  CSTAMP(1);

  const int xc = lane & 15, xq = lane >> 4;
  const bf16_t* xrow0 = xs0 + (size_t)(xc < M ? xc : M) * LDX0 + 8 * xq;
  const bf16_t* xrow1 = xs1 + (size_t)(xc < M ? xc : M) * LDX1 + 8 * xq;
  int parity = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

  auto unit_end = [&](int p, int ui, int unit) {
    float* rbuf = red + parity * kStreamWaves * 256;
    parity ^= 1;
    *reinterpret_cast<f32x4*>(rbuf + wave * 256 + lane * 4) = acc;
    __syncthreads();
    if (!epi_row) return;
    auto total = [&](int e) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kStreamWaves; ++w) sum += rbuf[w * 256 + e];
      return sum;
    };
    const ChainPhase& ph = a.ph[p];
    if (p == 0) {
      const int tile = ph.half ? unit >> 1 : unit;
      const int n = tile * 16 + nl;
      if ((!ph.half || (nl >> 3) == (unit & 1)) && n < H) {
        const float pre0 = e_b0[ui * 16 + nl], sc0 = e_s0[ui * 16 + nl];
        const float pre1 = e_res[(m * kMaxUnits + ui) * 16 + nl];
        const float y = total(tid) * sc0 + pre0;
        const bf16_t hv = (bf16_t)(round_bf16(y) + pre1);
        a.h[(size_t)m * H + n] = hv;
        __hip_atomic_store(a.edge + (size_t)m * H + n, pack_word(hv, tag), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (KIND == KIND_O_UP) {
      const int n = unit * 8 + nl;
      if (nl < 8 && n < a.N1) {
        const float pre0 = e_b1[ui * 16 + nl], pre1 = e_b1[ui * 16 + nl + 8];
        const float sc0 = e_s1[ui * 16 + nl], sc1 = e_s1[ui * 16 + nl + 8];
        float g = total(tid) * sc0 + pre0, u = total(tid + 128) * sc1 + pre1;
        g = round_bf16(silu_f(round_bf16(g)));
        a.out1[(size_t)m * a.ld1 + n] = (bf16_t)(g * round_bf16(u));
      }
    } else if (nl < 8 && e_qp[m * 2] >= 0) {
      const ChainRope& rp = a.rope;
      const int q_pos = e_qp[m * 2], q_page = e_qp[m * 2 + 1];
      const int D = rp.D, half = D / 2, tiles_per_head = D / 16;
      const int head = unit / tiles_per_head, i = (unit % tiles_per_head) * 8 + nl;
      const float pre0 = e_b1[ui * 16 + nl], pre1 = e_b1[ui * 16 + nl + 8];
      const float sc0 = e_s1[ui * 16 + nl], sc1 = e_s1[ui * 16 + nl + 8];
      const float pre2 = e_cs[(m * kMaxUnits + ui) * 16 + nl];
      const float pre3 = e_cs[(m * kMaxUnits + ui) * 16 + nl + 8];
      float lo = round_bf16(total(tid) * sc0 + pre0), hi = round_bf16(total(tid + 128) * sc1 + pre1);
      if (head < rp.Hq + rp.Hkv && rp.cs) {
        const float nlo = lo * pre2 - hi * pre3, nhi = hi * pre2 + lo * pre3;
        lo = nlo;
        hi = nhi;
      }
      if (head < rp.Hq) {
        bf16_t* q = a.out1 + (size_t)m * a.ld1 + (size_t)head * D;
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
  };

  // one item: its ring slot through the MFMAs, the unit's end if it is the unit's last segment,
  // then the slot's refill three items ahead
  auto half_step = [&](bf16x8(&r)[kWLoads], int gi, bool refill) {
    int p, ui, unit, seg, s;
    decode(gi, p, ui, unit, seg, s);
    if (gi == n0 && p == 1) {
      CSTAMP(2);
      __syncthreads();  // phase 1 input staged by the row waves
      CSTAMP(3);
    }
    const bool live = !a.ph[p].half || ((lane >> 3) & 1) == (unit & 1);
    bf16x8 cur[kWLoads];
#pragma unroll
    for (int j = 0; j < kWLoads; ++j) cur[j] = live ? r[j] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (s == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* xp = (p ? xrow1 : xrow0) + (size_t)seg * kSeg * 32;
#pragma unroll
    for (int j = 0; j < kSeg; ++j) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(xp + j * 32);
      bf16x8 w;
      if constexpr (W8) {
        const u32x4 raw = __builtin_bit_cast(u32x4, cur[j >> 1]);
        w = fp8x8_to_bf16(raw[2 * (j & 1)], raw[2 * (j & 1) + 1]);
      } else {
        w = cur[j];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc, 0, 0, 0);
    }
    if (s == (p ? spw1 : spw0) - 1) unit_end(p, ui, unit);
    if (refill) issue(r, gi + 3);
  };

  // The first item is used (and, one segment per unit, published) before the ring fills up.
  half_step(r0, 0, false);
  if (wave == 0) { WSTAMP(1); }
  issue(r1, 1);
  issue(r2, 2);
  issue(r0, 3);
  int gi = 1;
  for (; gi + 2 < n_total; gi += 3) {
    half_step(r1, gi, true);
    half_step(r2, gi + 1, true);
    half_step(r0, gi + 2, true);
  }
  if (gi < n_total) half_step(r1, gi, false);
  if (gi + 1 < n_total) half_step(r2, gi + 1, false);
  CSTAMP(4);
  if (wave == 0) { WSTAMP(3); }
  // the next launch's tag (every workgroup read this one's before it published anything, and
  // workgroup 0 got here only after gathering every workgroup's words)
  if (wg == 0 && tid == 0) a.state[0] = tag + 1u;
}

int chain_cus() {
  static std::once_flag once;
  static int cus = 0;
  std::call_once(once, [] {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  });
  return cus;
}

size_t chain_lds(int M, int K0, int H) {
  return (size_t)(M + 1) * (K0 + kXPad) * 2 + (size_t)(M + 1) * (H + kXPad) * 2 + (size_t)H * 2 +
         kRedBytes + kEpiBytes;
}

template <int KIND, bool W8>
int launch_chain(ChainArgs& a, hipStream_t st) {
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<KIND, W8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess) {
    sd_set_error("sd_chain: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    return SD_ERR_HIP;
  }
  const int cus = chain_cus();
  SD_REQUIRE(cus >= 2, SD_ERR_HIP, "sd_chain: no device");
  // Every workgroup owns units of phase 0 (so workgroup 0, which gathers all of phase 0's
  // words, leaves only after every workgroup has read the tag), at most one per CU.
  int grid = a.ph[0].n_units < cus ? a.ph[0].n_units : cus;
  grid &= ~1;  // a workgroup keeps one row half for all its half-tile units
  const int most = a.ph[0].n_units > a.ph[1].n_units ? a.ph[0].n_units : a.ph[1].n_units;
  SD_REQUIRE(grid >= 2 && (most + grid - 1) / grid <= kMaxUnits, SD_ERR_UNSUPPORTED,
             "sd_chain: %d units on %d workgroups (at most %d each)", most, grid, kMaxUnits);
  hipLaunchKernelGGL((chain_kernel<KIND, W8>), dim3(grid), dim3(kStreamThreads + a.M * SD_WAVE),
                     chain_lds(a.M, a.ph[0].K, a.H), st, a);
  SD_CHECK_LAUNCH("sd_chain");
  return SD_OK;
}

bool chain_shape_ok(int M, int K0, int H, int N_units1) {
  return M >= 1 && M <= kMaxRows && K0 >= 2048 && K0 % 2048 == 0 && H >= 2048 && H % 2048 == 0 &&
         H <= 8192 && N_units1 >= 2 && chain_lds(M, K0, H) <= 160 * 1024;
}

void chain_state(void* state, ChainArgs& a) {
  a.state = reinterpret_cast<unsigned*>(state);
  a.edge = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 256);
}

}  // namespace

extern "C" size_t sd_chain_state_bytes(int hidden) {
  if (hidden < 1) return 0;
  return 256 + (size_t)kMaxRows * hidden * 8;
}

extern "C" int sd_chain_state_init(void* state, size_t bytes, int hidden, sd_stream_t stream) {
  SD_REQUIRE(state && hidden >= 1 && bytes >= sd_chain_state_bytes(hidden) &&
                 (reinterpret_cast<uintptr_t>(state) & 255u) == 0,
             SD_ERR_INVALID, "sd_chain_state_init: need a 256-byte aligned buffer of %zu bytes",
             sd_chain_state_bytes(hidden));
  hipError_t e = hipMemsetAsync(state, 0, sd_chain_state_bytes(hidden), (hipStream_t)stream);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_chain_state_init: %s", hipGetErrorString(e));
  const unsigned one = 1u;  // tags start at 1: a zeroed word never matches
  e = hipMemcpyAsync(state, &one, 4, hipMemcpyHostToDevice, (hipStream_t)stream);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_chain_state_init: %s", hipGetErrorString(e));
  e = hipStreamSynchronize((hipStream_t)stream);  // `one` lives on this stack frame
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_chain_state_init: %s", hipGetErrorString(e));
  return SD_OK;
}

extern "C" int sd_chain_errors(const void* state, int* failed) {
  SD_REQUIRE(state && failed, SD_ERR_INVALID, "sd_chain_errors: null pointer");
  unsigned v = 0;
  const hipError_t e = hipMemcpy(&v, reinterpret_cast<const char*>(state) + 4, 4, hipMemcpyDeviceToHost);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_chain_errors: %s", hipGetErrorString(e));
  *failed = (int)v;
  return SD_OK;
}

extern "C" int sd_chain_supported(int rows, int hidden, int q_dim, int ffn) {
  // o_up: K0 = q_dim; down_qkv: K0 = ffn
  return chain_shape_ok(rows, q_dim, hidden, 2) && chain_shape_ok(rows, ffn, hidden, 2) &&
         hidden % 16 == 0 && ffn % 8 == 0;
}

extern "C" int sd_chain_o_up(void* state, const void* o_w, const float* o_s, const void* o_b,
                             const void* attn, int64_t ld_attn, const void* attn_workspace,
                             size_t attn_workspace_bytes, const int32_t* seq_lens, int pos_offset,
                             int B, int q_len, int Hq, int Hkv, int D, int max_ctx,
                             int split_tokens, void* h, int hidden, const void* norm_w, float eps,
                             const void* up_w, const float* up_s, const void* up_b, void* act,
                             int64_t ld_act, int ffn, sd_stream_t stream) {
  SD_REQUIRE(state && o_w && h && norm_w && up_w && act && (attn || attn_workspace), SD_ERR_INVALID,
             "sd_chain_o_up: null pointer");
  SD_REQUIRE(B >= 1 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && (D == 64 || D == 128),
             SD_ERR_INVALID, "sd_chain_o_up: bad shape");
  const int M = B * q_len, KA = Hq * D;
  SD_REQUIRE(chain_shape_ok(M, KA, hidden, (ffn + 7) / 8) && ld_act >= ffn, SD_ERR_UNSUPPORTED,
             "sd_chain_o_up: rows %d, Hq*D %d, hidden %d outside the fused form (rows <= %d, "
             "widths multiples of 2048)", M, KA, hidden, kMaxRows);
  SD_REQUIRE((o_s != nullptr) == (up_s != nullptr), SD_ERR_INVALID,
             "sd_chain_o_up: both matrices fp8 or neither");
  ChainArgs a{};
  chain_state(state, a);
  const int o_tiles = (hidden + 15) / 16;
  a.ph[0] = ChainPhase{(const bf16_t*)o_w, o_s, (const bf16_t*)o_b, KA,
                       o_tiles < 256 ? 2 * o_tiles : o_tiles, o_tiles < 256};
  a.ph[1] = ChainPhase{(const bf16_t*)up_w, up_s, (const bf16_t*)up_b, hidden, (ffn + 7) / 8, 0};
  a.M = M; a.H = hidden; a.eps = eps;
  a.x0 = (const bf16_t*)attn; a.ldx0 = ld_attn;
  if (!attn) {
    SD_REQUIRE(seq_lens && split_tokens >= 128 && max_ctx >= 1, SD_ERR_INVALID,
               "sd_chain_o_up: merging partials needs seq_lens, split_tokens, max_ctx");
    const int G = Hq / Hkv;
    const int n_splits = (max_ctx + split_tokens - 1) / split_tokens;
    SD_REQUIRE(q_len * G <= 64 && n_splits >= 1 && n_splits <= kMergeMaxSplits, SD_ERR_UNSUPPORTED,
               "sd_chain_o_up: %d splits (at most %d)", n_splits, kMergeMaxSplits);
    const int m_pad = q_len * G <= 32 ? 32 : 64;
    const size_t rows = (size_t)B * Hkv * n_splits * m_pad;
    SD_REQUIRE(attn_workspace_bytes >= rows * (D + 2) * sizeof(float), SD_ERR_WORKSPACE,
               "sd_chain_o_up: workspace smaller than the partials");
    a.merge = ChainMerge{(const float*)attn_workspace, (const float*)attn_workspace + rows * D,
                         seq_lens, pos_offset, q_len, G, Hkv, D, n_splits, split_tokens, m_pad};
  } else {
    SD_REQUIRE(ld_attn >= KA && ld_attn % 8 == 0 && sd_aligned16(attn), SD_ERR_INVALID,
               "sd_chain_o_up: attention rows must be 16-byte aligned");
  }
  a.h = (bf16_t*)h;
  a.norm_w = (const bf16_t*)norm_w;
  a.out1 = (bf16_t*)act; a.ld1 = ld_act; a.N1 = ffn;
  return o_s ? launch_chain<KIND_O_UP, true>(a, (hipStream_t)stream)
             : launch_chain<KIND_O_UP, false>(a, (hipStream_t)stream);
}

extern "C" int sd_chain_down_qkv(void* state, const void* down_w, const float* down_s,
                                 const void* down_b, const void* act, int64_t ld_act, int ffn,
                                 void* h, int hidden, const void* norm_w, float eps,
                                 const void* qkv_w, const float* qkv_s, const void* qkv_b,
                                 void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                                 void* k_cache, void* v_cache, const int32_t* block_tables,
                                 int max_blocks_per_seq, const int32_t* seq_lens, int pos_offset,
                                 int B, int q_len, int Hq, int Hkv, int D, sd_stream_t stream) {
  SD_REQUIRE(state && down_w && act && h && norm_w && qkv_w && q_out && k_cache && v_cache &&
                 block_tables && seq_lens, SD_ERR_INVALID, "sd_chain_down_qkv: null pointer");
  SD_REQUIRE(B >= 1 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && D % 16 == 0 && D >= 16, SD_ERR_INVALID,
             "sd_chain_down_qkv: bad shape (D=%d)", D);
  SD_REQUIRE(!rope_cs || max_pos >= 1, SD_ERR_INVALID, "sd_chain_down_qkv: max_pos < 1");
  SD_REQUIRE(ldq >= (int64_t)Hq * D, SD_ERR_INVALID, "sd_chain_down_qkv: ldq too small");
  const int M = B * q_len, n_qkv_tiles = (Hq + 2 * Hkv) * D / 16;
  SD_REQUIRE(chain_shape_ok(M, ffn, hidden, n_qkv_tiles) && ld_act >= ffn && ld_act % 8 == 0 &&
                 sd_aligned16(act), SD_ERR_UNSUPPORTED,
             "sd_chain_down_qkv: rows %d, ffn %d, hidden %d outside the fused form (rows <= %d, "
             "widths multiples of 2048)", M, ffn, hidden, kMaxRows);
  SD_REQUIRE((down_s != nullptr) == (qkv_s != nullptr), SD_ERR_INVALID,
             "sd_chain_down_qkv: both matrices fp8 or neither");
  ChainArgs a{};
  chain_state(state, a);
  const int d_tiles = (hidden + 15) / 16;
  a.ph[0] = ChainPhase{(const bf16_t*)down_w, down_s, (const bf16_t*)down_b, ffn,
                       d_tiles < 256 ? 2 * d_tiles : d_tiles, d_tiles < 256};
  a.ph[1] = ChainPhase{(const bf16_t*)qkv_w, qkv_s, (const bf16_t*)qkv_b, hidden, n_qkv_tiles, 0};
  a.M = M; a.H = hidden; a.eps = eps;
  a.x0 = (const bf16_t*)act; a.ldx0 = ld_act;
  a.h = (bf16_t*)h;
  a.norm_w = (const bf16_t*)norm_w;
  a.out1 = (bf16_t*)q_out; a.ld1 = ldq;
  a.rope = ChainRope{rope_cs, max_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, block_tables,
                     max_blocks_per_seq, seq_lens, pos_offset, q_len, Hq, Hkv, D};
  return down_s ? launch_chain<KIND_DOWN_QKV, true>(a, (hipStream_t)stream)
                : launch_chain<KIND_DOWN_QKV, false>(a, (hipStream_t)stream);
}

#ifdef SD_CHAIN_STAMPS
extern "C" int sd_debug_read_chain_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_chain_stamps), sizeof(sd_chain_stamps));
}
extern "C" int sd_debug_read_chain_wg_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_chain_wg_stamps), sizeof(sd_chain_wg_stamps));
}
#endif
