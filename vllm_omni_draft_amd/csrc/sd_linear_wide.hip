// Linear layers at 17..256 token rows (batched draft passes, batched verify) over the same
// pre-tiled weights as the <= 16-row kernel.
//
// wide_gemm_kernel: a workgroup of 8 waves owns 16 weight tiles (256 weight rows; wave w the
// tiles w and w + 8 of the group) and one K range.  The activations of the K range pass through LDS in
// 128-column chunks (double buffered), shared by the 8 waves; the weights go HBM -> VGPR ->
// MFMA as in the narrow kernel, each A fragment feeding one MFMA per 16-row token tile.  With
// M = 192 rows an A fragment is used 12 times, so the kernel sits at the MFMA / LDS / HBM
// balance point rather than being a pure stream: the B fragments are read through a register
// ring whose order is pinned for the instruction scheduler, and no load sits inside a branch
// (see the kernel).  Tried and dropped: groups of 8..15 tiles to fill the CUs in one round
// (waves without a second tile still pay for it); 4-wave workgroups of 8 tiles, two per CU,
// with 64-column chunks (the x staging latency of a chunk is then longer than the chunk).
//
// wide_gemm4_kernel (further down) serves the many-row passes of some matrices with another
// chunk structure on four waves; same tiles, same sums, same bits.
//
// Matrices with few tiles split K over several workgroups.  Partial sums go to a workspace
// as f32 [split][row][tile row] and wide_epilogue_kernel adds the splits in index order and
// applies the epilogue, so a row's arithmetic depends on (N, K) only: never on how many rows
// share the pass.
//
// Build switches, for diagnostic and experiment builds only (the product library defines none):
//   SD_WIDE_STAMPS   in-kernel phase clocks of both GEMM kernels (make stamps,
//                    experiments/wide_stamps.py)
//   SD_WIDE_SWITCH   a run-time choice between the two many-row kernels (make wide-switch,
//                    experiments/bench_wide_shapes.py)
//   SD_WIDE_EXP      a quick build of one row count's bf16 instantiations (SD_WIDE_EXP_MT token
//                    tiles, default 12; SD_WIDE_BRING sets the B ring): experiments/build_wide_exp.sh
#include <mutex>
#include <type_traits>

#include "sd_common.h"

namespace {

constexpr int kWideThreads = 512;
constexpr int kWideWaves = kWideThreads / SD_WAVE;
constexpr int kTilesPerGroup = kWideWaves * 2;              // 16 tiles = 256 weight rows
// Matrices with fewer than kPairTiles tiles run with one tile per wave (8-tile groups): from
// kSmallTiles up that gives 128..255 workgroups in one pass with the epilogue inside; below,
// K is split (in shorter pieces than for the large matrices).
constexpr int kSmallTiles = 1024;
constexpr int kPairTiles = 2048;
constexpr int kChunkSteps = 4;                              // k-steps (32 columns) per LDS chunk
constexpr int kChunkCols = kChunkSteps * 32;
constexpr int kXLD = kChunkCols + 8;                        // padded LDS row (elements)
#ifndef SD_WIDE_BRING
#define SD_WIDE_BRING 4
#endif
constexpr int kBRing = SD_WIDE_BRING;                       // B fragments in flight per wave
constexpr int kFourWaveMT = 8;                              // token tiles from which 8-tile groups run on 4 waves
constexpr int EPI_QKV_ROPE = 100;

__device__ __forceinline__ float silu_w(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_tanh_w(float v) {
  const float c = 0.7978845608028654f;
  return 0.5f * v * (1.0f + tanhf(c * (v + 0.044715f * v * v * v)));
}

struct WideRope {
  const float* cs;
  int max_pos;
  bf16_t* kc;
  bf16_t* vc;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset, q_len, Hq, Hkv, D;
  RopePos mp;  // multimodal rotary positions (all null: the cache position)
};

struct WideEpiArgs {
  const float* part;  // [S][M][NR]
  int S, M, NR, N;
  const float* w_scale;  // fp8 weights: f32 scale per source row (indexed like bias); else null
  const float* x_scale;  // fp8 activations: f32 scale per token row; else null
  const bf16_t* bias;
  const bf16_t* residual;
  void* out;
  int64_t ldo;
  WideRope rope;
};

struct WideArgs {
  const bf16_t* W;  // tiled [n_tiles][K/32][64][8] bf16, or [n_tiles][K/64][64][16] fp8
  const bf16_t* x;
  int64_t ldx;
  float* part;      // [S][M][NR]
  int M, K, n_tiles, ks_per_split;
  int group_tiles;  // weight tiles per workgroup (8..16)
  WideEpiArgs epi;  // direct epilogue (one split): EPI >= 0
  const float* x_inv;  // fp8 activations: 448 / max |x[m]| per token row (wide_gemm_a8_kernel)
};

constexpr int EPI_PARTIAL = -1;  // several K splits: partial sums to the workspace

#ifdef SD_WIDE_STAMPS
// Debug build only: ticks (10 ns) wave 0 of workgroup 0 spent per phase, summed over the chunks:
// [0] waiting for the chunk's weights, [1] LDS reads + MFMAs, [2] waiting for the next chunk's x,
// [3] storing it, [4] the barrier, [5] issuing the next chunk's loads, [6] whole loop, [7] the
// loop in core clock cycles.
__device__ unsigned long long sd_wide_stamps[8][8];  // [wave][phase]
#define SD_WSTAMP(var) \
  __builtin_amdgcn_sched_barrier(0); \
  const unsigned long long var = wall_clock64(); \
  __builtin_amdgcn_sched_barrier(0)
#endif

// W8: fp8 (e4m3) weight tiles, two k-steps per 16-byte lane load, converted to bf16 when a
// chunk's fragments are taken out of the prefetch registers; the row scales are applied by the
// epilogue.
template <int MT, int EPI, int TPW, bool W8>  // MT token tiles of 16 rows; TPW tiles per wave
__global__ __launch_bounds__(kWideThreads) void wide_gemm_kernel(WideArgs a) {
  constexpr int kThreads = kWideThreads;
  constexpr int WAVES = kWideWaves;
  constexpr int kTilesPerWave = TPW;
  constexpr int kALoads = W8 ? kChunkSteps / 2 : kChunkSteps;  // 16-byte loads per tile and chunk
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem_raw);  // [2][MT*16][kXLD]
  constexpr int kRows = MT * 16;
  constexpr int kPieces = kRows * (kChunkCols / 8) / kThreads;  // 16-byte pieces per thread
  static_assert(kRows * (kChunkCols / 8) % kThreads == 0, "MT must be even");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = a.K / 32;
  const int NR = a.n_tiles * 16;
  const int ks_lo = blockIdx.y * a.ks_per_split;
  const int ks_hi = min(ks_lo + a.ks_per_split, KS);
  const int n_chunks = (ks_hi - ks_lo + kChunkSteps - 1) / kChunkSteps;
  // tile t of this wave: group base + wave + 8 t, if inside the group and the matrix
  const int tile0 = blockIdx.x * a.group_tiles + wave;
  auto tile_of = [&](int t) { return tile0 + t * WAVES; };
  auto tile_ok = [&](int t) {
    return wave + t * WAVES < a.group_tiles && tile_of(t) < a.n_tiles;
  };

  // Every load below is unconditional on a clamped address: a load inside a branch makes the
  // compiler wait for all outstanding loads at the join, which would drain the weight
  // prefetch at every chunk.  Rows >= M and tiles outside the group produce sums nobody
  // stores; columns past the split's K range are zeroed in x, so the weights read there
  // (clamped, finite) contribute nothing.
  bf16x8 x_stage[kPieces];
  auto load_x = [&](int chunk) {
    const int col0 = (ks_lo + chunk * kChunkSteps) * 32;
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const int p = tid + i * kThreads;
      const int row = min(p / (kChunkCols / 8), a.M - 1);
      const int col = col0 + (p % (kChunkCols / 8)) * 8;
      x_stage[i] = *reinterpret_cast<const bf16x8*>(a.x + (size_t)row * a.ldx + min(col, a.K - 8));
    }
  };
  auto store_x = [&](int buf, int chunk) {  // the zeroing happens here: no use of the data earlier
    const int col0 = (ks_lo + chunk * kChunkSteps) * 32;
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const int p = tid + i * kThreads;
      const int row = p / (kChunkCols / 8), c8 = p % (kChunkCols / 8);
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      *reinterpret_cast<bf16x8*>(xs + ((size_t)buf * kRows + row) * kXLD + c8 * 8) =
          col0 + c8 * 8 < ks_hi * 32 ? x_stage[i] : z;
    }
  };
  bf16x8 a_nxt[kTilesPerWave][kALoads];
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
      for (int j = 0; j < kALoads; ++j) {
        const int tile = min(tile_of(t), a.n_tiles - 1);
        // 512 bf16 (1 KiB) per wave load either way; chunks start at even k-steps
        const int step = W8 ? min((ks_lo + chunk * kChunkSteps) / 2 + j, KS / 2 - 1)
                            : min(ks_lo + chunk * kChunkSteps + j, KS - 1);
        a_nxt[t][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(
            a.W + ((size_t)tile * (W8 ? KS / 2 : KS) + step) * 512 + lane * 8));
      }
  };

  f32x4 acc[kTilesPerWave][MT];
#pragma unroll
  for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_x(0);
  load_a(0);
  store_x(0, 0);
  __syncthreads();
  const int xc = lane & 15, xq = lane >> 4;
#ifdef SD_WIDE_STAMPS
  unsigned long long w_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long w_begin = wall_clock64();
  const unsigned long long w_cyc0 = clock64();
#endif
  for (int c = 0; c < n_chunks; ++c) {
#ifdef SD_WIDE_STAMPS
    SD_WSTAMP(w0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SD_WSTAMP(w1);
    w_acc[0] += w1 - w0;
#endif
    bf16x8 a_cur[kTilesPerWave][kChunkSteps];
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
      for (int j = 0; j < kChunkSteps; ++j) {
        if constexpr (W8) {
          const u32x4 raw = __builtin_bit_cast(u32x4, a_nxt[t][j >> 1]);
          a_cur[t][j] = fp8x8_to_bf16(raw[2 * (j & 1)], raw[2 * (j & 1) + 1]);
        } else {
          a_cur[t][j] = a_nxt[t][j];
        }
      }
    if (c + 1 < n_chunks) {
      // x first (cache hits, stored to LDS at the end of this chunk), then the weights, which
      // may stay in flight past that store: loads return in issue order
      load_x(c + 1);
      __builtin_amdgcn_sched_barrier(0);
      load_a(c + 1);
    }
#ifdef SD_WIDE_STAMPS
    SD_WSTAMP(w1b);
    w_acc[5] += w1b - w1;
#endif
    const bf16_t* xb = xs + (size_t)(c & 1) * kRows * kXLD + (size_t)xc * kXLD + 8 * xq;
    // The chunk's B fragments, (k-step j, token tile mt) in order, through a ring of kBRing
    // registers: a fragment is requested kBRing MFMA pairs before its use, so the LDS latency
    // hides behind the matrix pipe instead of stalling every pair.
    constexpr int kReads = kChunkSteps * MT;
    auto read_b = [&](int idx) {
      return *reinterpret_cast<const bf16x8*>(xb + (size_t)(idx % MT) * 16 * kXLD + (idx / MT) * 32);
    };
    bf16x8 ring[kBRing];
#pragma unroll
    for (int i = 0; i < kBRing; ++i) ring[i] = read_b(i);
#pragma unroll
    for (int idx = 0; idx < kReads; ++idx) {
      const bf16x8 b = ring[idx % kBRing];
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t)
        acc[t][idx % MT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[t][idx / MT], b,
                                                                   acc[t][idx % MT], 0, 0, 0);
      if (idx + kBRing < kReads) ring[idx % kBRing] = read_b(idx + kBRing);
    }
    // Pin that order for the instruction scheduler (it otherwise sinks every LDS read to just
    // before its MFMA pair): kBRing reads, then {MFMA pair, read} ..., then the last pairs.
#pragma unroll
    for (int i = 0; i < kBRing; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int i = 0; i < kReads - kBRing; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, kTilesPerWave, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < kBRing; ++i) __builtin_amdgcn_sched_group_barrier(0x008, kTilesPerWave, 0);
#ifdef SD_WIDE_STAMPS
    SD_WSTAMP(w2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTilesPerWave * kALoads) : "memory");
    SD_WSTAMP(w3);
    if (c + 1 < n_chunks) store_x((c + 1) & 1, c + 1);
    SD_WSTAMP(w4);
    __syncthreads();
    SD_WSTAMP(w5);
    w_acc[1] += w2 - w1b;
    w_acc[2] += w3 - w2;
    w_acc[3] += w4 - w3;
    w_acc[4] += w5 - w4;
#else
    if (c + 1 < n_chunks) store_x((c + 1) & 1, c + 1);
    __syncthreads();
#endif
  }
#ifdef SD_WIDE_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
    for (int i = 0; i < 6; ++i) sd_wide_stamps[wave][i] = w_acc[i];
    sd_wide_stamps[wave][6] = wall_clock64() - w_begin;
    sd_wide_stamps[wave][7] = clock64() - w_cyc0;  // core clock cycles of the loop
  }
#endif

  // acc[t][mt][reg] = sum of (tile row 4*(lane>>4)+reg of tile tile0+t, token 16*mt + lane&15)
  if (EPI == EPI_PARTIAL) {
    float* part = a.part + (size_t)blockIdx.y * a.M * NR;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t) {
      if (!tile_ok(t)) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        if (m < a.M)
          *reinterpret_cast<f32x4*>(part + (size_t)m * NR + (size_t)tile_of(t) * 16 + 4 * xq) = acc[t][mt];
      }
    }
  } else {
    constexpr bool PAIRED = EPI == SD_EPI_SILU_MUL || EPI == EPI_QKV_ROPE;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t) {
      const bool t_ok = tile_ok(t);  // wave-uniform
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        if (PAIRED) {  // rows r+8 sit 32 lanes up
#pragma unroll
          for (int j = 0; j < 4; ++j) hi[j] = __shfl(acc[t][mt][j], (lane + 32) & 63, 64);
        }
        if (t_ok && m < a.M && (!PAIRED || xq < 2))
          emit_unit<EPI == EPI_PARTIAL ? 0 : EPI>(a.epi, m, tile_of(t), 4 * xq, acc[t][mt], hi);
      }
    }
  }
}

// The epilogue of 4 consecutive tile rows r0..r0+3 of tile `tile` for token row m: lo holds
// their sums; in the paired modes hi holds the sums of rows r0+8..r0+11 (the up rows of a
// gated tile, the rotation partners of a rotary tile) and r0 is 0 or 4.
template <int EPI>
__device__ __forceinline__ void emit_unit(const WideEpiArgs& a, int m, int tile, int r0, f32x4 lo,
                                          f32x4 hi) {
  if (a.x_scale) {  // sums of products of fp8 activations: back to the row's scale first
    const float xs = a.x_scale[m];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] *= xs;
      hi[j] *= xs;
    }
  }
  if (EPI == EPI_QKV_ROPE) {
    const WideRope& rp = a.rope;
    const int D = rp.D, half = D / 2, tiles_per_head = D / 16;
    const int head = tile / tiles_per_head;
    const int i0 = (tile % tiles_per_head) * 8 + r0;
    const int b = m / rp.q_len, t = m - b * rp.q_len;
    const int pos = rp.seq_lens[b] + rp.pos_offset + t;
    if (pos < 0) return;
    const int page = head >= rp.Hq ? rp.block_tables[(size_t)b * rp.max_blocks + (pos >> 4)] : 0;
    const int slot = pos & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j;
      float l = lo[j], h = hi[j];
      if (a.w_scale) {
        l *= a.w_scale[head * D + i];
        h *= a.w_scale[head * D + i + half];
      }
      if (a.bias) {
        l += (float)a.bias[head * D + i];
        h += (float)a.bias[head * D + i + half];
      }
      l = round_bf16(l);
      h = round_bf16(h);
      if (head < rp.Hq + rp.Hkv && rp.cs) {
        const int r_pos = max(sd_rope_position(rp.mp, pos, b, m, i), 0);
        const float* cs = rp.cs + (size_t)min(r_pos, rp.max_pos - 1) * D;
        const float c = cs[i], sn = cs[half + i];
        const float nl = l * c - h * sn, nh = h * c + l * sn;
        l = nl;
        h = nh;
      }
      if (head < rp.Hq) {
        bf16_t* q = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + (size_t)head * D;
        q[i] = (bf16_t)l;
        q[i + half] = (bf16_t)h;
      } else if (head < rp.Hq + rp.Hkv) {
        bf16_t* kp = rp.kc + ((size_t)page * rp.Hkv + (head - rp.Hq)) * D * 16;
        const int i2 = i + half;
        kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = (bf16_t)l;
        kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = (bf16_t)h;
      } else {
        bf16_t* vp = rp.vc + ((size_t)page * rp.Hkv + (head - rp.Hq - rp.Hkv)) * D * 16;
        const int sl = sd_v_slot(slot);
        vp[i * 16 + sl] = (bf16_t)l;
        vp[(i + half) * 16 + sl] = (bf16_t)h;
      }
    }
  } else if (EPI == SD_EPI_SILU_MUL) {
    const int n0 = tile * 8 + r0;
    if (n0 + 3 < a.N && ((a.ldo | a.N) & 3) == 0) {  // four outputs in one store
      f32x4 g = lo, up = hi;
      if (a.w_scale) {
        const f32x4 sg = *reinterpret_cast<const f32x4*>(a.w_scale + n0);
        const f32x4 su = *reinterpret_cast<const f32x4*>(a.w_scale + a.N + n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[j] *= sg[j];
          up[j] *= su[j];
        }
      }
      if (a.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[j] += (float)a.bias[n0 + j];
          up[j] += (float)a.bias[a.N + n0 + j];
        }
      }
      bf16x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[j] = (bf16_t)(round_bf16(silu_w(round_bf16(g[j]))) * round_bf16(up[j]));
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n0) = r;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j;
      if (n >= a.N) break;
      float g = lo[j], up = hi[j];
      if (a.w_scale) {
        g *= a.w_scale[n];
        up *= a.w_scale[a.N + n];
      }
      if (a.bias) {
        g += (float)a.bias[n];
        up += (float)a.bias[a.N + n];
      }
      g = round_bf16(silu_w(round_bf16(g)));
      reinterpret_cast<bf16_t*>(a.out)[(size_t)m * a.ldo + n] = (bf16_t)(g * round_bf16(up));
    }
  } else {
    const int n0 = tile * 16 + r0;
    if (n0 + 3 < a.N && (a.ldo & 3) == 0) {  // four outputs in one store
      f32x4 y = lo;
      if (a.w_scale) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(a.w_scale + n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] *= sv[j];
      }
      if (a.bias) {
        const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] += (float)bv[j];
      }
      const size_t o = (size_t)m * a.ldo + n0;
      if (EPI == SD_EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + o) = y;
        return;
      }
      bf16x4 r;
      bf16x4 res = {0, 0, 0, 0};
      if (EPI == SD_EPI_RESIDUAL) res = *reinterpret_cast<const bf16x4*>(a.residual + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (EPI == SD_EPI_STORE) r[j] = (bf16_t)y[j];
        else if (EPI == SD_EPI_RESIDUAL) r[j] = (bf16_t)(round_bf16(y[j]) + (float)res[j]);
        else if (EPI == SD_EPI_RELU) r[j] = (bf16_t)fmaxf(round_bf16(y[j]), 0.f);
        else r[j] = (bf16_t)gelu_tanh_w(round_bf16(y[j]));
      }
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + o) = r;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j;
      if (n >= a.N) break;
      float y = lo[j];
      if (a.w_scale) y *= a.w_scale[n];
      if (a.bias) y += (float)a.bias[n];
      const size_t o = (size_t)m * a.ldo + n;
      if (EPI == SD_EPI_F32) {
        reinterpret_cast<float*>(a.out)[o] = y;
      } else {
        bf16_t* ob = reinterpret_cast<bf16_t*>(a.out);
        if (EPI == SD_EPI_STORE) ob[o] = (bf16_t)y;
        else if (EPI == SD_EPI_RESIDUAL) ob[o] = (bf16_t)(round_bf16(y) + (float)a.residual[o]);
        else if (EPI == SD_EPI_RELU) ob[o] = (bf16_t)fmaxf(round_bf16(y), 0.f);
        else ob[o] = (bf16_t)gelu_tanh_w(round_bf16(y));
      }
    }
  }
}

// ------------------------------------------------------------------ many rows: four waves
// Passes of 65..256 rows over matrices whose 8-tile groups cover all of K (K % 256 == 0).  In
// the eight-wave kernel above a chunk's loads are issued in one burst ahead of its MFMAs: 80 KiB
// per chunk and CU (48 KiB of activations that every workgroup re-reads from L2, 32 KiB of
// weights) go through the CU's one 64 B/clk vector memory path, the burst holds the issuing
// waves for about as long as the chunk's MFMAs take (in-kernel clocks: 0.5 us of 2.1 per chunk
// at 192 rows), and the LDS stores of the next chunk follow as a third phase (0.4 us).  Here a
// group runs on FOUR waves, one per SIMD, with two tiles per wave (a B fragment read from LDS
// feeds two MFMAs: half the LDS reads per CU), and everything else a chunk needs is spread
// through its MFMA stream at fixed places:
//   * weights are prefetched two chunks ahead into the registers they are consumed from: the
//     fragments of k-step j of chunk c + 2 are requested right after the MFMAs of k-step j of
//     chunk c (two register sets, the chunk loop unrolled by two: no copies);
//   * activations are loaded two chunks ahead and stored one chunk ahead: piece i of chunk c + 1
//     (in registers since chunk c - 1) is stored to the idle LDS buffer and piece i of chunk
//     c + 2 requested into the same registers, one piece every four B fragments;
//   * loads past the last chunk re-read the last chunk (the same lines, just fetched);
//   * the chunk loop is unrolled by two, so a split must hold an even number of chunks.
// The sums of a tile run over the same chunks in the same order as in the kernel above: same
// bits (experiments/bench_wide_shapes.py compares them).  Measured (192 rows, Llama-3-8B gate/up):
// 1.6 us per chunk against 2.1-2.7; what is left over the 0.95 us of its 96 MFMAs is the issue
// time of its 20 vector memory instructions, which a single wave per SIMD cannot hide.  Tried on
// the same structure and dropped: eight waves (slower: twice the LDS reads), four tiles per wave
// for 16-tile groups (every register; 5-10 % slower than the eight-wave kernel), three tiles per
// wave with the 2049..3072-tile matrices cut into one round of 9..12-tile groups (78 against
// 83 us at 192 rows: inside the spread between runs), the output
// staged in LDS and written in whole lines (no gain: the gated epilogue's time is its
// arithmetic), B rings of 8 and 12 fragments (no change).
template <int MT, int EPI, int TPW, bool W8, int WAVES>
__global__ __launch_bounds__(WAVES * SD_WAVE) void wide_gemm4_kernel(WideArgs a) {
  constexpr int kW = WAVES;
  constexpr int kXP = MT * 16 * 16 / (WAVES * SD_WAVE);        // 16-byte x pieces per thread and chunk
  constexpr int kXRows = WAVES * SD_WAVE / 16;                 // token rows one piece index covers
  static_assert(MT * 256 % (WAVES * SD_WAVE) == 0 && kChunkSteps * MT % kXP == 0, "MT must be even");
#ifdef SD_WIDE_STAMPS
  const unsigned long long w_kernel = wall_clock64();
#endif
  constexpr int kALoads = W8 ? kChunkSteps / 2 : kChunkSteps;  // 16-byte loads per tile and chunk
  constexpr int kRows = MT * 16;
  constexpr int kReads = kChunkSteps * MT;                     // B fragments per chunk
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = a.K / 32;
  const int NR = a.n_tiles * 16;
  const int ks_lo = blockIdx.y * a.ks_per_split;
  const int ks_hi = min(ks_lo + a.ks_per_split, KS);
  const int n_chunks = (ks_hi - ks_lo) / kChunkSteps;          // whole chunks: K % 128 == 0
  const int tile0 = blockIdx.x * a.group_tiles + wave;
  auto tile_of = [&](int t) { return tile0 + t * kW; };
  auto tile_ok = [&](int t) { return wave + t * kW < a.group_tiles && tile_of(t) < a.n_tiles; };

  // x: piece i of a thread is the 16 bytes at token row kXRows i + (tid >> 4), columns
  // 8 (tid & 15).. of the chunk; rows >= M re-read row M - 1 (their sums are never stored)
  const int xr = tid >> 4, xc8 = tid & 15;
  uint32_t x_off[kXP];
#pragma unroll
  for (int i = 0; i < kXP; ++i)
    x_off[i] = (uint32_t)(((size_t)min(kXRows * i + xr, a.M - 1) * a.ldx + xc8 * 8) * sizeof(bf16_t));
  const char* x_base = reinterpret_cast<const char*>(a.x) + (size_t)ks_lo * 32 * sizeof(bf16_t);
  bf16x8 x_stage[kXP];
  auto load_x = [&](int i, int chunk) {
    const char* base = x_base + (size_t)min(chunk, n_chunks - 1) * (kChunkCols * sizeof(bf16_t));
    x_stage[i] = *reinterpret_cast<const bf16x8*>(base + x_off[i]);
  };
  const uint32_t xs_store = (uint32_t)((xr * kXLD + xc8 * 8) * sizeof(bf16_t));
  auto store_x = [&](int i, int buf) {
    *reinterpret_cast<bf16x8*>(smem_raw + xs_store +
                               (size_t)(buf * kRows + kXRows * i) * kXLD * sizeof(bf16_t)) = x_stage[i];
  };

  // weights: fragment (t, j) of a chunk is one 16-byte lane load at a wave-uniform base
  const uint32_t w_off = lane * 16;
  const char* w_base[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = __builtin_amdgcn_readfirstlane(min(tile_of(t), a.n_tiles - 1));
    w_base[t] = reinterpret_cast<const char*>(a.W) +
                ((size_t)tile * (W8 ? KS / 2 : KS) + (W8 ? ks_lo / 2 : ks_lo)) * 1024;
  }
  bf16x8 a_set[2][TPW][kALoads];
  auto load_w = [&](int set, int t, int j, int chunk) {
    const char* base = w_base[t] + ((size_t)min(chunk, n_chunks - 1) * kALoads + j) * 1024;
    a_set[set][t][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + w_off));
  };

  f32x4 acc[TPW][MT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < kXP; ++i) load_x(i, 0);
#pragma unroll
  for (int set = 0; set < 2; ++set)
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int j = 0; j < kALoads; ++j) load_w(set, t, j, set);
#pragma unroll
  for (int i = 0; i < kXP; ++i) store_x(i, 0);
#pragma unroll
  for (int i = 0; i < kXP; ++i) load_x(i, 1);
  __syncthreads();

  const int xc = lane & 15, xq = lane >> 4;
  const uint32_t xs_read = (uint32_t)((xc * kXLD + 8 * xq) * sizeof(bf16_t));
#ifdef SD_WIDE_STAMPS
  unsigned long long w_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long w_begin = wall_clock64();
  const unsigned long long w_cyc0 = clock64();
  unsigned long long w_last = w_begin;
#endif
  // One chunk: reads LDS buffer P and register set P; stores chunk c + 1 to buffer P ^ 1.
  auto chunk_body = [&](int c, auto parity) {
    constexpr int P = decltype(parity)::value;
    const char* xb = smem_raw + xs_read + (size_t)P * kRows * kXLD * sizeof(bf16_t);
    auto read_b = [&](int idx) {
      return *reinterpret_cast<const bf16x8*>(xb + ((size_t)(idx % MT) * 16 * kXLD + (idx / MT) * 32) *
                                                        sizeof(bf16_t));
    };
    bf16x8 ring[kBRing];
#pragma unroll
    for (int i = 0; i < kBRing; ++i) ring[i] = read_b(i);
    bf16x8 a_k[TPW];
#pragma unroll
    for (int idx = 0; idx < kReads; ++idx) {
      const int j = idx / MT, mt = idx % MT;
      if (mt == 0) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          if constexpr (W8) {
            const u32x4 raw = __builtin_bit_cast(u32x4, a_set[P][t][j >> 1]);
            a_k[t] = fp8x8_to_bf16(raw[2 * (j & 1)], raw[2 * (j & 1) + 1]);
          } else {
            a_k[t] = a_set[P][t][j];
          }
        }
      }
      const bf16x8 b = ring[idx % kBRing];
#pragma unroll
      for (int t = 0; t < TPW; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_k[t], b, acc[t][mt], 0, 0, 0);
      if (idx + kBRing < kReads) ring[idx % kBRing] = read_b(idx + kBRing);
      if (idx % (kReads / kXP) == 1) {  // one piece every kReads / kXP fragments
        store_x(idx / (kReads / kXP), P ^ 1);
        load_x(idx / (kReads / kXP), c + 2);
      }
      if (mt == MT - 1 && (!W8 || (j & 1))) {  // the registers of this k-step's fragments are free
#pragma unroll
        for (int t = 0; t < TPW; ++t) load_w(P, t, W8 ? j >> 1 : j, c + 2);
      }
    }
    // the same order for the instruction scheduler
#pragma unroll
    for (int i = 0; i < kBRing; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int idx = 0; idx < kReads; ++idx) {
      __builtin_amdgcn_sched_group_barrier(0x008, TPW, 0);
      if (idx + kBRing < kReads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (idx % (kReads / kXP) == 1) {
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if (idx % MT == MT - 1 && (!W8 || ((idx / MT) & 1))) __builtin_amdgcn_sched_group_barrier(0x020, TPW, 0);
    }
#ifdef SD_WIDE_STAMPS
    SD_WSTAMP(w1);
    __syncthreads();
    SD_WSTAMP(w2);
    w_acc[1] += w1 - w_last;
    w_acc[4] += w2 - w1;
    w_last = w2;
#else
    __syncthreads();
#endif
  };
  for (int c = 0; c < n_chunks; c += 2) {  // an even number of chunks (the host checks)
    chunk_body(c, std::integral_constant<int, 0>{});
    chunk_body(c + 1, std::integral_constant<int, 1>{});
  }
#ifdef SD_WIDE_STAMPS
  const unsigned long long w_loop_end = wall_clock64();
  const unsigned long long w_cyc1 = clock64();
#endif

  // acc[t][mt][reg] = sum of (tile row 4*(lane>>4)+reg of tile tile0+4t, token 16*mt + lane&15)
  if (EPI == EPI_PARTIAL) {
    float* part = a.part + (size_t)blockIdx.y * a.M * NR;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (!tile_ok(t)) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        if (m < a.M)
          *reinterpret_cast<f32x4*>(part + (size_t)m * NR + (size_t)tile_of(t) * 16 + 4 * xq) = acc[t][mt];
      }
    }
  } else {
    constexpr bool PAIRED = EPI == SD_EPI_SILU_MUL || EPI == EPI_QKV_ROPE;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const bool t_ok = tile_ok(t);  // wave-uniform
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        if (PAIRED) {  // rows r+8 sit 32 lanes up
#pragma unroll
          for (int j = 0; j < 4; ++j) hi[j] = __shfl(acc[t][mt][j], (lane + 32) & 63, 64);
        }
        if (t_ok && m < a.M && (!PAIRED || xq < 2))
          emit_unit<EPI == EPI_PARTIAL ? 0 : EPI>(a.epi, m, tile_of(t), 4 * xq, acc[t][mt], hi);
      }
    }
  }
#ifdef SD_WIDE_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
    for (int i = 0; i < 6; ++i) sd_wide_stamps[wave][i] = w_acc[i];
    sd_wide_stamps[wave][0] = (w_begin - w_kernel) * (unsigned long long)n_chunks;      // prologue (per launch)
    sd_wide_stamps[wave][2] = (wall_clock64() - w_loop_end) * (unsigned long long)n_chunks;  // epilogue
    sd_wide_stamps[wave][6] = w_loop_end - w_begin;
    sd_wide_stamps[wave][7] = w_cyc1 - w_cyc0;
  }
#endif
}

// ------------------------------------------------------------------ fp8 activations (W8A8)
// The same workgroup shape over fp8 weight tiles, with the activations quantised to e4m3 on
// their way into LDS (per token row: x * (448 / max |x|), clamped, rounded to nearest even by
// v_cvt_pk_fp8_f32) and the products on the block-scaled fp8 MFMA with unit scales
// (v_mfma_scale_f32_16x16x128_f8f6f4: one instruction per tile, token tile and 128-column
// chunk).  A chunk of a row is 128 bytes of LDS instead of 256, a weight tile's chunk 2 KiB of
// HBM instead of 4: the kernel's two limits at 128..256 rows.  The epilogue multiplies a sum by
// the row's scale (max |x| / 448) and the weight row's scale.  A lane's 32 weight bytes of a
// chunk are columns 32 h + 8 q + j (h = 0..3, q = lane >> 4, j = 0..7: two 16-byte loads of the
// fp8 tiling); the LDS image of a row puts those 32 bytes side by side, so a B fragment is two
// 16-byte reads.  A row's arithmetic depends on that row alone.
typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr int kLD8 = kChunkCols + 16;  // bytes per LDS row of fp8 activations

template <int MT, int EPI, int TPW>
__global__ __launch_bounds__(kWideThreads) void wide_gemm_a8_kernel(WideArgs a) {
  constexpr int kTilesPerWave = TPW;
  // Weight chunks in flight per wave.  Measured on MI355X (192 rows, Llama-3-8B gate/up): 2 to 4
  // chunks ahead are no faster than 1 (73.4 vs 72.3 us), so the weight stream's round trip is
  // not what bounds the kernel; the ring stays in the code for shapes where it might be.
  constexpr int kDepth = 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned char* xs = reinterpret_cast<unsigned char*>(smem_raw);  // [2][MT*16][kLD8]
  constexpr int kRows = MT * 16;
  constexpr int kPieces = kRows * (kChunkCols / 8) / kWideThreads;  // 8-column pieces per thread
  static_assert(kRows * (kChunkCols / 8) % kWideThreads == 0, "MT must be even");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = a.K / 32;
  const int NR = a.n_tiles * 16;
  const int ks_lo = blockIdx.y * a.ks_per_split;
  const int ks_hi = min(ks_lo + a.ks_per_split, KS);
  const int n_chunks = (ks_hi - ks_lo + kChunkSteps - 1) / kChunkSteps;
  const int tile0 = blockIdx.x * a.group_tiles + wave;
  auto tile_of = [&](int t) { return tile0 + t * kWideWaves; };
  auto tile_ok = [&](int t) {
    return wave + t * kWideWaves < a.group_tiles && tile_of(t) < a.n_tiles;
  };

  // as in wide_gemm_kernel: every load unconditional on a clamped address
  bf16x8 x_stage[kPieces];
  float inv[kPieces];
#pragma unroll
  for (int i = 0; i < kPieces; ++i)
    inv[i] = a.x_inv[min((tid + i * kWideThreads) / (kChunkCols / 8), a.M - 1)];
  auto load_x = [&](int chunk) {
    const int col0 = (ks_lo + chunk * kChunkSteps) * 32;
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const int p = tid + i * kWideThreads;
      const int row = min(p / (kChunkCols / 8), a.M - 1);
      const int col = col0 + (p % (kChunkCols / 8)) * 8;
      x_stage[i] = *reinterpret_cast<const bf16x8*>(a.x + (size_t)row * a.ldx + min(col, a.K - 8));
    }
  };
  // keep: all ones for a chunk of the split, zero for the padding chunks of the last round
  auto store_x = [&](int buf, unsigned keep) {
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const int p = tid + i * kWideThreads;
      const int row = p / (kChunkCols / 8), c8 = p % (kChunkCols / 8);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        f[j] = fminf(fmaxf((float)x_stage[i][j] * inv[i], -448.f), 448.f);
      unsigned w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0u, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
      unsigned w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0u, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
      // columns 8 c8 .. + 7 of the chunk: lane quarter c8 & 3, 32-column block c8 >> 2
      // (K and the splits are whole chunks here: nothing to zero inside a chunk)
      *reinterpret_cast<uint2*>(xs + ((size_t)buf * kRows + row) * kLD8 + 32 * (c8 & 3) + 8 * (c8 >> 2)) =
          uint2{w0 & keep, w1 & keep};
    }
  };
  u32x4 a_q[kDepth][kTilesPerWave][2];
  auto load_a = [&](int chunk, u32x4 (&dst)[kTilesPerWave][2]) {
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int tile = min(tile_of(t), a.n_tiles - 1);
        const int step = min((ks_lo + chunk * kChunkSteps) / 2 + j, KS / 2 - 1);
        dst[t][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
            a.W + ((size_t)tile * (KS / 2) + step) * 512 + lane * 8));
      }
  };

  f32x4 acc[kTilesPerWave][MT];
#pragma unroll
  for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_x(0);
#pragma unroll
  for (int d = 0; d < kDepth; ++d) load_a(d, a_q[d]);
  store_x(0, ~0u);
  __syncthreads();
  const int xc = lane & 15, xq = lane >> 4;
  // Rounds of kDepth chunks (the ring slot of a chunk is a compile-time index); the last round
  // is filled up with chunks of zeroed activations.  Nothing in a chunk is conditional: the
  // loads' addresses are clamped, and without a branch the chunk stays one basic block, so
  // the order pinned below holds.
  const int n_rounds = (n_chunks + kDepth - 1) / kDepth;
  for (int round = 0; round < n_rounds; ++round) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      const int c = round * kDepth + d;
      i32x8 a_cur[kTilesPerWave];
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t) {
        const u32x4 lo = a_q[d][t][0], hi = a_q[d][t][1];
        a_cur[t] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3],
                         (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
      load_x(c + 1);
      __builtin_amdgcn_sched_barrier(0);
      load_a(c + kDepth, a_q[d]);
      const unsigned char* xb = xs + ((size_t)(c & 1) * kRows + xc) * kLD8 + 32 * xq;
      auto read_b = [&](int mt) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(xb + (size_t)mt * 16 * kLD8);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(xb + (size_t)mt * 16 * kLD8 + 16);
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3],
                     (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      constexpr int kRing = 2;   // B fragments in flight (8 registers each)
      i32x8 ring[kRing];
#pragma unroll
      for (int i = 0; i < kRing; ++i) ring[i] = read_b(i);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const i32x8 b = ring[mt % kRing];
#pragma unroll
        for (int t = 0; t < kTilesPerWave; ++t)
          acc[t][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              a_cur[t], b, acc[t][mt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if (mt + kRing < MT) ring[mt % kRing] = read_b(mt + kRing);
      }
      // pin that order (the scheduler otherwise hoists every fragment read to the top of the
      // chunk: MT x 8 registers): kRing fragment reads, then {MFMAs of a token tile, a read} ...
#pragma unroll
      for (int i = 0; i < kRing; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int i = 0; i < MT - kRing; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, kTilesPerWave, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < kRing; ++i) __builtin_amdgcn_sched_group_barrier(0x008, kTilesPerWave, 0);
      store_x((c + 1) & 1, c + 1 < n_chunks ? ~0u : 0u);
      __syncthreads();
    }
  }

  if (EPI == EPI_PARTIAL) {
    float* part = a.part + (size_t)blockIdx.y * a.M * NR;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t) {
      if (!tile_ok(t)) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        if (m < a.M)
          *reinterpret_cast<f32x4*>(part + (size_t)m * NR + (size_t)tile_of(t) * 16 + 4 * xq) = acc[t][mt];
      }
    }
  } else {
    constexpr bool PAIRED = EPI == SD_EPI_SILU_MUL || EPI == EPI_QKV_ROPE;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t) {
      const bool t_ok = tile_ok(t);  // wave-uniform
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + xc;
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        if (PAIRED) {  // rows r+8 sit 32 lanes up
#pragma unroll
          for (int j = 0; j < 4; ++j) hi[j] = __shfl(acc[t][mt][j], (lane + 32) & 63, 64);
        }
        if (t_ok && m < a.M && (!PAIRED || xq < 2))
          emit_unit<EPI == EPI_PARTIAL ? 0 : EPI>(a.epi, m, tile_of(t), 4 * xq, acc[t][mt], hi);
      }
    }
  }
}

// scales[m] = max |x[m]| / 448 (1 for a row of zeros), scales[M + m] = its inverse factor 448 / max |x[m]|
__global__ __launch_bounds__(256) void row_scales_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                             int M, int K, float* __restrict__ scales) {
  __shared__ float s_max[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  const bf16_t* row = x + (size_t)m * ldx;
  float mx = 0.f;
  for (int c = tid * 8; c < K; c += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)v[j]));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const bool ok = mx > 0.f && mx < INFINITY;
    scales[m] = ok ? mx / 448.f : 1.f;
    scales[M + m] = ok ? 448.f / mx : 1.f;
  }
}

// One thread = one unit (see emit_unit) of one token row.  Splits are added in index order.
template <int EPI>
__global__ __launch_bounds__(256) void wide_epilogue_kernel(WideEpiArgs a) {
  constexpr bool PAIRED = EPI == SD_EPI_SILU_MUL || EPI == EPI_QKV_ROPE;
  const int units_per_row = PAIRED ? a.NR / 8 : a.NR / 4;  // paired: 2 units per tile
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)a.M * units_per_row) return;
  const int m = (int)(gid / units_per_row), u = (int)(gid % units_per_row);
  const int tile = PAIRED ? u >> 1 : u >> 2, r0 = PAIRED ? (u & 1) * 4 : (u & 3) * 4;
  f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.S; ++s) {
    const float* p = a.part + ((size_t)s * a.M + m) * a.NR + tile * 16 + r0;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] += v[j];
    if (PAIRED) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(p + 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] += w[j];
    }
  }
  emit_unit<EPI>(a, m, tile, r0, lo, hi);
}

// K split: a function of the matrix alone (never of M), so results do not depend on the
// number of rows in the pass.
void plan_split(int n_tiles, int K, int& S, int& ks_per_split, int& group_tiles) {
  const int KS = K / 32;
  const bool small = n_tiles < kSmallTiles;
  group_tiles = n_tiles < kPairTiles ? kWideWaves : kTilesPerGroup;
  const int groups = (n_tiles + group_tiles - 1) / group_tiles;
  // one split (epilogue inside the kernel, no partial sums) once half the CUs have a
  // workgroup; otherwise enough splits for every CU
  int want = groups >= 128 ? 1 : 256 / groups;  // never more workgroups than CUs: one round
  const int min_steps = small ? 4 : 8;              // k-steps (32 columns) per split, at least
  const int max_s = (KS + min_steps - 1) / min_steps;
  want = want < 1 ? 1 : (want > max_s ? max_s : want);
  ks_per_split = ((KS + want - 1) / want + kChunkSteps - 1) / kChunkSteps * kChunkSteps;
  S = (KS + ks_per_split - 1) / ks_per_split;
}

int tiles_of(int N, int epi) { return epi == SD_EPI_SILU_MUL ? (N + 7) / 8 : (N + 15) / 16; }

size_t wide_ws_bytes(int M, int n_tiles, int K) {
  int S, ksp, g;
  plan_split(n_tiles, K, S, ksp, g);
  return (size_t)S * M * n_tiles * 16 * sizeof(float);
}

template <int MT, int EPI, int TPW, bool W8>
int launch_gemm_w(const WideArgs& a, int S, hipStream_t st) {
  const size_t lds = (size_t)2 * MT * 16 * kXLD * sizeof(bf16_t);
  // once per instantiation and process (host threads may race here)
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_gemm_kernel<MT, EPI, TPW, W8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess) {
    sd_set_error("sd_linear_wide: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    return SD_ERR_HIP;
  }
  const int groups = (a.n_tiles + a.group_tiles - 1) / a.group_tiles;
  hipLaunchKernelGGL((wide_gemm_kernel<MT, EPI, TPW, W8>), dim3(groups, S), dim3(kWideThreads), lds,
                     st, a);
  SD_CHECK_LAUNCH("sd_linear_wide");
  return SD_OK;
}

template <int MT, int EPI, int TPW>
int launch_gemm_a8(const WideArgs& a, int S, hipStream_t st) {
  if constexpr (EPI == SD_EPI_RELU || EPI == SD_EPI_GELU_TANH) {
    sd_set_error("sd_linear_wide_a8: epilogue %d not available with fp8 activations", EPI);
    return SD_ERR_UNSUPPORTED;
  } else {
    const size_t lds = (size_t)2 * MT * 16 * kLD8;
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_gemm_a8_kernel<MT, EPI, TPW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) {
      sd_set_error("sd_linear_wide_a8: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
      return SD_ERR_HIP;
    }
    const int groups = (a.n_tiles + a.group_tiles - 1) / a.group_tiles;
    hipLaunchKernelGGL((wide_gemm_a8_kernel<MT, EPI, TPW>), dim3(groups, S), dim3(kWideThreads), lds,
                       st, a);
    SD_CHECK_LAUNCH("sd_linear_wide_a8");
    return SD_OK;
  }
}

template <int MT, int EPI, int TPW>
int launch_gemm_t(const WideArgs& a, int S, hipStream_t st) {
#ifdef SD_WIDE_EXP  // experiments: a quick build of the few instantiations under study
  return launch_gemm_w<MT, EPI, TPW, false>(a, S, st);
#else
  return a.epi.w_scale ? launch_gemm_w<MT, EPI, TPW, true>(a, S, st)
                       : launch_gemm_w<MT, EPI, TPW, false>(a, S, st);
#endif
}

template <int MT, int EPI, int TPW, bool W8, int WAVES>
int launch_gemm4_w(const WideArgs& a, int S, hipStream_t st) {
  const size_t lds = (size_t)2 * MT * 16 * kXLD * sizeof(bf16_t);
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_gemm4_kernel<MT, EPI, TPW, W8, WAVES>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess) {
    sd_set_error("sd_linear_wide: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    return SD_ERR_HIP;
  }
  const int groups = (a.n_tiles + a.group_tiles - 1) / a.group_tiles;
  hipLaunchKernelGGL((wide_gemm4_kernel<MT, EPI, TPW, W8, WAVES>), dim3(groups, S), dim3(WAVES * SD_WAVE),
                     lds, st, a);
  SD_CHECK_LAUNCH("sd_linear_wide");
  return SD_OK;
}

template <int MT, int EPI, int TPW, int WAVES>
int launch_gemm4(const WideArgs& a, int S, hipStream_t st) {
#ifdef SD_WIDE_EXP
  return launch_gemm4_w<MT, EPI, TPW, false, WAVES>(a, S, st);
#else
  return a.epi.w_scale ? launch_gemm4_w<MT, EPI, TPW, true, WAVES>(a, S, st)
                       : launch_gemm4_w<MT, EPI, TPW, false, WAVES>(a, S, st);
#endif
}

#if defined(SD_WIDE_STAMPS) || defined(SD_WIDE_SWITCH)
#define SD_WIDE_HAS_SWITCH 1
int g_wide_four_waves = -1;  // experiments: -1 the library's choice, 0 never the four-wave kernel
#endif

template <int MT, int EPI>
int launch_gemm(const WideArgs& a, int S, hipStream_t st) {
  const bool pair = a.group_tiles != kWideWaves;  // 16-tile groups
#ifndef SD_WIDE_EXP
  if (a.x_inv) return pair ? launch_gemm_a8<MT, EPI, 2>(a, S, st) : launch_gemm_a8<MT, EPI, 1>(a, S, st);
#endif
  // The four-wave kernel where it was measured faster (same bits either way): many rows,
  // 8-tile groups that cover the whole of K with the epilogue inside (the matrices of
  // 1024..2047 tiles: a gated projection of 8192..16383 outputs, say), an even number of whole
  // chunks.  Split-K groups run a handful of chunks each and gain nothing; 16-tile groups
  // would need four tiles per wave and every register there is.
  constexpr bool kHasFour = MT >= kFourWaveMT && EPI != EPI_PARTIAL && EPI != SD_EPI_RELU &&
                            EPI != SD_EPI_GELU_TANH;
  if constexpr (kHasFour) {
    bool four = !pair && S == 1 && a.K % (2 * kChunkCols) == 0;
#ifdef SD_WIDE_HAS_SWITCH
    if (g_wide_four_waves >= 0) four = four && g_wide_four_waves != 0;
#endif
    if (four) return launch_gemm4<MT, EPI, 2, 4>(a, S, st);
    // 16-tile groups at 256 rows: the eight-wave kernel needs more registers than a wave of
    // eight has (it spills); the same tiles in 8-tile groups on four waves, in more rounds.
    if constexpr (MT == 16) {
      bool split_pairs = pair && S == 1 && a.K % (2 * kChunkCols) == 0;
#ifdef SD_WIDE_HAS_SWITCH
      if (g_wide_four_waves >= 0) split_pairs = split_pairs && g_wide_four_waves != 0;
#endif
      if (split_pairs) {
        WideArgs b = a;
        b.group_tiles = kWideWaves;
        return launch_gemm4<MT, EPI, 2, 4>(b, S, st);
      }
    }
  }
  return pair ? launch_gemm_t<MT, EPI, 2>(a, S, st) : launch_gemm_t<MT, EPI, 1>(a, S, st);
}

template <int EPI>
int launch_epi(const WideEpiArgs& e, hipStream_t st) {
  constexpr bool PAIRED = EPI == SD_EPI_SILU_MUL || EPI == EPI_QKV_ROPE;
  const size_t units = (size_t)e.M * (PAIRED ? e.NR / 8 : e.NR / 4);
  hipLaunchKernelGGL((wide_epilogue_kernel<EPI>), dim3((unsigned)((units + 255) / 256)), dim3(256),
                     0, st, e);
  SD_CHECK_LAUNCH("sd_linear_wide(epilogue)");
  return SD_OK;
}

int epilogue_by_kind(const WideEpiArgs& e, int epi, hipStream_t st) {
  switch (epi) {
    case SD_EPI_STORE: return launch_epi<SD_EPI_STORE>(e, st);
    case SD_EPI_RESIDUAL: return launch_epi<SD_EPI_RESIDUAL>(e, st);
    case SD_EPI_SILU_MUL: return launch_epi<SD_EPI_SILU_MUL>(e, st);
    case SD_EPI_RELU: return launch_epi<SD_EPI_RELU>(e, st);
    case SD_EPI_F32: return launch_epi<SD_EPI_F32>(e, st);
    case SD_EPI_GELU_TANH: return launch_epi<SD_EPI_GELU_TANH>(e, st);
    case EPI_QKV_ROPE: return launch_epi<EPI_QKV_ROPE>(e, st);
  }
  sd_set_error("sd_linear_wide: unknown epilogue %d", epi);
  return SD_ERR_INVALID;
}

template <int MT, bool DIRECT>
int gemm_by_kind(const WideArgs& a, int epi, int S, hipStream_t st) {
  if (!DIRECT) return launch_gemm<MT, EPI_PARTIAL>(a, S, st);
  switch (epi) {
    case SD_EPI_STORE: return launch_gemm<MT, SD_EPI_STORE>(a, S, st);
    case SD_EPI_RESIDUAL: return launch_gemm<MT, SD_EPI_RESIDUAL>(a, S, st);
    case SD_EPI_SILU_MUL: return launch_gemm<MT, SD_EPI_SILU_MUL>(a, S, st);
#ifndef SD_WIDE_EXP
    case SD_EPI_RELU: return launch_gemm<MT, SD_EPI_RELU>(a, S, st);
    case SD_EPI_F32: return launch_gemm<MT, SD_EPI_F32>(a, S, st);
    case SD_EPI_GELU_TANH: return launch_gemm<MT, SD_EPI_GELU_TANH>(a, S, st);
    case EPI_QKV_ROPE: return launch_gemm<MT, EPI_QKV_ROPE>(a, S, st);
#endif
  }
  sd_set_error("sd_linear_wide: unknown epilogue %d", epi);
  return SD_ERR_INVALID;
}

template <bool DIRECT>
int gemm_by_rows(const WideArgs& a, int mt, int epi, int S, hipStream_t st) {
#ifdef SD_WIDE_EXP
#ifndef SD_WIDE_EXP_MT
#define SD_WIDE_EXP_MT 12
#endif
  return gemm_by_kind<SD_WIDE_EXP_MT, DIRECT>(a, epi, S, st);
#else
  if (mt <= 2) return gemm_by_kind<2, DIRECT>(a, epi, S, st);
  if (mt <= 4) return gemm_by_kind<4, DIRECT>(a, epi, S, st);
  if (mt <= 8) return gemm_by_kind<8, DIRECT>(a, epi, S, st);
  if (mt <= 12) return gemm_by_kind<12, DIRECT>(a, epi, S, st);
  return gemm_by_kind<16, DIRECT>(a, epi, S, st);
#endif
}

int wide_common(const void* W, const void* x, int64_t ldx, int M, int K, int n_tiles, int epi,
                WideEpiArgs& e, void* workspace, size_t ws_bytes, hipStream_t st,
                const float* x_scales = nullptr) {
  SD_REQUIRE(W && x && e.out && workspace, SD_ERR_INVALID, "sd_linear_wide: null pointer");
  SD_REQUIRE(M >= 0 && M <= 256, SD_ERR_UNSUPPORTED, "sd_linear_wide: M=%d outside 0..256", M);
  SD_REQUIRE(K >= 32 && K % 32 == 0 && n_tiles >= 1, SD_ERR_INVALID,
             "sd_linear_wide: K must be a positive multiple of 32 (K=%d)", K);
  SD_REQUIRE(!e.w_scale || K % 64 == 0, SD_ERR_UNSUPPORTED,
             "sd_linear_wide: fp8 weights need K %% 64 == 0 (K=%d)", K);
  SD_REQUIRE(ldx % 8 == 0 && ldx >= K && sd_aligned16(W) && sd_aligned16(x) &&
                 sd_aligned16(workspace), SD_ERR_INVALID,
             "sd_linear_wide: rows must be 16-byte aligned (ldx=%lld)", (long long)ldx);
  if (M == 0) return SD_OK;
  int S, ksp, g;
  plan_split(n_tiles, K, S, ksp, g);
  const size_t need = (size_t)S * M * n_tiles * 16 * sizeof(float);
  SD_REQUIRE(ws_bytes >= need, SD_ERR_WORKSPACE, "sd_linear_wide: workspace %zu < %zu bytes",
             ws_bytes, need);
  if (x_scales) {  // [2][M]: the rows' scales, then their inverse factors (sd_row_scales_fp8)
    SD_REQUIRE(e.w_scale && K % 128 == 0, SD_ERR_UNSUPPORTED,
               "sd_linear_wide_a8: fp8 activations need fp8 weights and K %% 128 == 0 (K=%d)", K);
    SD_REQUIRE(epi != SD_EPI_RELU && epi != SD_EPI_GELU_TANH, SD_ERR_UNSUPPORTED,
               "sd_linear_wide_a8: epilogue %d not available with fp8 activations", epi);
    e.x_scale = x_scales;
  }
  WideArgs a{(const bf16_t*)W, (const bf16_t*)x, ldx, (float*)workspace, M, K, n_tiles, ksp, g, e,
             x_scales ? x_scales + M : nullptr};
  e.part = (const float*)workspace;
  e.S = S;
  e.M = M;
  e.NR = n_tiles * 16;
  a.epi = e;
  const int mt = (M + 15) / 16;
  if (S == 1) return gemm_by_rows<true>(a, mt, epi, S, st);
  const int rc = gemm_by_rows<false>(a, mt, epi, S, st);
  if (rc != SD_OK) return rc;
  return epilogue_by_kind(e, epi, st);
}

}  // namespace

extern "C" size_t sd_linear_wide_workspace_bytes(int M, int N, int K, int epilogue) {
  if (M < 1 || N < 1 || K < 32 || K % 32) return 0;
  return wide_ws_bytes(M, tiles_of(N, epilogue), K);
}

static int linear_wide_entry(const void* W, const float* w_scale, const float* x_scales, const void* x,
                             int64_t ldx, const void* bias, const void* residual, void* out,
                             int64_t ldo, int M, int N, int K, int epilogue, void* workspace,
                             size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(epilogue >= SD_EPI_STORE && epilogue <= SD_EPI_GELU_TANH, SD_ERR_INVALID,
             "sd_linear_wide: unknown epilogue %d", epilogue);
  SD_REQUIRE(N >= 1 && ldo >= N, SD_ERR_INVALID, "sd_linear_wide: need N >= 1 and ldo >= N");
  SD_REQUIRE(epilogue != SD_EPI_RESIDUAL || residual, SD_ERR_INVALID,
             "sd_linear_wide: residual epilogue without residual pointer");
  WideEpiArgs e{};
  e.N = N;
  e.w_scale = w_scale;
  e.bias = (const bf16_t*)bias;
  e.residual = (const bf16_t*)residual;
  e.out = out;
  e.ldo = ldo;
  return wide_common(W, x, ldx, M, K, tiles_of(N, epilogue), epilogue, e, workspace,
                     workspace_bytes, (hipStream_t)stream, x_scales);
}

extern "C" int sd_linear_wide(const void* W, const float* w_scale, const void* x, int64_t ldx,
                              const void* bias,
                              const void* residual, void* out, int64_t ldo, int M, int N, int K,
                              int epilogue, void* workspace, size_t workspace_bytes,
                              sd_stream_t stream) {
  return linear_wide_entry(W, w_scale, nullptr, x, ldx, bias, residual, out, ldo, M, N, K, epilogue,
                           workspace, workspace_bytes, stream);
}

extern "C" int sd_linear_wide_a8(const void* W, const float* w_scale, const float* x_scales,
                                 const void* x, int64_t ldx, const void* bias, const void* residual,
                                 void* out, int64_t ldo, int M, int N, int K, int epilogue,
                                 void* workspace, size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(w_scale && x_scales, SD_ERR_INVALID,
             "sd_linear_wide_a8: needs fp8 weights (w_scale) and the rows' scales (sd_row_scales_fp8)");
  return linear_wide_entry(W, w_scale, x_scales, x, ldx, bias, residual, out, ldo, M, N, K, epilogue,
                           workspace, workspace_bytes, stream);
}

extern "C" int sd_row_scales_fp8(const void* x, int64_t ldx, int M, int K, float* scales,
                                 sd_stream_t stream) {
  SD_REQUIRE(x && scales && M >= 0 && K >= 8 && K % 8 == 0 && ldx >= K && ldx % 8 == 0 &&
                 sd_aligned16(x), SD_ERR_INVALID, "sd_row_scales_fp8: bad arguments (M=%d K=%d)", M, K);
  if (M == 0) return SD_OK;
  hipLaunchKernelGGL(row_scales_fp8_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, M, K, scales);
  SD_CHECK_LAUNCH("sd_row_scales_fp8");
  return SD_OK;
}

static int qkv_rope_wide_entry(const void* W, const float* w_scale, const float* x_scales,
                               const void* x, int64_t ldx, const void* bias, void* q_out, int64_t ldq,
                               const float* rope_cs, int max_pos, void* k_cache, void* v_cache,
                               const int32_t* block_tables, int max_blocks_per_seq,
                               const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                               int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                               void* workspace, size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(q_out && k_cache && v_cache && block_tables && seq_lens, SD_ERR_INVALID,
             "sd_qkv_rope_wide: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && D % 16 == 0 && D >= 16,
             SD_ERR_INVALID, "sd_qkv_rope_wide: bad shape (D=%d)", D);
  SD_REQUIRE(!rope_cs || max_pos >= 1, SD_ERR_INVALID, "sd_qkv_rope_wide: max_pos < 1");
  SD_REQUIRE(ldq >= (int64_t)Hq * D, SD_ERR_INVALID, "sd_qkv_rope_wide: ldq too small");
  SD_REQUIRE(!rope_pos || (rope_pos->section_t >= 0 && rope_pos->section_h >= 0 &&
                           rope_pos->section_t + rope_pos->section_h <= D / 2),
             SD_ERR_INVALID, "sd_qkv_rope_wide: rotary sections exceed D/2");
  const int N = (Hq + 2 * Hkv) * D;
  WideEpiArgs e{};
  e.N = N;
  e.w_scale = w_scale;
  e.bias = (const bf16_t*)bias;
  e.out = q_out;
  e.ldo = ldq;
  e.rope = WideRope{rope_cs, max_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, block_tables,
                    max_blocks_per_seq, seq_lens, pos_offset, q_len, Hq, Hkv, D,
                    sd_make_rope_pos(rope_pos, B * q_len)};
  return wide_common(W, x, ldx, B * q_len, K, N / 16, EPI_QKV_ROPE, e, workspace, workspace_bytes,
                     (hipStream_t)stream, x_scales);
}

extern "C" int sd_qkv_rope_wide(const void* W, const float* w_scale, const void* x, int64_t ldx,
                                const void* bias,
                                void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                                void* k_cache, void* v_cache, const int32_t* block_tables,
                                int max_blocks_per_seq, const int32_t* seq_lens, int pos_offset,
                                int B, int q_len, int Hq, int Hkv, int D, int K,
                                const sd_rope_positions* rope_pos, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream) {
  return qkv_rope_wide_entry(W, w_scale, nullptr, x, ldx, bias, q_out, ldq, rope_cs, max_pos, k_cache,
                             v_cache, block_tables, max_blocks_per_seq, seq_lens, pos_offset, B, q_len,
                             Hq, Hkv, D, K, rope_pos, workspace, workspace_bytes, stream);
}

extern "C" int sd_qkv_rope_wide_a8(const void* W, const float* w_scale, const float* x_scales,
                                   const void* x, int64_t ldx, const void* bias, void* q_out,
                                   int64_t ldq, const float* rope_cs, int max_pos, void* k_cache,
                                   void* v_cache, const int32_t* block_tables, int max_blocks_per_seq,
                                   const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                                   int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                                   void* workspace, size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(w_scale && x_scales, SD_ERR_INVALID,
             "sd_qkv_rope_wide_a8: needs fp8 weights (w_scale) and the rows' scales (sd_row_scales_fp8)");
  return qkv_rope_wide_entry(W, w_scale, x_scales, x, ldx, bias, q_out, ldq, rope_cs, max_pos, k_cache,
                             v_cache, block_tables, max_blocks_per_seq, seq_lens, pos_offset, B, q_len,
                             Hq, Hkv, D, K, rope_pos, workspace, workspace_bytes, stream);
}

#ifdef SD_WIDE_HAS_SWITCH
extern "C" void sd_debug_wide_four_waves(int v) { g_wide_four_waves = v; }
#endif
#ifdef SD_WIDE_STAMPS
extern "C" int sd_debug_read_wide_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_wide_stamps), sizeof(sd_wide_stamps));
}
#endif
