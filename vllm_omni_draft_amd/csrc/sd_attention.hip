// Paged-KV attention for the verify pass (k+1 query tokens per sequence) and
// for decode (1 or 2 query tokens), plus rotary embedding + cache append.
//
// One workgroup = 4 waves = one (sequence, KV head, key split).  All query
// rows that share the KV head (q_len * Hq/Hkv <= 64) ride in the N dimension
// of 32x32x16 bf16 MFMAs, so the context K/V is streamed from HBM exactly once
// per (sequence, layer, KV head).  The page layouts make every MFMA operand
// fragment a contiguous 16-byte piece of a page: K and V go HBM -> VGPR ->
// MFMA with 512 B / 1 KiB contiguous segments per wave instruction.
//
//   S^T[key][m] = K[key][:] . Q[m][:]          A = K page rows, B = Q^T
//   O^T[d][m]  += V^T[d][key] * P^T[key][m]    A = V^T page rows, B = P^T
//
// With keys on the MFMA row index, the S^T accumulator of a lane holds 16 keys
// of ONE query row, so the online softmax is lane-local plus one exchange
// between the two half-waves, and the exponentiated accumulator registers are
// directly the B operand of the P.V MFMA (the V page stores its 16 tokens in
// the key order those registers have: see sd_v_slot).
//
// Every query row's arithmetic depends only on that row's q, its position and
// the cache contents: tiles are 32 keys at absolute positions, tile j of a
// split always goes to wave j%4, waves and splits are merged in index order.
// So a token scored as one of k+1 verify rows gets bit-identical output to the
// same token scored alone.
#include <mutex>

#include "sd_common.h"

namespace {

struct AttnArgs {
  const bf16_t* q;
  int64_t ldq;
  const bf16_t* kc;
  const bf16_t* vc;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  int pos_offset;
  bf16_t* out;
  int64_t ldo;
  int q_len, Hq, Hkv;
  // A sequence's q_total new tokens are attended in q_blocks blocks of q_len tokens (the last
  // one may be shorter); grid z = sequence * q_blocks + block.  Ordinary passes: q_total ==
  // q_len, q_blocks == 1.
  int q_total, q_blocks;
  float scale_log2;
  int split_tokens, n_splits;
  float* part_o;   // [B, Hkv, n_splits, MT*32, D]
  float* part_ml;  // [B, Hkv, n_splits, MT*32, 2]
  // Token trees (null: every new token sees all new tokens before it): bit j of
  // tree_mask[b * q_len + t] says that new token t of sequence b attends new token j (its
  // ancestors and itself; j <= t).  Keys before the first new token are seen by every row.
  const unsigned long long* tree_mask;
  // Read-ahead of weights by workgroups with blockIdx.y >= n_splits (pf_rows rows of them).
  const char* pf_base[2];
  unsigned pf_tile_bytes[2], pf_units[2], pf_half[2];
  int pf_rows;
};

// A read-ahead workgroup: touches 4 bytes of every 64 of its share of the units, so that they
// sit in its XCD's L2 (and in the memory-side cache) when the launch that streams them starts.
// Workgroups go round the XCDs in linear-id order and the grid's x extent is a multiple of 8
// here, so blockIdx.x % 8 is this workgroup's XCD; it takes the units whose consumer
// workgroup lands on the same XCD.  Nothing is written.
__device__ __forceinline__ void read_ahead(const AttnArgs& a) {
  const unsigned tid = threadIdx.x;
  const unsigned xcd = blockIdx.x & 7;
  const unsigned gx8 = gridDim.x >> 3;
  const unsigned rank = (blockIdx.x >> 3) + gx8 * ((blockIdx.y - a.n_splits) + a.pf_rows * blockIdx.z);
  const unsigned n_rank = gx8 * a.pf_rows * gridDim.z;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const char* base = a.pf_base[r];
    if (!base) continue;
    const unsigned tb = a.pf_tile_bytes[r], half = a.pf_half[r];
    const unsigned pieces = (half ? tb / 2 : tb) / 64;  // 64-byte pieces of a unit
    // this workgroup's units: those whose consumer sits on this XCD (u % 8 == xcd: the
    // consumer's grid is a multiple of 8), dealt round the XCD's read-ahead workgroups
    const unsigned n_units = a.pf_units[r];
    const unsigned first = xcd + 8 * rank;
    if (first >= n_units) continue;
    const unsigned mine = (n_units - first + 8 * n_rank - 1) / (8 * n_rank);
    const unsigned total = mine * pieces;
    for (unsigned i0 = tid; i0 < total; i0 += 256 * 8) {
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned i = i0 + j * 256;
        i = i < total ? i : i0;
        const unsigned k = i / pieces, q = i - k * pieces;
        const unsigned u = first + k * 8 * n_rank;
        const char* t = base + (size_t)(half ? u >> 1 : u) * tb;
        const size_t off = half ? (size_t)(q >> 1) * 256 + (u & 1) * 128 + (q & 1) * 64 : (size_t)q * 64;
        v[j] = *reinterpret_cast<const unsigned*>(t + off);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(v[j]));
    }
  }
}

#ifdef SD_ATTN_STAMPS
__device__ unsigned long long sd_stamps[16];
#define STAMP(i)                                                              \
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
  sd_stamps[i] = wall_clock64()
#else
#define STAMP(i)
#endif

// NBUF == 1 (one 32-row query tile) keeps the kernel within 256 registers, so that two workgroups share a CU and one's
// prologue and merge overlap the other's stream (large batches); deeper rings run one
// workgroup per CU.
// TREE: the pass scores token trees (AttnArgs::tree_mask); a variant of its own, so that the
// ordinary kernels carry nothing for it (a run-time test in the shared kernel cost 1.2 us
// per launch).
template <int D, int MT, int NBUF, int BT_PER_THREAD, bool RELOAD, bool TREE = false>
__global__ __launch_bounds__(256, NBUF == 1 && MT == 1 ? 2 : 1) void paged_attn_kernel(AttnArgs a) {
  if ((int)blockIdx.y >= a.n_splits) {
    read_ahead(a);
    return;
  }
  STAMP(0);
  constexpr int KS = D / 16;  // k-steps of the K.Q^T product
  constexpr int DT = D / 32;  // 32-row tiles of O^T
  constexpr int LDM = 33;     // padded m-stride of the LDS merge image
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm_o = reinterpret_cast<float*>(smem_raw);          // [4][MT][D][LDM]
  float* sm_m = sm_o + 4 * MT * D * LDM;                      // [4][MT][32]
  float* sm_l = sm_m + 4 * MT * 32;                           // [4][MT][32]
  float* sm_w = sm_l + 4 * MT * 32;                           // [MT*32][4] wave weights
  float* sm_mx = sm_w + MT * 32 * 4;                          // [MT*32] row maximum
  float* sm_lt = sm_mx + MT * 32;                             // [MT*32] row sum
  int32_t* sm_bt = reinterpret_cast<int32_t*>(sm_lt + MT * 32);  // block-table slice

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int kvh = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  const int sb = b / a.q_blocks, qb = b - sb * a.q_blocks;  // sequence, block of its new tokens
  const int q_len = min(a.q_len, a.q_total - qb * a.q_len);
  const int row0 = sb * a.q_total + qb * a.q_len;           // token row of the block's first token
  const int G = a.Hq / a.Hkv;
  const int m_rows = q_len * G;
  const int split_start = split * a.split_tokens;

  // Everything whose address does not depend on the sequence length is requested at
  // once: the split's slice of the block table (entries past the sequence's last page are
  // read but never used), the Q^T fragments, and the length itself.  A block-table load
  // issued later, beside the K/V stream, would sit behind it in the in-order load counter.
  const int32_t* bt = a.block_tables + (size_t)sb * a.max_blocks;
  const int page0 = split_start >> 4;
  const int n_bt = a.split_tokens / 16 + 8 * NBUF;
  int32_t bt_reg[BT_PER_THREAD];
#pragma unroll
  for (int i = 0; i < BT_PER_THREAD; ++i) {
    const int e = tid + i * 256;
    bt_reg[i] = e < n_bt ? bt[min(page0 + e, a.max_blocks - 1)] : 0;
  }
  // Q^T fragments (B operand): lane (r, h) holds Q[m = r][16s + 8h .. +8].
  bf16x8 qf[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + r;
    const bool valid = m < m_rows;
    const int t = valid ? m / G : 0, g = valid ? m % G : 0;
    const bf16_t* qrow = a.q + (size_t)(row0 + t) * a.ldq + (size_t)(kvh * G + g) * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) v = *reinterpret_cast<const bf16x8*>(qrow + 16 * s + 8 * h);
      qf[mt][s] = v;
    }
  }
  unsigned long long anc_m[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + r;
    anc_m[mt] = TREE && m < m_rows ? a.tree_mask[row0 + m / G] : ~0ull;
  }
  const int pos0 = a.seq_lens[sb] + a.pos_offset + qb * a.q_len;
  const int kv_len = pos0 + q_len;
  if (split_start >= kv_len) return;
  STAMP(1);
  const int split_end = min(split_start + a.split_tokens, kv_len);
  const int last_page = (kv_len - 1) >> 4;
  int kvlen_m[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + r;
    kvlen_m[mt] = m < m_rows ? pos0 + m / G + 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < BT_PER_THREAD; ++i) {
    const int e = tid + i * 256;
    if (e < n_bt) sm_bt[e] = bt_reg[i];
  }
  __syncthreads();

  STAMP(2);
  f32x16 o_acc[MT][DT];
  float m_run[MT], l_run[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    m_run[mt] = -INFINITY;
    l_run[mt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o_acc[mt][dt][i] = 0.f;
  }

  const size_t page_elems = (size_t)D * 16;
  // K fragments (A operand): lane (r, h) holds K[key0 + r][16s + 8h .. +8], which is
  // chunk 2s+h, token r&15 of page r>>4.  V^T fragments (A operand): lane (r, h) holds
  // row d = 32dt + r, slots 8h..8h+7 of page s.
  //
  // The wave keeps NBUF tiles in registers and reloads each buffer in place as soon as
  // its last MFMA has consumed it, so nearly all buffer registers are loads in flight:
  // the stream is bound by bytes in flight per CU, not by arithmetic.  Loads past the
  // split are redirected to one 16-byte address (cheap, never consumed), which keeps the
  // load sequence branch-free and the wait counts exact.
  bf16x8 kf[NBUF][KS], vf[NBUF][2][DT];
  const bf16x8* dummy = reinterpret_cast<const bf16x8*>(a.kc);
  // Page of a live tile: its first page always belongs to the sequence; a second page
  // past the last one holds only masked keys, so the first stands in for it.
  auto page_of = [&](int tile, int second) {
    const int pg = tile * 2 + (second && tile * 2 + 1 <= last_page ? 1 : 0);
    return sm_bt[pg - page0];
  };
  auto load_k = [&](bf16x8(&dst)[KS], int tile) {
    const bool in = tile * 32 < split_end;
    const int blk = in ? page_of(tile, r >> 4) : 0;
    const bf16_t* kpage = a.kc + ((size_t)blk * a.Hkv + kvh) * page_elems;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8* src =
          reinterpret_cast<const bf16x8*>(kpage + ((2 * s + h) * 16 + (r & 15)) * 8);
      dst[s] = __builtin_nontemporal_load(in ? src : dummy);
    }
  };
  auto load_v = [&](bf16x8(&dst)[2][DT], int tile) {
    const bool in = tile * 32 < split_end;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int blk = in ? page_of(tile, s) : 0;
      const bf16_t* vpage = a.vc + ((size_t)blk * a.Hkv + kvh) * page_elems;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(vpage + (dt * 32 + r) * 16 + 8 * h);
        dst[s][dt] = __builtin_nontemporal_load(in ? src : dummy);
      }
    }
  };
  const int tile0 = (split_start >> 5) + w;
#pragma unroll
  for (int u = 0; u < NBUF; ++u) {
    load_k(kf[u], tile0 + 4 * u);
    load_v(vf[u], tile0 + 4 * u);
  }
  STAMP(3);
  for (int base = tile0; base * 32 < split_end; base += 4 * NBUF) {
#pragma unroll
    for (int u = 0; u < NBUF; ++u) {
      const int tile = base + 4 * u;
      const int key0 = tile * 32;
      const bool live = key0 < split_end;  // wave-uniform; dead tiles only keep the ring turning
      // wave-uniform: a tile that holds new tokens of a token tree takes the ancestor mask too
      const bool tree_tile = TREE && key0 + 32 > pos0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8 pf[2];
        float alpha = 1.f;
        if (live) {
          f32x16 s_acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) s_acc[i] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s)
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[u][s], qf[mt][s], s_acc, 0, 0, 0);

          // Online softmax of row m = 32mt + r over this lane's 16 keys.
          float sv[16];
          float tmax = -INFINITY;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            bool seen = key < kvlen_m[mt];
            if (tree_tile) seen = seen && (key < pos0 || ((anc_m[mt] >> ((key - pos0) & 63)) & 1ull));
            sv[i] = seen ? s_acc[i] * a.scale_log2 : -INFINITY;
            tmax = fmaxf(tmax, sv[i]);
          }
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
          const float m_new = fmaxf(m_run[mt], tmax);
          const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
          alpha = __builtin_amdgcn_exp2f(m_run[mt] - m_use);
          float psum = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(sv[i] - m_use);
            psum += p;
            pf[i >> 3][i & 7] = (bf16_t)p;
          }
          l_run[mt] = l_run[mt] * alpha + psum;
          m_run[mt] = m_new;
        }
        if (RELOAD && mt == MT - 1) load_k(kf[u], tile + 4 * NBUF);
        if (live) {
          // The running maximum settles after the first tiles; a factor of exactly 1 in every
          // lane leaves the accumulators as they are, so the wave skips the rescale (same bits).
          if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
              for (int i = 0; i < 16; ++i) o_acc[mt][dt][i] *= alpha;
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
              o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u][s][dt], pf[s],
                                                                      o_acc[mt][dt], 0, 0, 0);
          }
        }
      }
      if (RELOAD) load_v(vf[u], tile + 4 * NBUF);
    }
  }

  STAMP(4);
  // Publish this wave's (m, l, O^T) to LDS; merge the 4 waves in index order.
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float l_tot = l_run[mt] + __shfl_xor(l_run[mt], 32, 64);
    if (h == 0) {
      sm_m[(w * MT + mt) * 32 + r] = m_run[mt];
      sm_l[(w * MT + mt) * 32 + r] = l_tot;
    }
    float* so = sm_o + (size_t)(w * MT + mt) * D * LDM;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = dt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        so[d * LDM + r] = o_acc[mt][dt][i];
      }
  }
  __syncthreads();

  STAMP(5);
  const bool direct = a.n_splits == 1;
  const size_t part_base = ((size_t)(b * a.Hkv + kvh) * a.n_splits + split) * (MT * 32);
  // Per query row: the four wave weights, the row maximum and the row sum.
  if (tid < MT * 32) {
    const int mt = tid >> 5, mr = tid & 31;
    float wgt[4];
    float mx = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      wgt[ww] = sm_m[(ww * MT + mt) * 32 + mr];
      mx = fmaxf(mx, wgt[ww]);
    }
    float l = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      wgt[ww] = (mx == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(wgt[ww] - mx);
      l += wgt[ww] * sm_l[(ww * MT + mt) * 32 + mr];
      sm_w[tid * 4 + ww] = wgt[ww];
    }
    sm_mx[tid] = mx;
    sm_lt[tid] = l;
    if (!direct && tid < m_rows) {
      a.part_ml[(part_base + tid) * 2 + 0] = mx;
      a.part_ml[(part_base + tid) * 2 + 1] = l;
    }
  }
  __syncthreads();
  // Live rows only; a thread keeps one d and walks rows 256/D apart.  The waves are added
  // in index order.
  constexpr int RP = 256 / D;
  const int d = tid % D;
#pragma unroll 4
  for (int m = tid / D; m < m_rows; m += RP) {
    const int mt = m >> 5, mr = m & 31;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(sm_w + m * 4);
    float o = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
      o += wv[ww] * sm_o[((size_t)(ww * MT + mt) * D + d) * LDM + mr];
    if (direct) {
      const float l = sm_lt[m];
      const int t = m / G, g = m % G;
      a.out[(size_t)(row0 + t) * a.ldo + (size_t)(kvh * G + g) * D + d] =
          (bf16_t)(l > 0.f ? o / l : 0.f);
    } else {
      a.part_o[(part_base + m) * D + d] = o;
    }
  }
  STAMP(6);
}

// Merges the per-split partials of one (sequence, token, head) in split order.  Up to
// kMergeAhead splits are requested before the sequence length is known (partials of splits
// past the length are read and ignored), so the launch costs one memory round trip, not
// three dependent ones; the arithmetic is the same sum in split order either way.
constexpr int kMergeAhead = 8;
template <int D>
__global__ __launch_bounds__(D) void attn_merge_kernel(AttnArgs a, int m_pad) {
  const int d = threadIdx.x;
  const int head = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
  const int sb = b / a.q_blocks, qb = b - sb * a.q_blocks;
  const int q_len = min(a.q_len, a.q_total - qb * a.q_len);
  if (t >= q_len) return;
  const int G = a.Hq / a.Hkv;
  const int kvh = head / G, g = head % G;
  const int m = t * G + g;
  const size_t base = (size_t)(b * a.Hkv + kvh) * a.n_splits;
  float pm[kMergeAhead], pl[kMergeAhead], po[kMergeAhead];
#pragma unroll
  for (int s = 0; s < kMergeAhead; ++s) {
    const size_t row = (base + min(s, a.n_splits - 1)) * m_pad + m;
    pm[s] = a.part_ml[row * 2];
    pl[s] = a.part_ml[row * 2 + 1];
    po[s] = a.part_o[row * D + d];
  }
  const int kv_len = a.seq_lens[sb] + a.pos_offset + qb * a.q_len + q_len;
  const int n_valid = min((kv_len + a.split_tokens - 1) / a.split_tokens, a.n_splits);
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < kMergeAhead; ++s)
    if (s < n_valid) mx = fmaxf(mx, pm[s]);
  for (int s = kMergeAhead; s < n_valid; ++s)
    mx = fmaxf(mx, a.part_ml[((base + s) * m_pad + m) * 2]);
  float o = 0.f, l = 0.f;
  if (mx != -INFINITY) {
#pragma unroll
    for (int s = 0; s < kMergeAhead; ++s) {
      if (s < n_valid) {
        const float f = __builtin_amdgcn_exp2f(pm[s] - mx);
        o += f * po[s];
        l += f * pl[s];
      }
    }
    for (int s = kMergeAhead; s < n_valid; ++s) {
      const size_t row = (base + s) * m_pad + m;
      const float f = __builtin_amdgcn_exp2f(a.part_ml[row * 2] - mx);
      o += f * a.part_o[row * D + d];
      l += f * a.part_ml[row * 2 + 1];
    }
  }
  a.out[(size_t)(sb * a.q_total + qb * a.q_len + t) * a.ldo + (size_t)head * D + d] =
      (bf16_t)(l > 0.f ? o / l : 0.f);
}

// Token trees: after acceptance the accepted rows' cache entries move to their chain slots.
// One thread owns one element of K and of V of a (layer, sequence, KV head) and performs the
// moves in path order, so a slot is read before a later move overwrites it.
__global__ __launch_bounds__(128) void kv_compact_kernel(
    bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int64_t layer_stride_elems,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ seq_lens,
    int pos_offset, const int32_t* __restrict__ path, const int32_t* __restrict__ n_path, int k,
    int Hkv, int D) {
  const int b = blockIdx.x, kvh = blockIdx.y, layer = blockIdx.z, d = threadIdx.x;
  if (d >= D) return;
  const int n = min(max(n_path[b], 0), k);
  const int pos0 = seq_lens[b] + pos_offset;
  const size_t page_elems = (size_t)D * 16;
  bf16_t* kl = kc + (size_t)layer * layer_stride_elems;
  bf16_t* vl = vc + (size_t)layer * layer_stride_elems;
  const int32_t* bt = block_tables + (size_t)b * max_blocks;
  for (int j = 0; j < n; ++j) {
    const int src = pos0 + path[(size_t)b * k + j], dst = pos0 + j + 1;
    if (src == dst) continue;
    const size_t sp = ((size_t)bt[src >> 4] * Hkv + kvh) * page_elems;
    const size_t dp = ((size_t)bt[dst >> 4] * Hkv + kvh) * page_elems;
    const int st = src & 15, dt = dst & 15;
    kl[dp + ((d >> 3) * 16 + dt) * 8 + (d & 7)] = kl[sp + ((d >> 3) * 16 + st) * 8 + (d & 7)];
    vl[dp + d * 16 + sd_v_slot(dt)] = vl[sp + d * 16 + sd_v_slot(st)];
  }
}

__global__ __launch_bounds__(256) void rope_append_kernel(
    bf16_t* __restrict__ qkv, const float* __restrict__ rope_cs, int max_pos,
    bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, const int32_t* __restrict__ block_tables,
    int max_blocks, const int32_t* __restrict__ seq_lens, int pos_offset, int q_len, int Hq,
    int Hkv, int D) {
  const int row = blockIdx.x;  // token row
  const int b = row / q_len, t = row % q_len;
  const int pos = seq_lens[b] + pos_offset + t;
  if (pos < 0) return;
  const int half = D / 2;
  const int page = block_tables[(size_t)b * max_blocks + (pos >> 4)];
  const int slot = pos & 15;
  bf16_t* x = qkv + (size_t)row * (Hq + 2 * Hkv) * D;
  const float* cs = rope_cs ? rope_cs + (size_t)min(pos, max_pos - 1) * D : nullptr;
  const int n_items = (Hq + 2 * Hkv) * half;
  for (int it = threadIdx.x; it < n_items; it += blockDim.x) {
    const int head = it / half, i = it % half;
    bf16_t* hp = x + (size_t)head * D;
    float lo = (float)hp[i], hi = (float)hp[i + half];
    if (head < Hq + Hkv && cs) {
      const float c = cs[i], s = cs[half + i];
      const float nlo = lo * c - hi * s;
      const float nhi = hi * c + lo * s;
      lo = nlo;
      hi = nhi;
    }
    if (head < Hq) {
      hp[i] = (bf16_t)lo;
      hp[i + half] = (bf16_t)hi;
    } else if (head < Hq + Hkv) {
      const int kh = head - Hq;
      bf16_t* kp = kc + ((size_t)page * Hkv + kh) * D * 16;
      kp[((i >> 3) * 16 + slot) * 8 + (i & 7)] = (bf16_t)lo;
      const int i2 = i + half;
      kp[((i2 >> 3) * 16 + slot) * 8 + (i2 & 7)] = (bf16_t)hi;
    } else {
      const int vh = head - Hq - Hkv;
      bf16_t* vp = vc + ((size_t)page * Hkv + vh) * D * 16;
      const int sl = sd_v_slot(slot);
      vp[i * 16 + sl] = (bf16_t)lo;
      vp[(i + half) * 16 + sl] = (bf16_t)hi;
    }
  }
}

template <int D, int MT, int NBUF>
size_t attn_lds_bytes(int split_tokens) {
  const size_t floats = (size_t)4 * MT * D * 33 + 2 * 4 * MT * 32 + MT * 32 * 6;
  return floats * sizeof(float) + (size_t)(split_tokens / 16 + 8 * NBUF) * sizeof(int32_t);
}

template <int D, int MT, int NBUF, int BTP, bool RELOAD, bool TREE = false>
int launch_attn_bt(const AttnArgs& a, int B, hipStream_t stream) {
  const size_t lds = attn_lds_bytes<D, MT, NBUF>(a.split_tokens);
  // raise the dynamic-LDS limit once per instantiation and process (host threads may race here)
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(paged_attn_kernel<D, MT, NBUF, BTP, RELOAD, TREE>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess) {
    sd_set_error("sd_paged_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    return SD_ERR_HIP;
  }
  hipLaunchKernelGGL((paged_attn_kernel<D, MT, NBUF, BTP, RELOAD, TREE>),
                     dim3(a.Hkv, a.n_splits + a.pf_rows, B), dim3(256), lds, stream, a);
  SD_CHECK_LAUNCH("sd_paged_attention");
  if (a.n_splits > 1 && a.out) {
    hipLaunchKernelGGL((attn_merge_kernel<D>), dim3(a.Hq, a.q_len, B), dim3(D), 0, stream, a,
                       MT * 32);
    SD_CHECK_LAUNCH("sd_paged_attention(merge)");
  }
  return SD_OK;
}

template <int D, int MT>
int launch_attn(const AttnArgs& a, int B, hipStream_t stream) {
  // Ring depth: as many 32-key tiles in flight per wave as the register file allows.
  constexpr int NBUF = D == 64 ? (MT == 1 ? 4 : 2) : (MT == 1 ? 3 : 1);
  // A split of at most NBUF tiles per wave is loaded whole, with no ring to keep turning.
  const int tiles_per_wave = a.split_tokens / 128;
  if (a.tree_mask) {  // token trees: the ring variant serves every split size
    if (a.split_tokens / 16 + 8 * NBUF <= 256) return launch_attn_bt<D, MT, NBUF, 1, true, true>(a, B, stream);
    return launch_attn_bt<D, MT, NBUF, 4, true, true>(a, B, stream);
  }
  if (tiles_per_wave == 1) return launch_attn_bt<D, MT, 1, 1, false>(a, B, stream);
  if constexpr (NBUF >= 2) {
    if (tiles_per_wave == 2) return launch_attn_bt<D, MT, 2, 1, false>(a, B, stream);
  }
  if constexpr (NBUF >= 3) {
    if (tiles_per_wave == 3) return launch_attn_bt<D, MT, 3, 1, false>(a, B, stream);
  }
  if constexpr (NBUF >= 4) {
    if (tiles_per_wave == 4) return launch_attn_bt<D, MT, 4, 1, false>(a, B, stream);
  }
  // Enough workgroups for two per CU: the one-tile ring, two workgroups per CU.
  if (MT == 1 && (long)B * a.Hkv * a.n_splits >= 512) {
    if (a.split_tokens / 16 + 8 <= 256) return launch_attn_bt<D, MT, 1, 1, true>(a, B, stream);
    return launch_attn_bt<D, MT, 1, 4, true>(a, B, stream);
  }
  // block-table entries staged per thread: 1 covers splits up to ~3.5k keys
  const int n_bt = a.split_tokens / 16 + 8 * NBUF;
  if (n_bt <= 256) return launch_attn_bt<D, MT, NBUF, 1, true>(a, B, stream);
  if (n_bt <= 1024) return launch_attn_bt<D, MT, NBUF, 4, true>(a, B, stream);
  sd_set_error("sd_paged_attention: split_tokens %d too large", a.split_tokens);
  return SD_ERR_UNSUPPORTED;
}

int n_splits_for(int max_ctx, int split_tokens) {
  return max(1, (max_ctx + split_tokens - 1) / split_tokens);
}

// read-ahead asked for the next attention launch of this host thread
thread_local sd_pf_range t_pf[2];
thread_local int t_pf_n = 0;

}  // namespace

void sd_attn_prefetch_next(const sd_pf_range* r, int n) {
  t_pf_n = n < 0 ? 0 : (n > 2 ? 2 : n);
  for (int i = 0; i < t_pf_n; ++i) t_pf[i] = r[i];
}

#ifdef SD_ATTN_STAMPS
extern "C" int sd_debug_read_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sd_stamps), sizeof(sd_stamps));
}
#endif

extern "C" size_t sd_paged_attention_workspace_bytes(int B, int q_len, int Hq, int D,
                                                     int max_ctx, int split_tokens) {
  (void)q_len;
  if (B <= 0 || split_tokens <= 0) return 0;
  const int n_splits = n_splits_for(max_ctx, split_tokens);
  if (n_splits == 1) return 16;
  // Rows are padded to 64 per KV head; Hq bounds Hkv.
  const size_t rows = (size_t)B * Hq * n_splits * 64;
  return rows * D * sizeof(float) + rows * 2 * sizeof(float) + 256;
}

extern "C" int sd_paged_attention(const void* q, int64_t ldq, const void* k_cache,
                                  const void* v_cache, const int32_t* block_tables,
                                  int max_blocks_per_seq, const int32_t* seq_lens,
                                  int pos_offset, void* out, int64_t ldo, int B, int q_len,
                                  int Hq, int Hkv, int D, float scale, int max_ctx,
                                  int split_tokens, void* workspace, size_t workspace_bytes,
                                  sd_stream_t stream) {
  return sd_paged_attention_blocked(q, ldq, k_cache, v_cache, block_tables, max_blocks_per_seq,
                                    seq_lens, pos_offset, out, ldo, B, q_len, q_len, Hq, Hkv, D,
                                    scale, max_ctx, split_tokens, workspace, workspace_bytes, stream);
}

static int attention_launch(const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                            const int32_t* block_tables, int max_blocks_per_seq,
                            const int32_t* seq_lens, int pos_offset, void* out, int64_t ldo, int B,
                            int q_total, int q_len, int Hq, int Hkv, int D, float scale,
                            int max_ctx, int split_tokens, void* workspace,
                            size_t workspace_bytes, const uint64_t* tree_mask, sd_stream_t stream);

extern "C" int sd_paged_attention_blocked(const void* q, int64_t ldq, const void* k_cache,
                                          const void* v_cache, const int32_t* block_tables,
                                          int max_blocks_per_seq, const int32_t* seq_lens,
                                          int pos_offset, void* out, int64_t ldo, int B,
                                          int q_total, int q_len, int Hq, int Hkv, int D,
                                          float scale, int max_ctx, int split_tokens,
                                          void* workspace, size_t workspace_bytes,
                                          sd_stream_t stream) {
  return attention_launch(q, ldq, k_cache, v_cache, block_tables, max_blocks_per_seq, seq_lens,
                          pos_offset, out, ldo, B, q_total, q_len, Hq, Hkv, D, scale, max_ctx,
                          split_tokens, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int sd_paged_attention_tree(const void* q, int64_t ldq, const void* k_cache,
                                       const void* v_cache, const int32_t* block_tables,
                                       int max_blocks_per_seq, const int32_t* seq_lens,
                                       int pos_offset, const uint64_t* tree_mask, void* out,
                                       int64_t ldo, int B, int q_len, int Hq, int Hkv, int D,
                                       float scale, int max_ctx, int split_tokens, void* workspace,
                                       size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(tree_mask, SD_ERR_INVALID, "sd_paged_attention_tree: tree_mask is NULL");
  SD_REQUIRE(q_len >= 1 && q_len <= 64, SD_ERR_UNSUPPORTED,
             "sd_paged_attention_tree: %d new tokens per sequence outside 1..64", q_len);
  return attention_launch(q, ldq, k_cache, v_cache, block_tables, max_blocks_per_seq, seq_lens,
                          pos_offset, out, ldo, B, q_len, q_len, Hq, Hkv, D, scale, max_ctx,
                          split_tokens, workspace, workspace_bytes, tree_mask, stream);
}

static int attention_launch(const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                            const int32_t* block_tables, int max_blocks_per_seq,
                            const int32_t* seq_lens, int pos_offset, void* out, int64_t ldo, int B,
                            int q_total, int q_len, int Hq, int Hkv, int D, float scale,
                            int max_ctx, int split_tokens, void* workspace,
                            size_t workspace_bytes, const uint64_t* tree_mask, sd_stream_t stream) {
  SD_REQUIRE(q_total >= q_len && q_len >= 1, SD_ERR_INVALID,
             "sd_paged_attention_blocked: need 1 <= q_block (%d) <= q_total (%d)", q_len, q_total);
  const int q_blocks = (q_total + q_len - 1) / q_len;
  SD_REQUIRE(out || q_blocks == 1, SD_ERR_INVALID,
             "sd_paged_attention_blocked: partials-only output needs one block per sequence");
  SD_REQUIRE((long)B * q_blocks <= 65535, SD_ERR_UNSUPPORTED,
             "sd_paged_attention_blocked: %ld blocks exceed the grid", (long)B * q_blocks);
  SD_REQUIRE(q && k_cache && v_cache && block_tables && seq_lens, SD_ERR_INVALID,
             "sd_paged_attention: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0, SD_ERR_INVALID,
             "sd_paged_attention: bad head counts (Hq=%d Hkv=%d)", Hq, Hkv);
  SD_REQUIRE(D == 64 || D == 128, SD_ERR_UNSUPPORTED,
             "sd_paged_attention: head_dim %d not in {64,128}", D);
  const int G = Hq / Hkv;
  SD_REQUIRE(q_len * G <= 64, SD_ERR_UNSUPPORTED,
             "sd_paged_attention: q_len*(Hq/Hkv) = %d > 64", q_len * G);
  SD_REQUIRE(split_tokens >= 128 && split_tokens % 128 == 0 && split_tokens <= 15872,
             SD_ERR_INVALID,
             "sd_paged_attention: split_tokens %d must be a multiple of 128 in 128..15872",
             split_tokens);
  SD_REQUIRE(max_ctx >= 1 && max_blocks_per_seq * 16 >= max_ctx, SD_ERR_INVALID,
             "sd_paged_attention: max_ctx %d exceeds block table (%d pages)", max_ctx,
             max_blocks_per_seq);
  SD_REQUIRE(ldq % 8 == 0 && sd_aligned16(q) && (!out || ldo >= (int64_t)Hq * D), SD_ERR_INVALID,
             "sd_paged_attention: q rows must be 16-byte aligned");
  if (B == 0) return SD_OK;
  AttnArgs a;
  a.q = (const bf16_t*)q;
  a.ldq = ldq;
  a.kc = (const bf16_t*)k_cache;
  a.vc = (const bf16_t*)v_cache;
  a.block_tables = block_tables;
  a.max_blocks = max_blocks_per_seq;
  a.seq_lens = seq_lens;
  a.pos_offset = pos_offset;
  a.out = (bf16_t*)out;
  a.ldo = ldo;
  a.q_len = q_len;
  a.q_total = q_total;
  a.q_blocks = q_blocks;
  a.Hq = Hq;
  a.Hkv = Hkv;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.split_tokens = split_tokens;
  a.n_splits = n_splits_for(max_ctx, split_tokens);
  a.part_o = nullptr;
  a.part_ml = nullptr;
  a.tree_mask = (const unsigned long long*)tree_mask;
  a.pf_rows = 0;
  for (int r = 0; r < 2; ++r) {
    a.pf_base[r] = nullptr;
    a.pf_tile_bytes[r] = a.pf_units[r] = a.pf_half[r] = 0;
  }
  const int n_pf = t_pf_n;
  t_pf_n = 0;
  // Read-ahead workgroups fill the launch up to one workgroup per compute unit; they need the
  // grid's x extent (Hkv) to be a multiple of 8 to know their XCD.
  const long attn_wgs = (long)Hkv * a.n_splits * B * q_blocks;
  if (n_pf > 0 && Hkv % 8 == 0 && attn_wgs <= 192) {
    int used = 0;
    for (int r = 0; r < n_pf; ++r) {
      const sd_pf_range& f = t_pf[r];
      if (!f.base || f.n_units == 0 || f.tile_bytes % 128 != 0 || f.consumer_grid % 8 != 0) continue;
      a.pf_base[used] = (const char*)f.base;
      a.pf_tile_bytes[used] = f.tile_bytes;
      a.pf_units[used] = f.n_units;
      a.pf_half[used] = f.half;
      ++used;
    }
    if (used) a.pf_rows = (int)((256 - attn_wgs) / ((long)Hkv * B * q_blocks));
  }
  SD_REQUIRE(out || a.n_splits > 1, SD_ERR_INVALID,
             "sd_paged_attention: out == NULL (partials only) needs more than one split");
  const int MT = q_len * G <= 32 ? 1 : 2;
  if (a.n_splits > 1) {
    const size_t rows = (size_t)B * q_blocks * Hkv * a.n_splits * (MT * 32);
    const size_t need = rows * D * sizeof(float) + rows * 2 * sizeof(float);
    SD_REQUIRE(workspace && workspace_bytes >= need, SD_ERR_WORKSPACE,
               "sd_paged_attention: workspace %zu < %zu bytes", workspace_bytes, need);
    a.part_o = (float*)workspace;
    a.part_ml = a.part_o + rows * D;
  }
  hipStream_t st = (hipStream_t)stream;
  const int Z = B * q_blocks;  // workgroup columns: one per (sequence, block)
  if (D == 128) return MT == 1 ? launch_attn<128, 1>(a, Z, st) : launch_attn<128, 2>(a, Z, st);
  return MT == 1 ? launch_attn<64, 1>(a, Z, st) : launch_attn<64, 2>(a, Z, st);
}

// ------------------------------------------------------------------ allocation-free form
// The split partials of sd_paged_attention_static live in this array instead of a caller's
// workspace.  Every GPU the library runs on holds its own copy until the process ends.  In
// host code the name is only a handle for hipGetSymbolAddress: the device address it yields
// (one per device) is what kernels and memory calls are given.
static __device__ __align__(256) unsigned char scratch[SD_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch`, per device

// The array's address on the device that `stream` belongs to (the null stream: the current
// device), looked up once per device with that device current.
int static_scratch(hipStream_t stream, void** out) {
  int cur = -1;
  hipError_t e = hipGetDevice(&cur);
  SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_paged_attention_static: hipGetDevice: %s",
             hipGetErrorString(e));
  int dev = cur;
  if (stream) {
    hipDevice_t sdev = 0;
    e = hipStreamGetDevice(stream, &sdev);
    SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_paged_attention_static: hipStreamGetDevice: %s",
               hipGetErrorString(e));
    dev = (int)sdev;
  }
  SD_REQUIRE(dev >= 0 && dev < kMaxDevices, SD_ERR_HIP,
             "sd_paged_attention_static: device %d outside 0..%d", dev, kMaxDevices - 1);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {  // first call for this device
    if (dev != cur) {
      e = hipSetDevice(dev);
      SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_paged_attention_static: hipSetDevice(%d): %s", dev,
                 hipGetErrorString(e));
    }
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (dev != cur) (void)hipSetDevice(cur);
    SD_REQUIRE(e == hipSuccess && p, SD_ERR_HIP,
               "sd_paged_attention_static: hipGetSymbolAddress(scratch) on device %d: %s", dev,
               hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return SD_OK;
}
}  // namespace

extern "C" int sd_paged_attention_static(const void* q, int64_t ldq, const void* k_cache,
                                         const void* v_cache, const int32_t* block_tables,
                                         int max_blocks_per_seq, const int32_t* seq_lens,
                                         int pos_offset, void* out, int64_t ldo, int B, int q_len,
                                         int Hq, int Hkv, int D, float scale, int max_ctx,
                                         int split_tokens, sd_stream_t stream) {
  SD_REQUIRE(out, SD_ERR_INVALID,
             "sd_paged_attention_static: out must not be NULL (the partials stay inside the library)");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && max_ctx >= 1 &&
                 split_tokens >= 128 && (D == 64 || D == 128),
             SD_ERR_INVALID, "sd_paged_attention_static: bad shape");
  const int n_splits = n_splits_for(max_ctx, split_tokens);
  if (n_splits > 1) {
    const size_t m_pad = q_len * (Hq / Hkv) <= 32 ? 32 : 64;
    const size_t need = (size_t)B * Hkv * n_splits * m_pad * (D + 2) * sizeof(float);
    SD_REQUIRE(need <= SD_STATIC_SCRATCH_BYTES, SD_ERR_WORKSPACE,
               "sd_paged_attention_static: %zu bytes of split partials (B=%d Hkv=%d splits=%d) exceed "
               "the built-in scratch of %zu bytes: use sd_paged_attention with a workspace",
               need, B, Hkv, n_splits, (size_t)SD_STATIC_SCRATCH_BYTES);
  }
  void* ws = nullptr;
  const int rc = static_scratch((hipStream_t)stream, &ws);
  if (rc != SD_OK) return rc;
  // Nothing in the scratch must start at zero: every partial a merge reads was written by
  // the same call's attention launch (splits past a sequence's length are never read).
  return sd_paged_attention_blocked(q, ldq, k_cache, v_cache, block_tables, max_blocks_per_seq,
                                    seq_lens, pos_offset, out, ldo, B, q_len, q_len, Hq, Hkv, D, scale,
                                    max_ctx, split_tokens, ws, SD_STATIC_SCRATCH_BYTES, stream);
}

extern "C" int sd_kv_compact(void* k_cache, void* v_cache, int64_t layer_stride_bytes, int n_layers,
                             const int32_t* block_tables, int max_blocks_per_seq,
                             const int32_t* seq_lens, int pos_offset, const int32_t* path,
                             const int32_t* n_path, int k, int B, int Hkv, int D,
                             sd_stream_t stream) {
  SD_REQUIRE(k_cache && v_cache && block_tables && seq_lens && path && n_path, SD_ERR_INVALID,
             "sd_kv_compact: null pointer");
  SD_REQUIRE(B >= 0 && n_layers >= 1 && n_layers <= 65535 && Hkv >= 1 && Hkv <= 65535 && k >= 1 &&
                 (D == 64 || D == 128) && layer_stride_bytes % 2 == 0,
             SD_ERR_INVALID, "sd_kv_compact: bad shape (layers=%d Hkv=%d D=%d k=%d)", n_layers, Hkv, D, k);
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(kv_compact_kernel, dim3(B, Hkv, n_layers), dim3(128), 0, (hipStream_t)stream,
                     (bf16_t*)k_cache, (bf16_t*)v_cache, layer_stride_bytes / 2, block_tables,
                     max_blocks_per_seq, seq_lens, pos_offset, path, n_path, k, Hkv, D);
  SD_CHECK_LAUNCH("sd_kv_compact");
  return SD_OK;
}

extern "C" int sd_rope_append(void* qkv, const float* rope_cs, int max_pos, void* k_cache,
                              void* v_cache, const int32_t* block_tables,
                              int max_blocks_per_seq, const int32_t* seq_lens, int pos_offset,
                              int B, int q_len, int Hq, int Hkv, int D, sd_stream_t stream) {
  SD_REQUIRE(qkv && k_cache && v_cache && block_tables && seq_lens, SD_ERR_INVALID,
             "sd_rope_append: null pointer");
  SD_REQUIRE(B >= 0 && q_len >= 1 && Hq >= 1 && Hkv >= 1 && D % 16 == 0 && D >= 16,
             SD_ERR_INVALID, "sd_rope_append: bad shape (D=%d)", D);
  SD_REQUIRE(!rope_cs || max_pos >= 1, SD_ERR_INVALID, "sd_rope_append: max_pos < 1");
  if (B == 0) return SD_OK;
  hipLaunchKernelGGL(rope_append_kernel, dim3(B * q_len), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)qkv, rope_cs, max_pos, (bf16_t*)k_cache, (bf16_t*)v_cache,
                     block_tables, max_blocks_per_seq, seq_lens, pos_offset, q_len, Hq, Hkv, D);
  SD_CHECK_LAUNCH("sd_rope_append");
  return SD_OK;
}
