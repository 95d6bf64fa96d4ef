// Host runtime of the library: error text, device info, the decoder-model
// forward pass (a fixed sequence of kernel launches on the caller's stream,
// no synchronisation, no allocation) and hipGraph capture helpers so that a
// whole speculative step replays as one launch.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "sd_common.h"
#include <rccl/rccl.h>

namespace {
thread_local char g_err[512] = "";
}

void sd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sd_version(void) { return 100; }

extern "C" int sd_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return SD_ERR_INVALID;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = '\0';
  return SD_OK;
}

extern "C" int sd_device_info(int* cu_count, char* name, size_t n) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    sd_set_error("sd_device_info: hipGetDevice: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    sd_set_error("sd_device_info: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (name && n) {
    strncpy(name, prop.gcnArchName, n - 1);
    name[n - 1] = '\0';
  }
  return SD_OK;
}

struct sd_model {
  sd_model_desc d;
  std::vector<sd_layer_weights> layers;
};

extern "C" sd_model* sd_model_create(const sd_model_desc* desc) {
  if (!desc || !desc->layers || desc->n_layers < 1) {
    sd_set_error("sd_model_create: null descriptor or no layers");
    return nullptr;
  }
  const sd_model_desc& d = *desc;
  if (d.arch != SD_ARCH_LLAMA && d.arch != SD_ARCH_OPT) {
    sd_set_error("sd_model_create: unknown arch %d", d.arch);
    return nullptr;
  }
  if (d.hidden % 32 || d.ffn % 32 || d.vocab % 16 || (d.head_dim != 64 && d.head_dim != 128) ||
      d.n_heads % d.n_kv_heads || d.vocab < 1 || !d.embed || !d.lm_head || !d.final_norm_w) {
    sd_set_error("sd_model_create: unsupported shape (H=%d F=%d D=%d Hq=%d Hkv=%d)", d.hidden,
                 d.ffn, d.head_dim, d.n_heads, d.n_kv_heads);
    return nullptr;
  }
  if (d.weight_fp8 ? (!d.lm_head_s || d.hidden % 64 || d.ffn % 64 ||
                      (d.n_heads * d.head_dim) % 64)
                   : d.lm_head_s != nullptr) {
    sd_set_error("sd_model_create: fp8 weights need an LM head scale and H, F, Hq*D multiples of 64");
    return nullptr;
  }
  if (d.lm_vocab < 0 || d.lm_vocab % 16 || d.lm_vocab > d.vocab) {
    sd_set_error("sd_model_create: lm_vocab %d must be 0 or a multiple of 16 up to vocab", d.lm_vocab);
    return nullptr;
  }
  if (d.mrope_section_t < 0 || d.mrope_section_h < 0 ||
      d.mrope_section_t + d.mrope_section_h > d.head_dim / 2 ||
      ((d.mrope_section_t || d.mrope_section_h) && d.arch != SD_ARCH_LLAMA)) {
    sd_set_error("sd_model_create: rotary sections %d, %d need a rotary model and t + h <= D/2",
                 d.mrope_section_t, d.mrope_section_h);
    return nullptr;
  }
  if (d.arch == SD_ARCH_LLAMA && !d.rope_cs) {
    sd_set_error("sd_model_create: llama arch needs a rope table");
    return nullptr;
  }
  if (d.arch == SD_ARCH_OPT && !d.pos_table) {
    sd_set_error("sd_model_create: opt arch needs a position table");
    return nullptr;
  }
  sd_model* m = new sd_model;
  m->d = d;
  m->layers.assign(d.layers, d.layers + d.n_layers);
  m->d.layers = m->layers.data();
  for (int i = 0; i < d.n_layers; ++i) {
    const sd_layer_weights& l = m->layers[i];
    if (!l.attn_norm_w || !l.qkv_w || !l.o_w || !l.mlp_norm_w || !l.up_w || !l.down_w) {
      sd_set_error("sd_model_create: layer %d has a null weight", i);
      delete m;
      return nullptr;
    }
    const bool scaled = l.qkv_s && l.o_s && l.up_s && l.down_s;
    const bool plain = !l.qkv_s && !l.o_s && !l.up_s && !l.down_s;
    if (d.weight_fp8 ? !scaled : !plain) {
      sd_set_error("sd_model_create: layer %d: weight scales must be given for fp8 weights and "
                   "only for them", i);
      delete m;
      return nullptr;
    }
  }
  return m;
}

extern "C" void sd_model_destroy(sd_model* m) { delete m; }

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
  size_t h, xn, qkv, attn, act, attn_ws, attn_ws_bytes, wide_ws, wide_ws_bytes, slabs, acc, x_scales, total;
};

// wide: the pass uses the 17..256-row linear kernels (and their partial-sum scratch).
Workspace plan_workspace(const sd_model_desc& d, int rows, int max_ctx, int split_tokens,
                         bool wide) {
  Workspace w;
  const size_t qkv_cols = (size_t)(d.n_heads + 2 * d.n_kv_heads) * d.head_dim;
  size_t off = 0;
  w.h = off;    off += align256((size_t)rows * d.hidden * 2);
  w.xn = off;   off += align256((size_t)rows * d.hidden * 2);
  w.qkv = off;  off += align256((size_t)rows * qkv_cols * 2);
  w.attn = off; off += align256((size_t)rows * d.n_heads * d.head_dim * 2);
  w.act = off;  off += align256((size_t)rows * d.ffn * 2);
  // rows bounds the number of sequences; partials are 64 rows per (sequence, KV head, split)
  const size_t n_splits = (size_t)(max_ctx + split_tokens - 1) / split_tokens;
  w.attn_ws_bytes = n_splits <= 1 ? 256 : (size_t)rows * d.n_kv_heads * n_splits * 64 *
                                              (d.head_dim + 2) * sizeof(float) + 256;
  w.attn_ws = off; off += align256(w.attn_ws_bytes);
  w.wide_ws_bytes = 0;
  if (wide) {
    const int H = d.hidden, F = d.ffn, QD = d.n_heads * d.head_dim;
    const int up_epi = d.arch == SD_ARCH_LLAMA ? SD_EPI_SILU_MUL : SD_EPI_RELU;
    const size_t need[5] = {
        sd_linear_wide_workspace_bytes(rows, (int)qkv_cols, H, SD_EPI_STORE),
        sd_linear_wide_workspace_bytes(rows, H, QD, SD_EPI_RESIDUAL),
        sd_linear_wide_workspace_bytes(rows, F, H, up_epi),
        sd_linear_wide_workspace_bytes(rows, H, F, SD_EPI_RESIDUAL),
        sd_linear_wide_workspace_bytes(rows, d.lm_vocab > 0 ? d.lm_vocab : d.vocab, H, SD_EPI_F32)};
    for (size_t n : need) w.wide_ws_bytes = n > w.wide_ws_bytes ? n : w.wide_ws_bytes;
  }
  w.wide_ws = off; off += align256(w.wide_ws_bytes);
  // the three-launch draft layer: one f32 partial row per KV head and token row, and the f32
  // accumulator of the down projection
  const size_t fr = (size_t)(rows < SD_DRAFT_LAYER_MAX_ROWS ? rows : SD_DRAFT_LAYER_MAX_ROWS);
  w.slabs = off; off += align256(fr * d.n_kv_heads * d.hidden * sizeof(float));
  w.acc = off;   off += align256(fr * d.hidden * sizeof(float));
  // fp8 activations of the wide family: a scale and its inverse factor per token row
  w.x_scales = off; off += align256((size_t)2 * rows * sizeof(float));
  w.total = off;
  return w;
}

}  // namespace

extern "C" size_t sd_model_workspace_bytes(const sd_model* m, int max_rows, int max_ctx,
                                           int split_tokens) {
  if (!m || max_rows < 1) return 0;
  return plan_workspace(m->d, max_rows, max_ctx, split_tokens, max_rows > 16).total;
}

#define SD_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != SD_OK) return rc__; \
  } while (0)

extern "C" int sd_model_forward(sd_model* m, const sd_kv_cache* kv, const sd_forward_args* a,
                                void* workspace, size_t workspace_bytes, sd_stream_t stream) {
  SD_REQUIRE(m && kv && a && workspace, SD_ERR_INVALID, "sd_model_forward: null argument");
  const sd_model_desc& d = m->d;
  const int rows = a->B * a->q_len;
  const bool wide = a->wide != 0;
  SD_REQUIRE(a->B >= 1 && a->q_len >= 1 && rows <= (wide ? 256 : 16), SD_ERR_UNSUPPORTED,
             "sd_model_forward: B*q_len = %d outside 1..%d", rows, wide ? 256 : 16);
  SD_REQUIRE(a->q_len * (d.n_heads / d.n_kv_heads) <= 64 || wide, SD_ERR_UNSUPPORTED,
             "sd_model_forward: q_len %d * (Hq/Hkv) > 64 needs the wide family", a->q_len);
  SD_REQUIRE((a->token_ids || a->input_embeds) && a->seq_lens && kv->k_cache && kv->v_cache && kv->block_tables,
             SD_ERR_INVALID, "sd_model_forward: null buffer");
  SD_REQUIRE(a->max_ctx >= 1 && a->max_ctx <= d.max_pos, SD_ERR_INVALID,
             "sd_model_forward: max_ctx %d outside 1..max_pos %d", a->max_ctx, d.max_pos);
  const Workspace w = plan_workspace(d, rows, a->max_ctx, a->split_tokens, wide);
  SD_REQUIRE(workspace_bytes >= w.total, SD_ERR_WORKSPACE,
             "sd_model_forward: workspace %zu < %zu bytes", workspace_bytes, w.total);
  SD_REQUIRE(sd_aligned16(workspace), SD_ERR_INVALID, "sd_model_forward: workspace unaligned");

  char* base = (char*)workspace;
  void* h = base + w.h;
  void* xn = base + w.xn;
  void* qkv = base + w.qkv;
  void* attn = base + w.attn;
  void* act = base + w.act;
  void* attn_ws = base + w.attn_ws;
  const int H = d.hidden, D = d.head_dim, Hq = d.n_heads, Hkv = d.n_kv_heads, F = d.ffn;
  const bool llama = d.arch == SD_ARCH_LLAMA;

  if (a->input_embeds) {
    // rows given by the caller (image embeddings merged into the token embeddings)
    SD_REQUIRE(llama, SD_ERR_UNSUPPORTED, "sd_model_forward: input_embeds need a rotary model");
    const hipError_t e = hipMemcpyAsync(h, a->input_embeds, (size_t)rows * H * 2,
                                        hipMemcpyDeviceToDevice, (hipStream_t)stream);
    SD_REQUIRE(e == hipSuccess, SD_ERR_HIP, "sd_model_forward: copying input_embeds: %s",
               hipGetErrorString(e));
  } else {
    SD_TRY(sd_embed(a->token_ids, a->tok_stride, a->tok_offset, d.embed, d.embed_tiled, d.pos_table,
                    d.pos_bias, a->seq_lens, a->pos_offset, h, a->B, a->q_len, H, d.vocab, stream));
  }
  // multimodal rotary positions, for a model that has sections
  const bool mrope = d.mrope_section_t || d.mrope_section_h;
  // A token tree's rows are rotated at their depth, not at their cache slot: explicit
  // positions, every frequency from the one component.
  const bool tree = a->tree_mask != nullptr;
  SD_REQUIRE(!tree || (llama && !mrope && a->rope_pos && !a->input_embeds &&
                       a->q_len * (Hq / Hkv) <= 64),
             SD_ERR_UNSUPPORTED,
             "sd_model_forward: a token tree needs a rotary model without sections, rope_pos, "
             "token ids and q_len * (Hq/Hkv) <= 64");
  const sd_rope_positions rope_pos{a->rope_pos, tree ? nullptr : a->rope_delta,
                                   tree ? D / 2 : d.mrope_section_t, tree ? 0 : d.mrope_section_h};
  const sd_rope_positions* rpp =
      tree || (mrope && (a->rope_pos || a->rope_delta)) ? &rope_pos : nullptr;
  SD_REQUIRE(mrope || tree || (!a->rope_pos && !a->rope_delta), SD_ERR_INVALID,
             "sd_model_forward: rotary positions given to a model without rotary sections");
  // Layers as attention + two fused launches (sd_chain_*), when the caller brought a state and
  // the pass has the shape for it; same bits as the five launches below.
  const bool chain = a->chain_state && !wide && llama && !d.allreduce && !tree && !rpp &&
                     sd_fused_norm_fits(rows, H, false) &&
                     sd_chain_supported(rows, H, Hq * D, F) != 0;
  // The draft model's three-launch layer: launch 1 here, launches 2 and 3 in sd_draft_layer.hip.
  bool fused3 = a->draft_fused && !wide && llama && !d.allreduce && !tree && !rpp && !chain &&
                !d.weight_fp8 && !a->input_embeds && sd_fused_norm_fits(rows, H, false) &&
                sd_draft_layer_supported(rows, a->q_len, H, Hq, Hkv, D, F, a->max_ctx) != 0;
  for (int li = 0; fused3 && li < d.n_layers; ++li)
    fused3 = m->layers[li].down_w_km && !m->layers[li].qkv_b && !m->layers[li].o_b &&
             !m->layers[li].up_b && !m->layers[li].down_b;
  if (fused3) {
    float* slabs = (float*)(base + w.slabs);
    float* acc = (float*)(base + w.acc);
    void* h_cur = h;
    void* h_alt = xn;
    for (int li = 0; li < d.n_layers; ++li) {
      const sd_layer_weights& l = m->layers[li];
      void* kc = (char*)kv->k_cache + (size_t)li * kv->layer_stride_bytes;
      void* vc = (char*)kv->v_cache + (size_t)li * kv->layer_stride_bytes;
      // 1. (h, the previous layer's down-projection sums) -> q, cache entries, new residual rows
      const sd_row_adds in1{li ? acc : nullptr, H, h_alt};
      SD_TRY(sd_norm_qkv_rope_adds(l.qkv_w, nullptr, h_cur, H, l.attn_norm_w, nullptr, d.norm_eps, 0,
                                   nullptr, qkv, (int64_t)Hq * D, d.rope_cs, d.max_pos, kc, vc,
                                   kv->block_tables, kv->max_blocks_per_seq, a->seq_lens, a->pos_offset,
                                   a->B, a->q_len, Hq, Hkv, D, H, nullptr, &in1, stream));
      if (li) { void* t = h_cur; h_cur = h_alt; h_alt = t; }
      // 2. attention + output projection -> one partial row per KV head; clears the accumulator
      SD_TRY(sd_draft_attn_oproj(qkv, (int64_t)Hq * D, kc, vc, kv->block_tables, kv->max_blocks_per_seq,
                                 a->seq_lens, a->pos_offset, l.o_w, slabs, acc, rows * H, a->B,
                                 a->q_len, Hq, Hkv, D, H, d.attn_scale, a->max_ctx, stream));
      // 3. (h, the Hkv partial rows) -> new residual rows; MLP sums into the accumulator
      SD_TRY(sd_draft_mlp(h_cur, H, slabs, Hkv, h_alt, l.mlp_norm_w, d.norm_eps, l.up_w, l.down_w_km,
                          acc, rows, H, F, stream));
      { void* t = h_cur; h_cur = h_alt; h_alt = t; }
    }
    if (a->logits) {
      const int lm_vocab = d.lm_vocab > 0 ? d.lm_vocab : d.vocab;
      const sd_row_adds fin{acc, H, nullptr};
      SD_TRY(sd_norm_linear_adds(d.lm_head, nullptr, h_cur, H, d.final_norm_w, nullptr, d.norm_eps, 0,
                                 nullptr, a->logits, lm_vocab, rows, lm_vocab, H, SD_EPI_F32, &fin, stream));
    }
    if (a->argmax_val && a->argmax_idx) {  // (this head takes an additive input: through its logits)
      const int lm_vocab = d.lm_vocab > 0 ? d.lm_vocab : d.vocab;
      SD_REQUIRE(a->logits, SD_ERR_INVALID,
                 "sd_model_forward: a three-launch pass needs `logits` for the slice maxima");
      SD_TRY(sd_argmax_partial(a->logits, lm_vocab, a->argmax_val, a->argmax_idx, rows, lm_vocab, stream));
    }
    return SD_OK;
  }
  for (int li = 0; li < d.n_layers; ++li) {
    const sd_layer_weights& l = m->layers[li];
    void* kc = (char*)kv->k_cache + (size_t)li * kv->layer_stride_bytes;
    void* vc = (char*)kv->v_cache + (size_t)li * kv->layer_stride_bytes;
    if (wide) {
      // 17..256 token rows: stand-alone norms, activations shared through LDS by the wide
      // linear kernels; same layer otherwise.
      void* wws = base + w.wide_ws;
      const bool lead = !d.allreduce || d.tp_rank == 0;
      const int proj_epi = lead ? SD_EPI_RESIDUAL : SD_EPI_STORE;
      // fp8 activations (opt-in): every matrix of the layer is fp8 and every width a multiple of 128
      const bool a8 = a->wide_a8 && llama && l.qkv_s && l.o_s && l.up_s && l.down_s && H % 128 == 0 &&
                      F % 128 == 0 && (Hq * D) % 128 == 0;
      float* xsc = reinterpret_cast<float*>(base + w.x_scales);
      if (llama) SD_TRY(sd_rmsnorm(h, H, l.attn_norm_w, xn, H, rows, H, d.norm_eps, stream));
      else SD_TRY(sd_layernorm(h, H, l.attn_norm_w, l.attn_norm_b, xn, H, rows, H, d.norm_eps, stream));
      if (a8) {
        SD_TRY(sd_row_scales_fp8(xn, H, rows, H, xsc, stream));
        SD_TRY(sd_qkv_rope_wide_a8(l.qkv_w, l.qkv_s, xsc, xn, H, l.qkv_b, qkv, (int64_t)Hq * D,
                                   d.rope_cs, d.max_pos, kc, vc, kv->block_tables,
                                   kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, a->B, a->q_len,
                                   Hq, Hkv, D, H, rpp, wws, w.wide_ws_bytes, stream));
      } else {
        SD_TRY(sd_qkv_rope_wide(l.qkv_w, l.qkv_s, xn, H, l.qkv_b, qkv, (int64_t)Hq * D,
                                llama ? d.rope_cs : nullptr, d.max_pos, kc, vc, kv->block_tables,
                                kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, a->B, a->q_len,
                                Hq, Hkv, D, H, rpp, wws, w.wide_ws_bytes, stream));
      }
      // One workgroup column covers 64 / (Hq/Hkv) query tokens of a sequence; a longer chunk
      // (prefill) is attended in blocks of that many tokens by the same launch (their K/V is
      // in the cache already, the causal mask is per row).
      const int q_blk = 64 / (Hq / Hkv);
      if (tree) {
        SD_TRY(sd_paged_attention_tree(qkv, (int64_t)Hq * D, kc, vc, kv->block_tables,
                                       kv->max_blocks_per_seq, a->seq_lens, a->pos_offset,
                                       a->tree_mask, attn, (int64_t)Hq * D, a->B, a->q_len, Hq, Hkv,
                                       D, d.attn_scale, a->max_ctx, a->split_tokens, attn_ws,
                                       w.attn_ws_bytes, stream));
      } else {
        SD_TRY(sd_paged_attention_blocked(qkv, (int64_t)Hq * D, kc, vc, kv->block_tables,
                                          kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, attn,
                                          (int64_t)Hq * D, a->B, a->q_len,
                                          a->q_len < q_blk ? a->q_len : q_blk, Hq, Hkv, D,
                                          d.attn_scale, a->max_ctx, a->split_tokens, attn_ws,
                                          w.attn_ws_bytes, stream));
      }
      if (a8) {
        SD_TRY(sd_row_scales_fp8(attn, (int64_t)Hq * D, rows, Hq * D, xsc, stream));
        SD_TRY(sd_linear_wide_a8(l.o_w, l.o_s, xsc, attn, (int64_t)Hq * D, lead ? l.o_b : nullptr, h, h, H,
                                 rows, H, Hq * D, proj_epi, wws, w.wide_ws_bytes, stream));
      } else {
        SD_TRY(sd_linear_wide(l.o_w, l.o_s, attn, (int64_t)Hq * D, lead ? l.o_b : nullptr, h, h, H, rows, H,
                              Hq * D, proj_epi, wws, w.wide_ws_bytes, stream));
      }
      if (d.allreduce) SD_TRY(d.allreduce(h, (size_t)rows * H, d.allreduce_ctx, stream));
      if (llama) SD_TRY(sd_rmsnorm(h, H, l.mlp_norm_w, xn, H, rows, H, d.norm_eps, stream));
      else SD_TRY(sd_layernorm(h, H, l.mlp_norm_w, l.mlp_norm_b, xn, H, rows, H, d.norm_eps, stream));
      if (a8) {
        SD_TRY(sd_row_scales_fp8(xn, H, rows, H, xsc, stream));
        SD_TRY(sd_linear_wide_a8(l.up_w, l.up_s, xsc, xn, H, l.up_b, nullptr, act, F, rows, F, H,
                                 SD_EPI_SILU_MUL, wws, w.wide_ws_bytes, stream));
        SD_TRY(sd_row_scales_fp8(act, F, rows, F, xsc, stream));
        SD_TRY(sd_linear_wide_a8(l.down_w, l.down_s, xsc, act, F, lead ? l.down_b : nullptr, h, h, H, rows,
                                 H, F, proj_epi, wws, w.wide_ws_bytes, stream));
      } else {
        SD_TRY(sd_linear_wide(l.up_w, l.up_s, xn, H, l.up_b, nullptr, act, F, rows, F, H,
                              llama ? SD_EPI_SILU_MUL : SD_EPI_RELU, wws, w.wide_ws_bytes, stream));
        SD_TRY(sd_linear_wide(l.down_w, l.down_s, act, F, lead ? l.down_b : nullptr, h, h, H, rows, H, F,
                              proj_epi, wws, w.wide_ws_bytes, stream));
      }
      if (d.allreduce) SD_TRY(d.allreduce(h, (size_t)rows * H, d.allreduce_ctx, stream));
      continue;
    }
    const bool fuse = sd_fused_norm_fits(rows, H, !llama);
    if (chain && li > 0) {
      // q and the cache entries of this layer came out of the previous layer's fused launch
    } else if (fuse) {
      SD_TRY(sd_norm_qkv_rope(l.qkv_w, l.qkv_s, h, H, l.attn_norm_w, l.attn_norm_b, d.norm_eps, !llama,
                              l.qkv_b, qkv, (int64_t)Hq * D, llama ? d.rope_cs : nullptr, d.max_pos,
                              kc, vc, kv->block_tables, kv->max_blocks_per_seq, a->seq_lens,
                              a->pos_offset, a->B, a->q_len, Hq, Hkv, D, H, rpp, stream));
    } else {
      if (llama) SD_TRY(sd_rmsnorm(h, H, l.attn_norm_w, xn, H, rows, H, d.norm_eps, stream));
      else SD_TRY(sd_layernorm(h, H, l.attn_norm_w, l.attn_norm_b, xn, H, rows, H, d.norm_eps, stream));
      SD_TRY(sd_norm_qkv_rope(l.qkv_w, l.qkv_s, xn, H, nullptr, nullptr, 0.f, 0, l.qkv_b, qkv,
                              (int64_t)Hq * D, llama ? d.rope_cs : nullptr, d.max_pos, kc, vc,
                              kv->block_tables, kv->max_blocks_per_seq, a->seq_lens, a->pos_offset,
                              a->B, a->q_len, Hq, Hkv, D, H, rpp, stream));
    }
    // With few rows and splits the output projection merges the attention partials itself
    // (every workgroup redoes the small merge) and the merge launch is saved.
    const int n_splits = (a->max_ctx + a->split_tokens - 1) / a->split_tokens;
    const bool merge_in_proj = n_splits > 1 && n_splits <= 8 && rows * n_splits <= 16 &&
                               sd_fused_norm_fits(rows, Hq * D, false);
    if (a->read_ahead_mb > 0 && !chain) {
      // the attention launch's idle compute units fetch what the next two launches stream
      const size_t cap = (size_t)a->read_ahead_mb << 20;
      sd_pf_range pf[2];
      pf[0] = sd_linear_pf_range(l.o_w, H, Hq * D, SD_EPI_RESIDUAL, l.o_s != nullptr, 0, cap);
      const size_t took = (size_t)pf[0].n_units * (pf[0].half ? pf[0].tile_bytes / 2 : pf[0].tile_bytes);
      pf[1] = sd_linear_pf_range(l.up_w, F, H, llama ? SD_EPI_SILU_MUL : SD_EPI_RELU,
                                 l.up_s != nullptr, fuse, cap > took ? cap - took : 0);
      sd_attn_prefetch_next(pf, 2);
    }
    if (tree) {
      SD_TRY(sd_paged_attention_tree(qkv, (int64_t)Hq * D, kc, vc, kv->block_tables,
                                     kv->max_blocks_per_seq, a->seq_lens, a->pos_offset,
                                     a->tree_mask, merge_in_proj ? nullptr : attn, (int64_t)Hq * D,
                                     a->B, a->q_len, Hq, Hkv, D, d.attn_scale, a->max_ctx,
                                     a->split_tokens, attn_ws, w.attn_ws_bytes, stream));
    } else {
      SD_TRY(sd_paged_attention(qkv, (int64_t)Hq * D, kc, vc, kv->block_tables,
                                kv->max_blocks_per_seq, a->seq_lens, a->pos_offset,
                                merge_in_proj ? nullptr : attn, (int64_t)Hq * D, a->B, a->q_len, Hq,
                                Hkv, D, d.attn_scale, a->max_ctx, a->split_tokens, attn_ws,
                                w.attn_ws_bytes, stream));
    }
    // Row-parallel projections under tensor parallelism: rank 0 adds the residual and the
    // bias, the others store their partial product, and the sum over ranks lands in h.
    if (chain) {
      SD_TRY(sd_chain_o_up(a->chain_state, l.o_w, l.o_s, l.o_b, merge_in_proj ? nullptr : attn,
                           (int64_t)Hq * D, attn_ws, w.attn_ws_bytes, a->seq_lens, a->pos_offset,
                           a->B, a->q_len, Hq, Hkv, D, a->max_ctx, a->split_tokens, h, H,
                           l.mlp_norm_w, d.norm_eps, l.up_w, l.up_s, l.up_b, act, F, F, stream));
      if (li + 1 < d.n_layers) {
        const sd_layer_weights& nx = m->layers[li + 1];
        SD_TRY(sd_chain_down_qkv(a->chain_state, l.down_w, l.down_s, l.down_b, act, F, F, h, H,
                                 nx.attn_norm_w, d.norm_eps, nx.qkv_w, nx.qkv_s, nx.qkv_b, qkv,
                                 (int64_t)Hq * D, d.rope_cs, d.max_pos,
                                 (char*)kc + kv->layer_stride_bytes,
                                 (char*)vc + kv->layer_stride_bytes, kv->block_tables,
                                 kv->max_blocks_per_seq, a->seq_lens, a->pos_offset, a->B, a->q_len,
                                 Hq, Hkv, D, stream));
      } else {
        SD_TRY(sd_linear(l.down_w, l.down_s, act, F, l.down_b, h, h, H, rows, H, F, SD_EPI_RESIDUAL,
                         stream));
      }
      continue;
    }
    const bool lead = !d.allreduce || d.tp_rank == 0;
    const int proj_epi = lead ? SD_EPI_RESIDUAL : SD_EPI_STORE;
    if (merge_in_proj) {
      SD_TRY(sd_linear_attn_merge(l.o_w, l.o_s, attn_ws, w.attn_ws_bytes, a->seq_lens, a->pos_offset,
                                  a->B, a->q_len, Hq, Hkv, D, a->max_ctx, a->split_tokens,
                                  lead ? l.o_b : nullptr, h, h, H, H, proj_epi, stream));
    } else {
      SD_TRY(sd_linear(l.o_w, l.o_s, attn, (int64_t)Hq * D, lead ? l.o_b : nullptr, h, h, H, rows, H,
                       Hq * D, proj_epi, stream));
    }
    if (d.allreduce) SD_TRY(d.allreduce(h, (size_t)rows * H, d.allreduce_ctx, stream));
    const int act_epi = llama ? SD_EPI_SILU_MUL : SD_EPI_RELU;
    if (fuse) {
      SD_TRY(sd_norm_linear(l.up_w, l.up_s, h, H, l.mlp_norm_w, l.mlp_norm_b, d.norm_eps, !llama, l.up_b,
                            act, F, rows, F, H, act_epi, stream));
    } else {
      if (llama) SD_TRY(sd_rmsnorm(h, H, l.mlp_norm_w, xn, H, rows, H, d.norm_eps, stream));
      else SD_TRY(sd_layernorm(h, H, l.mlp_norm_w, l.mlp_norm_b, xn, H, rows, H, d.norm_eps, stream));
      SD_TRY(sd_linear(l.up_w, l.up_s, xn, H, l.up_b, nullptr, act, F, rows, F, H, act_epi, stream));
    }
    SD_TRY(sd_linear(l.down_w, l.down_s, act, F, lead ? l.down_b : nullptr, h, h, H, rows, H, F, proj_epi,
                     stream));
    if (d.allreduce) SD_TRY(d.allreduce(h, (size_t)rows * H, d.allreduce_ctx, stream));
  }
  const int lm_vocab = d.lm_vocab > 0 ? d.lm_vocab : d.vocab;
  if (a->logits && wide) {
    if (llama) SD_TRY(sd_rmsnorm(h, H, d.final_norm_w, xn, H, rows, H, d.norm_eps, stream));
    else SD_TRY(sd_layernorm(h, H, d.final_norm_w, d.final_norm_b, xn, H, rows, H, d.norm_eps, stream));
    if (a->wide_a8 && llama && d.lm_head_s && H % 128 == 0) {
      float* xsc = reinterpret_cast<float*>(base + w.x_scales);
      SD_TRY(sd_row_scales_fp8(xn, H, rows, H, xsc, stream));
      SD_TRY(sd_linear_wide_a8(d.lm_head, d.lm_head_s, xsc, xn, H, nullptr, nullptr, a->logits, lm_vocab, rows,
                               lm_vocab, H, SD_EPI_F32, base + w.wide_ws, w.wide_ws_bytes, stream));
    } else {
      SD_TRY(sd_linear_wide(d.lm_head, d.lm_head_s, xn, H, nullptr, nullptr, a->logits, lm_vocab, rows, lm_vocab, H,
                            SD_EPI_F32, base + w.wide_ws, w.wide_ws_bytes, stream));
    }
  } else if (a->argmax_val && a->argmax_idx && !fused3 && sd_fused_norm_fits(rows, H, !llama)) {
    // greedy pass: the head's workgroups leave the slice maxima themselves, no logits
    SD_TRY(sd_norm_linear_argmax(d.lm_head, d.lm_head_s, h, H, d.final_norm_w, d.final_norm_b, d.norm_eps,
                                 !llama, a->argmax_val, a->argmax_idx, rows, lm_vocab, H, stream));
    return SD_OK;
  } else if (a->logits) {
    if (sd_fused_norm_fits(rows, H, !llama)) {
      SD_TRY(sd_norm_linear(d.lm_head, d.lm_head_s, h, H, d.final_norm_w, d.final_norm_b, d.norm_eps, !llama,
                            nullptr, a->logits, lm_vocab, rows, lm_vocab, H, SD_EPI_F32, stream));
    } else {
      if (llama) SD_TRY(sd_rmsnorm(h, H, d.final_norm_w, xn, H, rows, H, d.norm_eps, stream));
      else SD_TRY(sd_layernorm(h, H, d.final_norm_w, d.final_norm_b, xn, H, rows, H, d.norm_eps, stream));
      SD_TRY(sd_linear(d.lm_head, d.lm_head_s, xn, H, nullptr, nullptr, a->logits, lm_vocab, rows, lm_vocab, H,
                       SD_EPI_F32, stream));
    }
  }
  if (a->argmax_val && a->argmax_idx) {  // (the head could not leave the slice maxima itself)
    SD_REQUIRE(a->logits, SD_ERR_INVALID,
               "sd_model_forward: this pass's shape needs `logits` for the slice maxima");
    SD_TRY(sd_argmax_partial(a->logits, lm_vocab, a->argmax_val, a->argmax_idx, rows, lm_vocab, stream));
  }
  return SD_OK;
}

struct sd_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" int sd_graph_begin(sd_stream_t stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    sd_set_error("sd_graph_begin: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  return SD_OK;
}

extern "C" int sd_graph_end(sd_stream_t stream, sd_graph** out) {
  SD_REQUIRE(out, SD_ERR_INVALID, "sd_graph_end: null out");
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) {
    sd_set_error("sd_graph_end: capture failed: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    sd_set_error("sd_graph_end: instantiate failed: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  *out = new sd_graph{g, ex};
  return SD_OK;
}

extern "C" int sd_graph_launch(sd_graph* g, sd_stream_t stream) {
  SD_REQUIRE(g, SD_ERR_INVALID, "sd_graph_launch: null graph");
  hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    sd_set_error("sd_graph_launch: %s", hipGetErrorString(e));
    return SD_ERR_HIP;
  }
  return SD_OK;
}

extern "C" void sd_graph_destroy(sd_graph* g) {
  if (!g) return;
  (void)hipGraphExecDestroy(g->exec);   // nothing to report to: the handle is gone either way
  (void)hipGraphDestroy(g->graph);
  delete g;
}

// ------------------------------------------------------------------ tensor-parallel group
struct sd_tp_comm {
  ncclComm_t comm;
  int rank, world;
};

static_assert(sizeof(ncclUniqueId) <= SD_TP_ID_BYTES, "RCCL unique id does not fit");

extern "C" int sd_tp_unique_id(void* id_out) {
  SD_REQUIRE(id_out, SD_ERR_INVALID, "sd_tp_unique_id: null pointer");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  SD_REQUIRE(r == ncclSuccess, SD_ERR_HIP, "sd_tp_unique_id: %s", ncclGetErrorString(r));
  memset(id_out, 0, SD_TP_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return SD_OK;
}

extern "C" sd_tp_comm* sd_tp_comm_create(const void* id, int rank, int world_size) {
  if (!id || world_size < 1 || rank < 0 || rank >= world_size) {
    sd_set_error("sd_tp_comm_create: bad arguments (rank %d of %d)", rank, world_size);
    return nullptr;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm;
  const ncclResult_t r = ncclCommInitRank(&comm, world_size, uid, rank);
  if (r != ncclSuccess) {
    sd_set_error("sd_tp_comm_create: ncclCommInitRank: %s", ncclGetErrorString(r));
    return nullptr;
  }
  return new sd_tp_comm{comm, rank, world_size};
}

extern "C" void sd_tp_comm_destroy(sd_tp_comm* c) {
  if (!c) return;
  ncclCommDestroy(c->comm);
  delete c;
}

extern "C" int sd_tp_allgather(const void* send, void* recv, size_t bytes_per_rank, void* ctx,
                               sd_stream_t stream) {
  SD_REQUIRE(send && recv && ctx, SD_ERR_INVALID, "sd_tp_allgather: null pointer");
  sd_tp_comm* c = static_cast<sd_tp_comm*>(ctx);
  const ncclResult_t r = ncclAllGather(send, recv, bytes_per_rank, ncclChar, c->comm,
                                       (hipStream_t)stream);
  SD_REQUIRE(r == ncclSuccess, SD_ERR_HIP, "sd_tp_allgather: %s", ncclGetErrorString(r));
  return SD_OK;
}

extern "C" int sd_tp_allreduce(void* buf, size_t n_bf16, void* ctx, sd_stream_t stream) {
  SD_REQUIRE(buf && ctx, SD_ERR_INVALID, "sd_tp_allreduce: null pointer");
  sd_tp_comm* c = static_cast<sd_tp_comm*>(ctx);
  const ncclResult_t r = ncclAllReduce(buf, buf, n_bf16, ncclBfloat16, ncclSum, c->comm,
                                       (hipStream_t)stream);
  SD_REQUIRE(r == ncclSuccess, SD_ERR_HIP, "sd_tp_allreduce: %s", ncclGetErrorString(r));
  return SD_OK;
}
