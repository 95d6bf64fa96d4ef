"""DecoderModel: the host handle of one model (target or draft) on one GPU.

It owns the native model descriptor (sd_model_create) and a workspace, and
enqueues whole forward passes with one C call; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops
from .config import ModelConfig
from .kv_cache import PagedKVCache
from .weights import ModelWeights

MAX_ROWS = 16        # token rows one forward pass may carry (B * q_len), narrow kernel family
MAX_ROWS_WIDE = 256  # the same for the wide family (batched draft passes / verify)
PREFILL_ROWS = 256   # prompt tokens of one sequence per prefill pass (wide family)


def _ptr(t):
    if t is None:
        return None
    return t.data.data_ptr() if hasattr(t, "mode") else t.data_ptr()


class DecoderModel:
    def __init__(self, weights: ModelWeights, max_ctx: int, split_tokens: int = 256,
                 tp_group=None, wide: bool = False, wide_prefill: bool = True,
                 fused_pairs: bool = False, fused_layers: bool = False,
                 read_ahead_mb: int = 0, fp8_activations: bool = False):
        """tp_group (parallel.TensorParallelGroup or any object with rank, allreduce_ptr,
        ctx_ptr): `weights` is then this rank's shard (weights.shard_state_dict)."""
        self.weights = weights  # keeps the device tensors alive
        self.tp_group = tp_group  # keeps the communicator / callback alive
        self.cfg: ModelConfig = weights.cfg
        cfg = self.cfg
        fp8 = bool(getattr(weights, "fp8", False))
        # wide: every pass of this model goes through the 17..256-row linear kernels (one
        # kernel family per model, because the two round differently)
        self.wide = bool(wide)
        self.max_rows = MAX_ROWS_WIDE if wide else MAX_ROWS
        # Prompt chunks of one sequence always may take the wide family (up to PREFILL_ROWS
        # tokens per pass): every engine prefills the same way, so streams stay comparable.
        self.prefill_rows = PREFILL_ROWS if (wide or wide_prefill) else min(
            MAX_ROWS, 64 // (cfg.n_heads // cfg.n_kv_heads))
        if max_ctx > cfg.max_pos:
            raise ValueError(f"max_ctx {max_ctx} exceeds the model's max_pos {cfg.max_pos}")
        self.max_ctx = max_ctx
        self.split_tokens = split_tokens
        # MiB of the next launches' weights that every layer's attention launch reads ahead on
        # its idle compute units (sd_forward_args.read_ahead_mb); results do not change
        self.read_ahead_mb = int(read_ahead_mb)
        # passes of the wide family quantise their activations to fp8 per token row and multiply
        # on the fp8 matrix instruction (sd_forward_args.wide_a8); fp8 weights only
        if fp8_activations and not fp8:
            raise ValueError("fp8_activations needs fp8 weights")
        self.fp8_activations = bool(fp8_activations)
        lib = _lib.load()
        # fused_layers (draft models): passes of up to ops.DRAFT_LAYER_MAX_ROWS rows run every
        # layer as three launches (sd_draft_attn_oproj, sd_draft_mlp); they need the down
        # projection's tiles a second time, ordered k-step first.  The rows differ from the
        # five-launch layer's in the last bits, which a draft model may.
        self.fused_layers = bool(
            fused_layers and tp_group is None and not fp8 and cfg.arch == "llama" and not cfg.mrope_section
            and not cfg.qkv_bias and ops.on_gpu()
            and ops.draft_layer_supported(1, 1, cfg.hidden, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim,
                                          cfg.ffn, min(max_ctx, ops.DRAFT_LAYER_MAX_CTX)))
        self._down_km = [ops.kmajor_tiles(l.down_w) for l in weights.layers] if self.fused_layers else None
        layers = (_lib.LayerWeights * cfg.n_layers)()
        for i, l in enumerate(weights.layers):
            for name, _ in _lib.LayerWeights._fields_:
                if name == "down_w_km":
                    setattr(layers[i], name, _ptr(self._down_km[i]) if self._down_km else None)
                elif name.endswith("_s"):   # scale of the fp8 matrix of the same stem
                    setattr(layers[i], name, _ptr(getattr(l, name[:-2] + "_w").scale))
                else:
                    setattr(layers[i], name, _ptr(getattr(l, name)))
        desc = _lib.ModelDesc(
            arch=0 if cfg.arch == "llama" else 1, n_layers=cfg.n_layers, hidden=cfg.hidden,
            n_heads=cfg.n_heads, n_kv_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, ffn=cfg.ffn,
            vocab=cfg.vocab, max_pos=cfg.max_pos, pos_bias=cfg.pos_bias if cfg.arch == "opt" else 0,
            norm_eps=cfg.norm_eps, attn_scale=cfg.head_dim ** -0.5,
            embed=_ptr(weights.embed), embed_tiled=int(weights.embed_tiled),
            pos_table=_ptr(weights.pos_table),
            rope_cs=_ptr(weights.rope_cs), final_norm_w=_ptr(weights.final_norm_w),
            final_norm_b=_ptr(weights.final_norm_b), lm_head=_ptr(weights.lm_head), layers=layers,
            tp_rank=tp_group.rank if tp_group else 0,
            allreduce=tp_group.allreduce_ptr if tp_group else None,
            allreduce_ctx=tp_group.ctx_ptr if tp_group else None,
            weight_fp8=int(fp8), lm_head_s=_ptr(weights.lm_head.scale), lm_vocab=cfg.lm_vocab,
            mrope_section_t=cfg.mrope_section[0] if cfg.mrope_section else 0,
            mrope_section_h=cfg.mrope_section[1] if cfg.mrope_section else 0)
        self._handle = lib.sd_model_create(C.byref(desc))
        if not self._handle:
            raise _lib.SdError(f"sd_model_create failed: {_lib.last_error()}")
        nbytes = lib.sd_model_workspace_bytes(self._handle, max(self.max_rows, self.prefill_rows),
                                              max_ctx, split_tokens)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=weights.embed.device)
        # fused_pairs: passes of <= 4 rows run each layer as attention + sd_chain_o_up +
        # sd_chain_down_qkv (three launches for five, same bits).  Opt-in: on MI355X the five
        # launches are faster (an in-launch exchange costs what a launch boundary costs).
        self.chain_state = None
        if fused_pairs:
            if tp_group is not None or cfg.arch != "llama" or cfg.mrope_section or not ops.on_gpu() \
                    or not ops.chain_supported(1, cfg.hidden, cfg.n_heads * cfg.head_dim, cfg.ffn):
                raise _lib.SdError("fused_pairs needs an unsharded rotary model on the GPU build whose "
                                   "hidden, Hq*D and ffn widths are multiples of 2048")
            self.chain_state = ops.chain_state(cfg.hidden, device=weights.embed.device)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.load().sd_model_destroy(h)
            except Exception:
                pass
            self._handle = None

    def forward(self, kv: PagedKVCache, token_ids: torch.Tensor, tok_stride: int, tok_offset: int,
                seq_lens: torch.Tensor, pos_offset: int, B: int, q_len: int,
                logits: Optional[torch.Tensor], max_ctx: Optional[int] = None,
                wide: Optional[bool] = None, input_embeds: Optional[torch.Tensor] = None,
                rope_pos: Optional[torch.Tensor] = None,
                rope_delta: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None,
                tree_mask: Optional[torch.Tensor] = None,
                argmax_parts: Optional[tuple] = None) -> None:
        """Enqueues one pass over B*q_len token rows on the current stream.
        Token t of sequence b is token_ids[b*tok_stride + tok_offset + t] at
        position seq_lens[b] + pos_offset + t.  `logits` ([B*q_len, vocab] f32)
        may be None to skip the LM head.  Multimodal prompts: input_embeds (bf16
        [B*q_len, hidden]) replace the embedding lookup; a model with mrope_section takes
        its rotary positions from rope_pos (int32 [3, B*q_len]) or, for text after an image,
        from the cache position + rope_delta[b] (int32 [B]).  workspace: another buffer of
        the size of self.workspace, for passes that run concurrently on several streams.
        Token trees: tree_mask (int64 [B*q_len], bit j of entry b*q_len+t: row t attends new
        row j) with rope_pos = the rows' rotary positions (int32 [B*q_len]); rotary models
        without mrope_section only.  argmax_parts (part_val f32, part_idx i32, each
        [B*q_len, ops.ARGMAX_PARTS]): the pass leaves the slice maxima of every row's logits there
        (greedy consumers: ops.propose_from_partials / verify_from_partials /
        argmax_from_partials), straight from the LM head where the pass's shape allows (then
        `logits` is not written), else through `logits`, which must then be given."""
        ws = self.workspace if workspace is None else workspace
        if ws.numel() < self.workspace.numel() or ws.dtype != torch.uint8:
            raise _lib.SdError("workspace must be a uint8 tensor of at least self.workspace's size")
        rows = B * q_len
        if input_embeds is not None and (input_embeds.dtype != torch.bfloat16
                                         or not input_embeds.is_contiguous()
                                         or tuple(input_embeds.shape) != (rows, self.cfg.hidden)):
            raise _lib.SdError("input_embeds must be a contiguous bfloat16 [B*q_len, hidden] tensor")
        if tree_mask is not None:
            if (tree_mask.dtype != torch.int64 or not tree_mask.is_contiguous()
                    or tree_mask.numel() < rows):
                raise _lib.SdError("tree_mask must be a contiguous int64 tensor of B*q_len entries")
            if (rope_pos is None or rope_pos.dtype != torch.int32 or not rope_pos.is_contiguous()
                    or rope_pos.numel() < rows or self.cfg.mrope_section or self.cfg.arch != "llama"):
                raise _lib.SdError("a token tree needs int32 rope_pos [B*q_len] and a rotary model "
                                   "without mrope_section")
        elif rope_pos is not None and (rope_pos.dtype != torch.int32 or not rope_pos.is_contiguous()
                                       or tuple(rope_pos.shape) != (3, rows)):
            raise _lib.SdError("rope_pos must be a contiguous int32 [3, B*q_len] tensor")
        if rope_delta is not None and (rope_delta.dtype != torch.int32 or rope_delta.numel() < B):
            raise _lib.SdError("rope_delta must be an int32 tensor of at least B entries")
        if argmax_parts is not None:
            pv, pi = argmax_parts
            if (pv.dtype != torch.float32 or pi.dtype != torch.int32 or not pv.is_contiguous()
                    or not pi.is_contiguous() or pv.numel() < rows * ops.ARGMAX_PARTS
                    or pi.numel() < rows * ops.ARGMAX_PARTS):
                raise _lib.SdError("argmax_parts: contiguous (f32, i32) tensors of [B*q_len, ARGMAX_PARTS]")
        if ((rope_pos is not None and tree_mask is None) or rope_delta is not None) and not self.cfg.mrope_section:
            raise _lib.SdError("rotary positions given to a model without mrope_section")
        if token_ids.dtype != torch.int32 or seq_lens.dtype != torch.int32:
            raise _lib.SdError("token_ids and seq_lens must be int32 device tensors")
        if logits is not None and (logits.dtype != torch.float32 or not logits.is_contiguous()
                                   or logits.numel() < B * q_len * self.cfg.head_rows):
            raise _lib.SdError("logits must be a contiguous float32 [B*q_len, head rows] tensor")
        kvd = _lib.KvCacheDesc(k_cache=kv.k.data_ptr(), v_cache=kv.v.data_ptr(),
                               layer_stride_bytes=kv.layer_stride_bytes,
                               block_tables=kv.block_tables.data_ptr(),
                               max_blocks_per_seq=kv.max_blocks_per_seq)
        args = _lib.ForwardArgs(B=B, q_len=q_len, token_ids=token_ids.data_ptr(),
                                tok_stride=tok_stride, tok_offset=tok_offset,
                                seq_lens=seq_lens.data_ptr(), pos_offset=pos_offset,
                                max_ctx=max_ctx or self.max_ctx, split_tokens=self.split_tokens,
                                logits=_ptr(logits), wide=int(self.wide if wide is None else wide),
                                input_embeds=_ptr(input_embeds), rope_pos=_ptr(rope_pos),
                                rope_delta=_ptr(rope_delta), tree_mask=_ptr(tree_mask),
                                chain_state=_ptr(self.chain_state) if workspace is None else None,
                                draft_fused=int(self.fused_layers),
                                read_ahead_mb=int(self.read_ahead_mb),
                                wide_a8=int(self.fp8_activations),
                                argmax_val=_ptr(argmax_parts[0]) if argmax_parts else None,
                                argmax_idx=_ptr(argmax_parts[1]) if argmax_parts else None)
        rc = _lib.load().sd_model_forward(self._handle, C.byref(kvd), C.byref(args),
                                          ws.data_ptr(), ws.numel(),
                                          ops._stream())
        _lib.check(rc, "sd_model_forward")
