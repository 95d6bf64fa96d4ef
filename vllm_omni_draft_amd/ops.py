"""Tensor-level wrappers over the C ABI: unwrap `data_ptr()`, pass the current
HIP stream, raise on any non-zero status.  No computation happens in Python
and nothing here falls back to PyTorch ops."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

EPI_STORE, EPI_RESIDUAL, EPI_SILU_MUL, EPI_RELU, EPI_F32, EPI_GELU_TANH = range(6)
PAGE = 16


def on_gpu() -> bool:
    """False only in a process started with SD_AMD_BACKEND=cpu (the CPU build of the ABI)."""
    return _lib.backend() == "gpu"


def _stream():
    return torch.cuda.current_stream().cuda_stream if on_gpu() else None


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if on_gpu() and not t.is_cuda:
        raise _lib.SdError(f"{name}: tensor must live on the GPU (got {t.device}); "
                           "the GPU library has no CPU path")
    if not on_gpu() and t.is_cuda:
        raise _lib.SdError(f"{name}: this process runs the CPU build of the library "
                           f"(SD_AMD_BACKEND=cpu) and takes host tensors, got {t.device}")
    if t.dtype != dtype:
        raise _lib.SdError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.SdError(f"{name}: tensor must be contiguous")


def device_info():
    import ctypes as C
    cu = C.c_int(0)
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().sd_device_info(C.byref(cu), buf, 128), "sd_device_info")
    return cu.value, buf.value.decode()


# ---------------------------------------------------------------- acceptance
def sampling_scratch(rows: int, device="cuda") -> torch.Tensor:
    """Scratch for the split (many-workgroups-per-row) forms of softmax_rows, sample_rows and
    rejection_sample over up to `rows` rows / sequences."""
    return torch.empty(max(int(_lib.load().sd_sampling_scratch_bytes(rows)), 16),
                       dtype=torch.uint8, device=device)


def _scratch_ok(scratch, rows, what):
    if scratch.numel() < _lib.load().sd_sampling_scratch_bytes(rows):
        raise _lib.SdError(f"{what}: scratch too small for {rows} rows")


def rejection_sample(target_probs, bonus_ids, draft_probs, draft_ids, u_accept, u_recover,
                     proposal_lens=None, counters=None, scratch=None, out=None) -> torch.Tensor:
    """scratch (sampling_scratch(B)): use the split form."""
    B, k1, V = target_probs.shape
    k = k1 - 1
    _need(target_probs, torch.float32, "target_probs")
    _need(draft_probs, torch.float32, "draft_probs")
    _need(draft_ids, torch.int64, "draft_ids")
    _need(u_accept, torch.float32, "u_accept")
    _need(u_recover, torch.float32, "u_recover")
    bonus = bonus_ids.reshape(-1)
    _need(bonus, torch.int64, "bonus_ids")
    if tuple(draft_probs.shape) != (B, k, V) or tuple(draft_ids.shape) != (B, k) or \
            tuple(u_accept.shape) != (B, k) or tuple(u_recover.shape) != (B, k + 1) or \
            bonus.numel() != B:
        raise _lib.SdError("rejection_sample: inconsistent shapes")
    if proposal_lens is not None:
        _need(proposal_lens, torch.int32, "proposal_lens")
    if counters is not None:
        _need(counters, torch.int64, "counters")
    if out is None:
        out = torch.empty(B, k + 1, dtype=torch.int64, device=target_probs.device)
    if scratch is not None:
        _scratch_ok(scratch, B, "rejection_sample")
        _lib.check(_lib.load().sd_rejection_sample_split(
            _p(target_probs), _p(bonus), _p(draft_probs), _p(draft_ids), _p(u_accept),
            _p(u_recover), _p(proposal_lens), _p(out), _p(counters), B, k, V, _p(scratch),
            _stream()), "sd_rejection_sample_split")
        return out
    _lib.check(_lib.load().sd_rejection_sample(
        _p(target_probs), _p(bonus), _p(draft_probs), _p(draft_ids), _p(u_accept), _p(u_recover),
        _p(proposal_lens), _p(out), _p(counters), B, k, V, _stream()), "sd_rejection_sample")
    return out


def typical_acceptance(target_probs, bonus_ids, draft_ids, posterior_threshold=0.09,
                       posterior_alpha=0.3, proposal_lens=None, counters=None, out=None):
    """Typical acceptance over probability rows [B, k+1, V]; returns [B, k+1] int64."""
    B, k1, V = target_probs.shape
    k = k1 - 1
    _need(target_probs, torch.float32, "target_probs")
    _need(draft_ids, torch.int64, "draft_ids")
    bonus = bonus_ids.reshape(-1)
    _need(bonus, torch.int64, "bonus_ids")
    if tuple(draft_ids.shape) != (B, k) or bonus.numel() != B:
        raise _lib.SdError("typical_acceptance: inconsistent shapes")
    if proposal_lens is not None:
        _need(proposal_lens, torch.int32, "proposal_lens")
    if counters is not None:
        _need(counters, torch.int64, "counters")
    if out is None:
        out = torch.empty(B, k + 1, dtype=torch.int64, device=target_probs.device)
    _lib.check(_lib.load().sd_typical_acceptance(
        _p(target_probs), _p(bonus), _p(draft_ids), _p(proposal_lens), float(posterior_threshold),
        float(posterior_alpha), _p(out), _p(counters), B, k, V, _stream()), "sd_typical_acceptance")
    return out


def greedy_verify(target_ids, draft_ids, proposal_lens=None, counters=None, out=None):
    B, k1 = target_ids.shape
    k = k1 - 1
    _need(target_ids, torch.int64, "target_ids")
    _need(draft_ids, torch.int64, "draft_ids")
    if tuple(draft_ids.shape) != (B, k):
        raise _lib.SdError("greedy_verify: inconsistent shapes")
    if proposal_lens is not None:
        _need(proposal_lens, torch.int32, "proposal_lens")
    if out is None:
        out = torch.empty(B, k + 1, dtype=torch.int64, device=target_ids.device)
    _lib.check(_lib.load().sd_greedy_verify(_p(target_ids), _p(draft_ids), _p(proposal_lens),
                                            _p(out), _p(counters), B, k, _stream()),
               "sd_greedy_verify")
    return out


def argmax_rows(logits: torch.Tensor, out: Optional[torch.Tensor] = None, *, row_offset: int = 0,
                row_step: int = 1) -> torch.Tensor:
    """argmax over the last dim of rows row_offset, row_offset+row_step, ..."""
    _need(logits, torch.float32, "logits")
    R_all, V = logits.shape
    R = (R_all - row_offset + row_step - 1) // row_step
    if out is None:
        out = torch.empty(R, dtype=torch.int64, device=logits.device)
    _lib.check(_lib.load().sd_argmax_rows(logits.data_ptr() + row_offset * V * 4, V * row_step,
                                          _p(out), R, V, _stream()), "sd_argmax_rows")
    return out


def softmax_rows(logits: torch.Tensor, temperature=1.0, out=None, *, row_offset=0,
                 row_step=1, out_ptr=None, out_ld=None, scratch=None,
                 rows_per_seq: int = 1) -> torch.Tensor:
    """softmax(logits / T) of rows row_offset, row_offset+row_step, ...; the
    result goes to `out` ([R, V]) or to raw address out_ptr with row stride out_ld.
    temperature: a number, or a float32 device tensor with one temperature per sequence
    (processed row r belongs to sequence r // rows_per_seq; 0 = all mass on the row's first
    maximum)."""
    _need(logits, torch.float32, "logits")
    R_all, V = logits.shape
    R = (R_all - row_offset + row_step - 1) // row_step
    if out is None and out_ptr is None:
        out = torch.empty(R, V, dtype=torch.float32, device=logits.device)
    dst = out_ptr if out_ptr is not None else out.data_ptr()
    if isinstance(temperature, torch.Tensor):
        _need(temperature, torch.float32, "temperature")
        if rows_per_seq < 1 or temperature.numel() * rows_per_seq < R:
            raise _lib.SdError("softmax_rows: fewer temperatures than sequences")
        if scratch is not None:
            _scratch_ok(scratch, R, "softmax_rows")
        _lib.check(_lib.load().sd_softmax_rows_seq(
            logits.data_ptr() + row_offset * V * 4, V * row_step, _p(temperature), rows_per_seq, dst,
            out_ld or V, R, V, _p(scratch), _stream()), "sd_softmax_rows_seq")
        return out
    if scratch is not None:
        _scratch_ok(scratch, R, "softmax_rows")
        _lib.check(_lib.load().sd_softmax_rows_split(
            logits.data_ptr() + row_offset * V * 4, V * row_step, float(temperature), dst,
            out_ld or V, R, V, _p(scratch), _stream()), "sd_softmax_rows_split")
        return out
    _lib.check(_lib.load().sd_softmax_rows(logits.data_ptr() + row_offset * V * 4, V * row_step,
                                           float(temperature), dst, out_ld or V, R, V, _stream()),
               "sd_softmax_rows")
    return out


def add_logit_bias(logits: torch.Tensor, bias: torch.Tensor, enabled: torch.Tensor,
                   rows_per_seq: int = 1) -> None:
    """logits[r] += bias[r // rows_per_seq] in place for the sequences whose enabled flag is set
    (float32 rows, int32 flags; -inf bans a token; see sd_add_logit_bias).  bias may be a column
    slice of a wider tensor (a rank's share of the vocabulary)."""
    _need(logits, torch.float32, "logits")
    _need(enabled, torch.int32, "enabled")
    if bias.dtype != torch.float32 or bias.dim() != 2 or bias.stride(1) != 1:
        raise _lib.SdError("add_logit_bias: bias must be float32 rows")
    R, V = logits.shape
    n_seq = -(-R // rows_per_seq)
    if bias.shape[1] != V or bias.shape[0] < n_seq or enabled.numel() < n_seq:
        raise _lib.SdError("add_logit_bias: bias / flags do not cover the rows' sequences")
    _lib.check(_lib.load().sd_add_logit_bias(_p(logits), logits.stride(0), bias.data_ptr(), bias.stride(0),
                                             _p(enabled), rows_per_seq, R, V, _stream()),
               "sd_add_logit_bias")


LOGPROBS_TOP_MAX = 32   # SD_LOGPROBS_TOP_MAX


def token_logprobs(logits: torch.Tensor, tokens: torch.Tensor, n_top: int = 0):
    """Log-probabilities of tokens[r] under raw logits rows [R, V] (float32), the n_top most
    likely entries of each row and the chosen token's rank (sd_token_logprobs).  tokens: int32
    [R] on the logits' device; an id outside the vocabulary skips its row.  Returns
    (ids [R, n_top + 1] int32, logprobs [R, n_top + 1] float32, ranks [R] int32)."""
    _need(logits, torch.float32, "logits")
    _need(tokens, torch.int32, "tokens")
    R, V = logits.shape
    if tokens.numel() != R:
        raise _lib.SdError(f"token_logprobs: {tokens.numel()} tokens for {R} rows")
    ids = torch.full((R, n_top + 1), -1, dtype=torch.int32, device=logits.device)
    lps = torch.full((R, n_top + 1), float("-inf"), dtype=torch.float32, device=logits.device)
    ranks = torch.zeros(R, dtype=torch.int32, device=logits.device)
    _lib.check(_lib.load().sd_token_logprobs(_p(logits), logits.stride(0), _p(tokens), R, V, n_top,
                                             _p(ids), _p(lps), _p(ranks), _stream()), "sd_token_logprobs")
    return ids, lps, ranks


def filter_probs(probs: torch.Tensor, top_k=None, top_p=None, min_p=None, rows_per_seq: int = 1) -> None:
    """In-place top-k / min-p / top-p filtering of probability rows [R, V] with per-sequence
    parameters (int32 / float32 / float32 device tensors, row r -> sequence r // rows_per_seq;
    see sd_filter_probs_seq)."""
    _need(probs, torch.float32, "probs")
    R, V = probs.shape
    n_seq = -(-R // rows_per_seq)
    for t, dt, name in ((top_k, torch.int32, "top_k"), (top_p, torch.float32, "top_p"),
                        (min_p, torch.float32, "min_p")):
        if t is not None:
            _need(t, dt, name)
            if t.numel() < n_seq:
                raise _lib.SdError(f"filter_probs: {name} shorter than the number of sequences")
    _lib.check(_lib.load().sd_filter_probs_seq(_p(probs), probs.stride(0), _p(top_k), _p(top_p), _p(min_p),
                                               rows_per_seq, R, V, _stream()), "sd_filter_probs_seq")


def sample_rows(probs: torch.Tensor, uniforms: torch.Tensor, out=None, *, row_offset=0,
                row_step=1, scratch=None) -> torch.Tensor:
    """Inverse-CDF draw from rows row_offset, row_offset+row_step, ... of probs [*, V]."""
    _need(probs, torch.float32, "probs")
    _need(uniforms, torch.float32, "uniforms")
    R_all, V = probs.shape
    R = (R_all - row_offset + row_step - 1) // row_step
    if uniforms.numel() < R:
        raise _lib.SdError("sample_rows: need one uniform per row")
    if out is None:
        out = torch.empty(R, dtype=torch.int64, device=probs.device)
    if scratch is not None:
        _scratch_ok(scratch, R, "sample_rows")
        _lib.check(_lib.load().sd_sample_rows_split(
            probs.data_ptr() + row_offset * V * 4, V * row_step, _p(uniforms), _p(out), R, V,
            _p(scratch), _stream()), "sd_sample_rows_split")
        return out
    _lib.check(_lib.load().sd_sample_rows(probs.data_ptr() + row_offset * V * 4, V * row_step,
                                          _p(uniforms), _p(out), R, V, _stream()),
               "sd_sample_rows")
    return out


# ---------------------------------------------------------------- attention
def rope_append(qkv, rope_cs, k_cache, v_cache, block_tables, seq_lens, pos_offset, B, q_len,
                Hq, Hkv, D) -> None:
    _need(qkv, torch.bfloat16, "qkv")
    _need(k_cache, torch.bfloat16, "k_cache")
    _need(v_cache, torch.bfloat16, "v_cache")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    max_pos = 0
    if rope_cs is not None:
        _need(rope_cs, torch.float32, "rope_cs")
        max_pos = rope_cs.shape[0]
    _lib.check(_lib.load().sd_rope_append(
        _p(qkv), _p(rope_cs), max_pos, _p(k_cache), _p(v_cache), _p(block_tables),
        block_tables.shape[1], _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D, _stream()),
        "sd_rope_append")


def paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split_tokens) -> int:
    return int(_lib.load().sd_paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx,
                                                              split_tokens))


def paged_attention(q, k_cache, v_cache, block_tables, seq_lens, pos_offset, B, q_len, Hq, Hkv, D,
                    scale, max_ctx, split_tokens=256, ldq=None, workspace=None,
                    partials_only=False) -> Optional[torch.Tensor]:
    """q: [B*q_len, >= Hq*D] bf16 rows (row stride ldq elements).  partials_only: leave the
    per-split partials in `workspace` for linear_attn_merge and return None."""
    _need(q, torch.bfloat16, "q")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    if ldq is None:
        ldq = q.shape[-1] if q.dim() == 2 else Hq * D
    out = None if partials_only else torch.empty(B * q_len, Hq * D, dtype=torch.bfloat16,
                                                 device=q.device)
    need = paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split_tokens)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    _lib.check(_lib.load().sd_paged_attention(
        _p(q), ldq, _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1],
        _p(seq_lens), pos_offset, _p(out), Hq * D, B, q_len, Hq, Hkv, D, float(scale), max_ctx,
        split_tokens, _p(workspace), workspace.numel(), _stream()), "sd_paged_attention")
    return out


STATIC_SCRATCH_BYTES = 16 << 20


def paged_attention_static(q, k_cache, v_cache, block_tables, seq_lens, pos_offset, B, q_len, Hq,
                           Hkv, D, scale, max_ctx, split_tokens, out: torch.Tensor,
                           ldq=None) -> torch.Tensor:
    """paged_attention without a workspace and without any allocation: the split partials
    live in an array inside the library; `out` ([B*q_len, Hq*D] bf16) is the caller's.  Calls
    that overlap on one GPU race (they share the array)."""
    _need(q, torch.bfloat16, "q")
    _need(out, torch.bfloat16, "out")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    if out.numel() < B * q_len * Hq * D:
        raise _lib.SdError("paged_attention_static: out too small")
    if ldq is None:
        ldq = q.shape[-1] if q.dim() == 2 else Hq * D
    _lib.check(_lib.load().sd_paged_attention_static(
        _p(q), ldq, _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1],
        _p(seq_lens), pos_offset, _p(out), Hq * D, B, q_len, Hq, Hkv, D, float(scale), max_ctx,
        split_tokens, _stream()), "sd_paged_attention_static")
    return out


def paged_attention_tree(q, k_cache, v_cache, block_tables, seq_lens, pos_offset, tree_mask, B, q_len,
                         Hq, Hkv, D, scale, max_ctx, split_tokens=256, workspace=None) -> torch.Tensor:
    """paged_attention over a token tree: tree_mask [B*q_len] int64 (bit j of entry b*q_len+t:
    new token t attends new token j)."""
    _need(q, torch.bfloat16, "q")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    _need(tree_mask, torch.int64, "tree_mask")
    if tree_mask.numel() < B * q_len:
        raise _lib.SdError("paged_attention_tree: tree_mask needs B*q_len entries")
    out = torch.empty(B * q_len, Hq * D, dtype=torch.bfloat16, device=q.device)
    need = paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split_tokens)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    _lib.check(_lib.load().sd_paged_attention_tree(
        _p(q), q.shape[-1], _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1],
        _p(seq_lens), pos_offset, _p(tree_mask), _p(out), Hq * D, B, q_len, Hq, Hkv, D, float(scale),
        max_ctx, split_tokens, _p(workspace), workspace.numel(), _stream()),
        "sd_paged_attention_tree")
    return out


def map_tokens(tokens: torch.Tensor, col: int, token_map: torch.Tensor) -> None:
    """tokens[:, col] = token_map[tokens[:, col]] (int32 [B, stride]; see sd_map_tokens)."""
    _need(tokens, torch.int32, "tokens")
    _need(token_map, torch.int32, "token_map")
    B, stride = tokens.shape
    _lib.check(_lib.load().sd_map_tokens(_p(tokens), stride, col, _p(token_map), token_map.numel(), B,
                                         _stream()), "sd_map_tokens")


def tree_prepare(tok_buf, tok_col, tree_tok, tree_depth, seq_lens, pos_offset, rope_pos) -> None:
    """tree_tok[:, 0] = tok_buf[:, tok_col]; rope_pos[b, t] = seq_lens[b] + pos_offset + depth[b, t]."""
    for t, n in ((tok_buf, "tok_buf"), (tree_tok, "tree_tok"), (tree_depth, "tree_depth"),
                 (seq_lens, "seq_lens"), (rope_pos, "rope_pos")):
        _need(t, torch.int32, n)
    B, q_len = tree_tok.shape
    if tuple(tree_depth.shape) != (B, q_len) or rope_pos.numel() < B * q_len or tok_buf.shape[0] < B:
        raise _lib.SdError("tree_prepare: inconsistent shapes")
    _lib.check(_lib.load().sd_tree_prepare(_p(tok_buf), tok_buf.shape[1], tok_col, _p(tree_tok),
                                           _p(tree_depth), _p(seq_lens), pos_offset, _p(rope_pos), B,
                                           q_len, _stream()), "sd_tree_prepare")


TOPK_TOKENS_MAX = 8


def topk_tokens(logits, b: int, out_tok, out_col: int, *, row_offset: int = 0, row_step: int = 1,
                rows: int = 1, B: Optional[int] = None) -> None:
    """out_tok[s, out_col + r*b + c] = the c-th largest entry of logits row row_offset + s*row_step + r
    as a token (largest first; ties: lowest index), r < rows, c < b <= TOPK_TOKENS_MAX."""
    _need(logits, torch.float32, "logits")
    _need(out_tok, torch.int32, "out_tok")
    n = out_tok.shape[0] if B is None else B
    _lib.check(_lib.load().sd_topk_tokens(_p(logits), logits.stride(0), logits.shape[1], row_offset,
                                          row_step, rows, b, _p(out_tok), out_tok.stride(0), out_col, n,
                                          _stream()), "sd_topk_tokens")


def tree_accept_greedy(target_ids, tree_tok, tree_parent, tree_len, k, out_ids, path, n_path,
                       counters=None) -> None:
    """Greedy acceptance walk over a token tree (see sd_tree_accept_greedy)."""
    _need(target_ids, torch.int64, "target_ids")
    for t, n in ((tree_tok, "tree_tok"), (tree_parent, "tree_parent"), (tree_len, "tree_len"),
                 (path, "path"), (n_path, "n_path")):
        _need(t, torch.int32, n)
    _need(out_ids, torch.int64, "out_ids")
    B, q_len = tree_tok.shape
    if (target_ids.numel() < B * q_len or tuple(tree_parent.shape) != (B, q_len)
            or tuple(out_ids.shape) != (B, k + 1) or tuple(path.shape) != (B, k)
            or tree_len.numel() < B or n_path.numel() < B):
        raise _lib.SdError("tree_accept_greedy: inconsistent shapes")
    _lib.check(_lib.load().sd_tree_accept_greedy(_p(target_ids), _p(tree_tok), _p(tree_parent),
                                                 _p(tree_len), q_len, k, _p(out_ids), _p(path),
                                                 _p(n_path), _p(counters), B, _stream()),
               "sd_tree_accept_greedy")


def tree_accept_sample(target_probs, tree_tok, tree_parent, tree_len, k, u_accept, u_final, out_ids,
                       path, n_path, counters=None) -> None:
    """Acceptance walk over a token tree under sampling, for candidates without a draft
    distribution (see sd_tree_accept_sample).  target_probs [B * q_len, V] float32."""
    _need(target_probs, torch.float32, "target_probs")
    for t, n in ((tree_tok, "tree_tok"), (tree_parent, "tree_parent"), (tree_len, "tree_len"),
                 (path, "path"), (n_path, "n_path")):
        _need(t, torch.int32, n)
    _need(u_accept, torch.float32, "u_accept")
    _need(u_final, torch.float32, "u_final")
    _need(out_ids, torch.int64, "out_ids")
    B, q_len = tree_tok.shape
    if (target_probs.dim() != 2 or target_probs.shape[0] < B * q_len
            or tuple(tree_parent.shape) != (B, q_len) or tuple(u_accept.shape) != (B, q_len)
            or u_final.numel() < B or tuple(out_ids.shape) != (B, k + 1) or tuple(path.shape) != (B, k)
            or tree_len.numel() < B or n_path.numel() < B):
        raise _lib.SdError("tree_accept_sample: inconsistent shapes")
    _lib.check(_lib.load().sd_tree_accept_sample(
        _p(target_probs), target_probs.stride(0), _p(tree_tok), _p(tree_parent), _p(tree_len), q_len, k,
        _p(u_accept), _p(u_final), _p(out_ids), _p(path), _p(n_path), _p(counters), B,
        target_probs.shape[1], _stream()), "sd_tree_accept_sample")


def kv_compact(k_cache, v_cache, layer_stride_bytes, n_layers, block_tables, seq_lens, pos_offset,
               path, n_path, Hkv, D) -> None:
    """Moves the accepted tree rows' K/V to their chain slots (see sd_kv_compact)."""
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    _need(path, torch.int32, "path")
    _need(n_path, torch.int32, "n_path")
    B, k = path.shape
    _lib.check(_lib.load().sd_kv_compact(_p(k_cache), _p(v_cache), layer_stride_bytes, n_layers,
                                         _p(block_tables), block_tables.shape[1], _p(seq_lens),
                                         pos_offset, _p(path), _p(n_path), k, B, Hkv, D, _stream()),
               "sd_kv_compact")


def paged_attention_blocked(q, k_cache, v_cache, block_tables, seq_lens, pos_offset, B, q_total,
                            q_block, Hq, Hkv, D, scale, max_ctx, split_tokens=256,
                            workspace=None) -> torch.Tensor:
    """Attention of q_total new tokens per sequence in blocks of q_block tokens, one launch."""
    _need(q, torch.bfloat16, "q")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    out = torch.empty(B * q_total, Hq * D, dtype=torch.bfloat16, device=q.device)
    n_blocks = -(-q_total // q_block)
    need = paged_attention_workspace_bytes(B * n_blocks, q_block, Hq, D, max_ctx, split_tokens)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    _lib.check(_lib.load().sd_paged_attention_blocked(
        _p(q), q.shape[-1], _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1],
        _p(seq_lens), pos_offset, _p(out), Hq * D, B, q_total, q_block, Hq, Hkv, D, float(scale),
        max_ctx, split_tokens, _p(workspace), workspace.numel(), _stream()),
        "sd_paged_attention_blocked")
    return out


# ---------------------------------------------------------------- dense ops
TILE_PLAIN, TILE_GATED, TILE_ROTARY = 0, 1, 2


class TiledWeight:
    """A weight matrix re-laid-out for the kernels (see include/sd_amd.h):
    `data` is the flat tiled buffer (bf16, or uint8 holding fp8-e4m3 bytes), N the number of
    OUTPUT columns of the linear (F for a gated [2F, K] matrix), K the input width; `scale`
    (fp8 only) is the f32 scale per source row."""

    __slots__ = ("data", "N", "K", "mode", "scale")

    def __init__(self, data: torch.Tensor, N: int, K: int, mode: int,
                 scale: Optional[torch.Tensor] = None):
        self.data, self.N, self.K, self.mode, self.scale = data, N, K, mode, scale

    def nbytes(self) -> int:
        return self.data.numel() * self.data.element_size() + (
            0 if self.scale is None else self.scale.numel() * 4)


FP8_MAX = 448.0  # largest finite e4m3 value


def quantize_fp8(W: torch.Tensor):
    """Row-major [rows, K] -> (fp8-e4m3 [rows, K], f32 scale [rows]): per-row absmax scaling,
    W[r] ~ q[r] * scale[r]."""
    Wf = W.float()
    scale = (Wf.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX).to(torch.float32)
    q = (Wf / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.contiguous(), scale.contiguous()


def tile_weights(W: torch.Tensor, mode: int = TILE_PLAIN, head_dim: int = 0,
                 scale: Optional[torch.Tensor] = None) -> TiledWeight:
    """Row-major [rows, K] bf16 -> TiledWeight (one pass on the GPU, at load time).  An
    fp8-e4m3 W with its per-row f32 `scale` gives the fp8 tiling."""
    rows, K = W.shape
    N = rows // 2 if mode == TILE_GATED else rows
    lib = _lib.load()
    if W.dtype == torch.float8_e4m3fn:
        if scale is None or scale.dtype != torch.float32 or scale.numel() != rows:
            raise _lib.SdError("tile_weights: fp8 weights need an f32 scale per row")
        n = int(lib.sd_tiled_weight_elems(N, K, mode)) if K % 64 == 0 else 0
        if n == 0:
            raise _lib.SdError(f"tile_weights: unsupported fp8 shape {tuple(W.shape)} (K must be a multiple of 64)")
        src = W.contiguous().view(torch.uint8)
        dst = torch.empty(n, dtype=torch.uint8, device=W.device)
        _lib.check(lib.sd_tile_weights_fp8(_p(src), _p(dst), N, K, mode, head_dim, _stream()),
                   "sd_tile_weights_fp8")
        return TiledWeight(dst, N, K, mode, scale.contiguous())
    _need(W, torch.bfloat16, "W")
    n = int(lib.sd_tiled_weight_elems(N, K, mode))
    if n == 0:
        raise _lib.SdError(f"tile_weights: unsupported shape {tuple(W.shape)} (K must be a multiple of 32)")
    dst = torch.empty(n, dtype=torch.bfloat16, device=W.device)
    _lib.check(lib.sd_tile_weights(_p(W), _p(dst), N, K, mode, head_dim, _stream()),
               "sd_tile_weights")
    return TiledWeight(dst, N, K, mode)


def _tiled(W, epilogue, head_dim=0) -> TiledWeight:
    if isinstance(W, TiledWeight):
        return W
    mode = TILE_GATED if epilogue == EPI_SILU_MUL else (TILE_ROTARY if head_dim else TILE_PLAIN)
    return tile_weights(W, mode, head_dim)


def linear(W, x, bias=None, residual=None, epilogue=EPI_STORE, out=None) -> torch.Tensor:
    """W: TiledWeight (or a row-major tensor, tiled on the fly — tests only)."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, epilogue)
    M, K = x.shape
    N = Wt.N
    if Wt.K != K or (Wt.mode == TILE_GATED) != (epilogue == EPI_SILU_MUL):
        raise _lib.SdError(f"linear: weight (N={Wt.N}, K={Wt.K}, mode={Wt.mode}) does not fit "
                           f"x {tuple(x.shape)} / epilogue {epilogue}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if epilogue == EPI_F32 else torch.bfloat16,
                          device=x.device)
    _lib.check(_lib.load().sd_linear(_p(Wt.data), _p(Wt.scale), _p(x), K, _p(bias), _p(residual), _p(out), N, M, N,
                                     K, epilogue, _stream()), "sd_linear")
    return out


def rmsnorm(x, w, eps) -> torch.Tensor:
    _need(x, torch.bfloat16, "x")
    _need(w, torch.bfloat16, "w")
    rows, H = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().sd_rmsnorm(_p(x), H, _p(w), _p(out), H, rows, H, float(eps), _stream()),
               "sd_rmsnorm")
    return out


def layernorm(x, w, b, eps) -> torch.Tensor:
    _need(x, torch.bfloat16, "x")
    rows, H = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().sd_layernorm(_p(x), H, _p(w), _p(b), _p(out), H, rows, H, float(eps),
                                        _stream()), "sd_layernorm")
    return out


def row_scales_fp8(x: torch.Tensor) -> torch.Tensor:
    """f32 [2, M]: every row's fp8 scale max|x[m]| / 448 and its inverse factor (sd_row_scales_fp8)."""
    _need(x, torch.bfloat16, "x")
    M, K = x.shape
    out = torch.empty(2, M, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sd_row_scales_fp8(_p(x), x.stride(0), M, K, _p(out), _stream()),
               "sd_row_scales_fp8")
    return out


def linear_wide(W, x, bias=None, residual=None, epilogue=EPI_STORE, out=None,
                workspace=None, fp8_activations: bool = False) -> torch.Tensor:
    """linear() for up to 256 token rows (its own summation order; see sd_linear_wide).
    fp8_activations (fp8 weights only): the rows are quantised to e4m3 with a scale per row and
    multiplied on the fp8 matrix instruction (sd_linear_wide_a8)."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, epilogue)
    M, K = x.shape
    N = Wt.N
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if epilogue == EPI_F32 else torch.bfloat16,
                          device=x.device)
    need = int(_lib.load().sd_linear_wide_workspace_bytes(M, N, K, epilogue))
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    if fp8_activations:
        sc = row_scales_fp8(x)
        _lib.check(_lib.load().sd_linear_wide_a8(_p(Wt.data), _p(Wt.scale), _p(sc), _p(x), x.stride(0),
                                                 _p(bias), _p(residual), _p(out), out.stride(0), M, N, K,
                                                 epilogue, _p(workspace), workspace.numel(), _stream()),
                   "sd_linear_wide_a8")
        return out
    _lib.check(_lib.load().sd_linear_wide(_p(Wt.data), _p(Wt.scale), _p(x), x.stride(0), _p(bias),
                                          _p(residual),
                                          _p(out), out.stride(0), M, N, K, epilogue, _p(workspace),
                                          workspace.numel(), _stream()), "sd_linear_wide")
    return out


def _rope_positions(rope_pos, rope_delta, mrope_section, rows):
    """ctypes sd_rope_positions (or None): rope_pos int32 [3, rows], rope_delta int32 [B],
    mrope_section = (temporal, height, width) frequency counts."""
    if rope_pos is None and rope_delta is None:
        return None
    if mrope_section is None:
        raise _lib.SdError("rotary positions need mrope_section")
    if rope_pos is not None:
        _need(rope_pos, torch.int32, "rope_pos")
        if tuple(rope_pos.shape) != (3, rows) or not rope_pos.is_contiguous():
            raise _lib.SdError(f"rope_pos must be a contiguous int32 [3, {rows}] tensor")
    if rope_delta is not None:
        _need(rope_delta, torch.int32, "rope_delta")
    import ctypes as C
    return C.byref(_lib.RopePositions(pos=_p(rope_pos), delta=_p(rope_delta),
                                      section_t=int(mrope_section[0]), section_h=int(mrope_section[1])))


def qkv_rope_wide(W, x, bias, rope_cs, k_cache, v_cache, block_tables, seq_lens, pos_offset, B,
                  q_len, Hq, Hkv, D, workspace=None, rope_pos=None, rope_delta=None,
                  mrope_section=None, fp8_activations: bool = False) -> torch.Tensor:
    """QKV projection + rotary embedding + cache append for up to 256 token rows; returns q.
    fp8_activations: as in linear_wide (sd_qkv_rope_wide_a8)."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, EPI_STORE, D)
    M, K = x.shape
    N = (Hq + 2 * Hkv) * D
    q = torch.empty(M, Hq * D, dtype=torch.bfloat16, device=x.device)
    need = int(_lib.load().sd_linear_wide_workspace_bytes(M, N, K, EPI_STORE))
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    if fp8_activations:
        sc = row_scales_fp8(x)
        _lib.check(_lib.load().sd_qkv_rope_wide_a8(
            _p(Wt.data), _p(Wt.scale), _p(sc), _p(x), x.stride(0), _p(bias), _p(q), Hq * D, _p(rope_cs),
            0 if rope_cs is None else rope_cs.shape[0], _p(k_cache), _p(v_cache), _p(block_tables),
            block_tables.shape[1], _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D, K,
            _rope_positions(rope_pos, rope_delta, mrope_section, M), _p(workspace),
            workspace.numel(), _stream()), "sd_qkv_rope_wide_a8")
        return q
    _lib.check(_lib.load().sd_qkv_rope_wide(
        _p(Wt.data), _p(Wt.scale), _p(x), x.stride(0), _p(bias), _p(q), Hq * D, _p(rope_cs),
        0 if rope_cs is None else rope_cs.shape[0], _p(k_cache), _p(v_cache), _p(block_tables),
        block_tables.shape[1], _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D, K,
        _rope_positions(rope_pos, rope_delta, mrope_section, M), _p(workspace),
        workspace.numel(), _stream()), "sd_qkv_rope_wide")
    return q


def linear_attn_merge(W, workspace, seq_lens, pos_offset, B, q_len, Hq, Hkv, D, max_ctx,
                      split_tokens, bias=None, residual=None, epilogue=EPI_STORE,
                      out=None) -> torch.Tensor:
    """W @ merge(attention partials in `workspace`), see paged_attention(partials_only=True)."""
    Wt = _tiled(W, epilogue)
    if out is None:
        out = torch.empty(B * q_len, Wt.N, dtype=torch.bfloat16, device=workspace.device)
    _lib.check(_lib.load().sd_linear_attn_merge(
        _p(Wt.data), _p(Wt.scale), _p(workspace), workspace.numel(), _p(seq_lens), pos_offset, B,
        q_len, Hq, Hkv,
        D, max_ctx, split_tokens, _p(bias), _p(residual), _p(out), out.stride(0), Wt.N, epilogue,
        _stream()), "sd_linear_attn_merge")
    return out


def fill_uniforms(out: torch.Tensor, seed: int, counter: torch.Tensor) -> None:
    """out (f32, contiguous) <- uniforms in [0, 1) from (seed, counter[0]); counter[0] += 1."""
    _need(out, torch.float32, "out")
    _need(counter, torch.int64, "counter")
    _lib.check(_lib.load().sd_fill_uniforms(_p(out), out.numel(), int(seed) & (2 ** 64 - 1),
                                            _p(counter), _stream()), "sd_fill_uniforms")


def uniform_regions(regions):
    """ctypes array for fill_uniforms_seq from (offset, n, stride_j, stride_b) tuples."""
    arr = (_lib.UniformRegion * len(regions))()
    for i, (off, n, sj, sb) in enumerate(regions):
        arr[i] = _lib.UniformRegion(off, n, sj, sb)
    return arr


def fill_uniforms_seq(out: torch.Tensor, regions, seeds: torch.Tensor, counters: torch.Tensor,
                      B: int) -> None:
    """Per-sequence uniform streams: sequence b's j-th number of the step is
    hash(seeds[b], counters[b], j), written into out at the regions' places
    (ops.uniform_regions); counters[:B] += 1."""
    _need(out, torch.float32, "out")
    _need(seeds, torch.int64, "seeds")
    _need(counters, torch.int64, "counters")
    if seeds.numel() < B or counters.numel() < B:
        raise _lib.SdError("fill_uniforms_seq: seeds / counters shorter than B")
    _lib.check(_lib.load().sd_fill_uniforms_seq(_p(out), regions, len(regions), _p(seeds),
                                                _p(counters), B, _stream()), "sd_fill_uniforms_seq")


def uniform24(seed: int, counter: int, j: int) -> float:
    """The number fill_uniforms_seq gives sequence (seed, counter) at index j, on the host."""
    m = (1 << 64) - 1
    z = (seed ^ ((counter * 0x9e3779b97f4a7c15) & m) ^ (((j << 32) | (j & 0xffffffff)) & m)) & m
    z = (z + 0x9e3779b97f4a7c15) & m
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & m
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & m
    z ^= z >> 31
    return (z >> 40) / 16777216.0


# ---------------------------------------------------------------- step glue
def store_proposal(ids, tok_buf, k, step) -> None:
    B = tok_buf.shape[0]
    _lib.check(_lib.load().sd_store_proposal(_p(ids), _p(tok_buf), k, step, B, _stream()),
               "sd_store_proposal")


DEBUG_PATTERN_MAX = 64


def debug_force_proposals(tok_buf, k, seq_lens, truth, pattern, step_counter, vocab) -> None:
    """Benchmark control: overwrite the proposals in tok_buf so that pattern[step_counter[b] %
    len(pattern)] of them are accepted (see sd_debug_force_proposals)."""
    _need(truth, torch.int32, "truth")
    _need(step_counter, torch.int32, "step_counter")
    B = tok_buf.shape[0]
    pat = bytes(int(a) for a in pattern)
    _lib.check(_lib.load().sd_debug_force_proposals(
        _p(tok_buf), k, _p(seq_lens), _p(truth), truth.shape[1], pat, len(pat), _p(step_counter),
        int(vocab), B, _stream()), "sd_debug_force_proposals")


def set_proposals(proposals, tok_buf, k) -> None:
    """proposals [B, k] int32 -> tok_buf[:, 2:2+k]."""
    _need(proposals, torch.int32, "proposals")
    _lib.check(_lib.load().sd_set_proposals(_p(proposals), _p(tok_buf), tok_buf.shape[0], k,
                                            _stream()), "sd_set_proposals")


def gather_proposals(tok_buf, draft_ids, k) -> None:
    _lib.check(_lib.load().sd_gather_proposals(_p(tok_buf), _p(draft_ids), tok_buf.shape[0], k,
                                               _stream()), "sd_gather_proposals")


def advance_state(out_ids, seq_lens, tok_buf, gen_tokens, gen_counts, k, step_out=None) -> None:
    """step_out: optional int32 [B, k+3] record {n_emitted, seq_len, tokens}; may be a
    pinned host tensor."""
    B = seq_lens.shape[0]
    _lib.check(_lib.load().sd_advance_state(
        _p(out_ids), _p(seq_lens), _p(tok_buf), _p(gen_tokens),
        0 if gen_tokens is None else gen_tokens.shape[1],
        0 if gen_tokens is None else gen_tokens.shape[1], _p(gen_counts), _p(step_out), B, k,
        _stream()), "sd_advance_state")


ARGMAX_PARTS = 512


def argmax_partial(logits, part_val, part_idx) -> None:
    """Stage 1 of the split row argmax over logits [R, V] f32."""
    _need(logits, torch.float32, "logits")
    R, V = logits.shape
    if part_val.numel() < R * ARGMAX_PARTS or part_idx.numel() < R * ARGMAX_PARTS:
        raise _lib.SdError("argmax_partial: partial buffers too small")
    _lib.check(_lib.load().sd_argmax_partial(_p(logits), logits.stride(0), _p(part_val),
                                             _p(part_idx), R, V, _stream()), "sd_argmax_partial")


def argmax_combine(gathered: torch.Tensor, n_ranks: int, R: int, shard_vocab: int,
                   out_val: torch.Tensor, out_idx: torch.Tensor) -> None:
    """gathered: n_ranks blocks of {[R, ARGMAX_PARTS] f32 maxima, [R, ARGMAX_PARTS] i32 local
    indices} (each rank's argmax_partial output, packed values-then-indices, in rank order) ->
    maxima over the whole vocabulary with global indices in out_val / out_idx."""
    _need(gathered, torch.float32, "gathered")
    n = R * ARGMAX_PARTS
    if gathered.numel() < n_ranks * 2 * n or out_val.numel() < n or out_idx.numel() < n:
        raise _lib.SdError("argmax_combine: buffers too small")
    _need(out_val, torch.float32, "out_val")
    _need(out_idx, torch.int32, "out_idx")
    base = gathered.data_ptr()
    _lib.check(_lib.load().sd_argmax_combine(base, base + 4 * n, n_ranks, R, shard_vocab, 2 * n,
                                             _p(out_val), _p(out_idx), _stream()),
               "sd_argmax_combine")


def argmax_from_partials(part_val, part_idx, R: int, out=None) -> torch.Tensor:
    if out is None:
        out = torch.empty(R, dtype=torch.int64, device=part_val.device)
    _need(out, torch.int64, "out")
    if out.numel() < R:
        raise _lib.SdError("argmax_from_partials: out too small")
    _lib.check(_lib.load().sd_argmax_from_partials(_p(part_val), _p(part_idx), _p(out), R,
                                                   _stream()), "sd_argmax_from_partials")
    return out


def propose_from_partials(part_val, part_idx, row_offset, row_step, tok_buf, k, step) -> None:
    B = tok_buf.shape[0]
    _lib.check(_lib.load().sd_propose_from_partials(
        _p(part_val), _p(part_idx), row_offset, row_step, _p(tok_buf), k, step, B, _stream()),
        "sd_propose_from_partials")


def verify_from_partials(part_val, part_idx, tok_buf, out_ids, counters, seq_lens, gen_tokens,
                         gen_counts, k, step_out=None, proposal_lens=None) -> None:
    B = seq_lens.shape[0]
    _lib.check(_lib.load().sd_verify_from_partials(
        _p(part_val), _p(part_idx), _p(tok_buf), _p(proposal_lens), _p(out_ids), _p(counters),
        _p(seq_lens), _p(gen_tokens), 0 if gen_tokens is None else gen_tokens.shape[1],
        0 if gen_tokens is None else gen_tokens.shape[1], _p(gen_counts), _p(step_out), B, k,
        _stream()), "sd_verify_from_partials")


def embed_tokens(ids: torch.Tensor, table: torch.Tensor, tiled: bool, hidden: int,
                 vocab: int) -> torch.Tensor:
    """Rows of the token embedding table ([vocab, hidden] bf16, row-major or in the PLAIN tile
    layout of a tied LM head) for int32 ids [n]; returns [n, hidden] bf16."""
    _need(ids, torch.int32, "ids")
    n = ids.numel()
    out = torch.empty(n, hidden, dtype=torch.bfloat16, device=ids.device)
    _lib.check(_lib.load().sd_embed(_p(ids), 1, 0, _p(table), int(tiled), None, 0, None, 0, _p(out),
                                    n, 1, hidden, vocab, _stream()), "sd_embed")
    return out


def norm_linear(W, x, norm_w, norm_b=None, eps=1e-5, layer_norm=False, bias=None,
                epilogue=EPI_STORE, out=None) -> torch.Tensor:
    """linear(norm(x)) in one kernel (RMSNorm, or LayerNorm when layer_norm)."""
    _need(x, torch.bfloat16, "x")
    _need(norm_w, torch.bfloat16, "norm_w")
    Wt = _tiled(W, epilogue)
    M, K = x.shape
    N = Wt.N
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if epilogue == EPI_F32 else torch.bfloat16,
                          device=x.device)
    _lib.check(_lib.load().sd_norm_linear(_p(Wt.data), _p(Wt.scale), _p(x), K, _p(norm_w), _p(norm_b), float(eps),
                                          int(layer_norm), _p(bias), _p(out), N, M, N, K, epilogue,
                                          _stream()), "sd_norm_linear")
    return out


def norm_linear_argmax(W, x, norm_w, norm_b=None, eps=1e-5, layer_norm=False, part_val=None,
                       part_idx=None):
    """The LM head of a greedy pass without logits: the slice maxima (part_val f32, part_idx i32,
    [M, ARGMAX_PARTS]) of norm(x) @ W^T, for argmax_from_partials / propose_from_partials /
    verify_from_partials (sd_norm_linear_argmax)."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, EPI_F32)
    M, K = x.shape
    if part_val is None:
        part_val = torch.empty(M, ARGMAX_PARTS, dtype=torch.float32, device=x.device)
        part_idx = torch.empty(M, ARGMAX_PARTS, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().sd_norm_linear_argmax(_p(Wt.data), _p(Wt.scale), _p(x), x.stride(0), _p(norm_w),
                                                 _p(norm_b), float(eps), int(layer_norm), _p(part_val),
                                                 _p(part_idx), M, Wt.N, K, _stream()),
               "sd_norm_linear_argmax")
    return part_val, part_idx


def norm_qkv_rope(W, x, norm_w, norm_b, eps, layer_norm, bias, rope_cs, k_cache, v_cache,
                  block_tables, seq_lens, pos_offset, B, q_len, Hq, Hkv, D, rope_pos=None,
                  rope_delta=None, mrope_section=None) -> torch.Tensor:
    """norm -> QKV projection -> rotary -> cache append; returns rotated q [B*q_len, Hq*D]."""
    _need(x, torch.bfloat16, "x")
    _need(block_tables, torch.int32, "block_tables")
    _need(seq_lens, torch.int32, "seq_lens")
    Wt = _tiled(W, EPI_STORE, head_dim=D)
    if Wt.mode != TILE_ROTARY:
        raise _lib.SdError("norm_qkv_rope: the QKV weight must be tiled in TILE_ROTARY mode")
    M, K = x.shape
    q = torch.empty(M, Hq * D, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().sd_norm_qkv_rope(
        _p(Wt.data), _p(Wt.scale), _p(x), K, _p(norm_w), _p(norm_b), float(eps), int(layer_norm),
        _p(bias), _p(q),
        Hq * D, _p(rope_cs), 0 if rope_cs is None else rope_cs.shape[0], _p(k_cache), _p(v_cache),
        _p(block_tables), block_tables.shape[1], _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D, K,
        _rope_positions(rope_pos, rope_delta, mrope_section, M), _stream()), "sd_norm_qkv_rope")
    return q


# ---------------------------------------------------------------- the draft model's three-launch layer
DRAFT_LAYER_MAX_ROWS = 4
DRAFT_LAYER_MAX_CTX = 2048


def draft_layer_supported(rows: int, q_len: int, hidden: int, n_heads: int, n_kv_heads: int,
                          head_dim: int, ffn: int, max_ctx: int) -> bool:
    return bool(_lib.load().sd_draft_layer_supported(rows, q_len, hidden, n_heads, n_kv_heads,
                                                     head_dim, ffn, max_ctx))


def kmajor_tiles(W: "TiledWeight") -> torch.Tensor:
    """The SD_TILE_PLAIN tiles of a bf16 weight ordered k-step first ([K/32][N/16][64][8]): the
    32 input columns a workgroup of sd_draft_mlp owns are then contiguous."""
    if W.mode != TILE_PLAIN or W.scale is not None or W.data.dtype != torch.bfloat16:
        raise _lib.SdError("kmajor_tiles: a bf16 weight tiled in TILE_PLAIN mode expected")
    n_tiles, ks = (W.N + 15) // 16, W.K // 32
    return W.data.view(n_tiles, ks, 512).transpose(0, 1).contiguous().view(-1)


def norm_qkv_rope_adds(W, x, norm_w, eps, rope_cs, k_cache, v_cache, block_tables, seq_lens, pos_offset,
                       B, q_len, Hq, Hkv, D, add=None, x_out=None) -> torch.Tensor:
    """sd_norm_qkv_rope_adds: RMSNorm of bf16(x + add) -> QKV -> rotary -> cache append; x_out
    (bf16 like x) receives x + add.  Returns rotated q."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, EPI_STORE, head_dim=D)
    M, K = x.shape
    q = torch.empty(M, Hq * D, dtype=torch.bfloat16, device=x.device)
    adds = None
    if add is not None:
        _need(add, torch.float32, "add")
        adds = C.byref(_lib.RowAdds(add=_p(add), ld_add=add.shape[-1], x_out=_p(x_out)))
    _lib.check(_lib.load().sd_norm_qkv_rope_adds(
        _p(Wt.data), _p(Wt.scale), _p(x), K, _p(norm_w), None, float(eps), 0, None, _p(q), Hq * D,
        _p(rope_cs), 0 if rope_cs is None else rope_cs.shape[0], _p(k_cache), _p(v_cache),
        _p(block_tables), block_tables.shape[1], _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D, K,
        None, adds, _stream()), "sd_norm_qkv_rope_adds")
    return q


def norm_linear_adds(W, x, norm_w, eps, add=None, x_out=None) -> torch.Tensor:
    """sd_norm_linear_adds with the F32 epilogue (an LM head): logits of RMSNorm(bf16(x + add))."""
    _need(x, torch.bfloat16, "x")
    Wt = _tiled(W, EPI_F32)
    M, K = x.shape
    out = torch.empty(M, Wt.N, dtype=torch.float32, device=x.device)
    adds = None
    if add is not None:
        _need(add, torch.float32, "add")
        adds = C.byref(_lib.RowAdds(add=_p(add), ld_add=add.shape[-1], x_out=_p(x_out)))
    _lib.check(_lib.load().sd_norm_linear_adds(
        _p(Wt.data), _p(Wt.scale), _p(x), K, _p(norm_w), None, float(eps), 0, None, _p(out), Wt.N, M,
        Wt.N, K, EPI_F32, adds, _stream()), "sd_norm_linear_adds")
    return out


def draft_attn_oproj(q, k_cache, v_cache, block_tables, seq_lens, pos_offset, o_w, B, q_len, Hq, Hkv, D,
                     scale, max_ctx, zero: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sd_draft_attn_oproj: attention over the whole context + output projection; returns the
    slabs [Hkv, B*q_len, H] f32 (one partial projection row per KV head).  zero: an f32 tensor
    the launch clears."""
    _need(q, torch.bfloat16, "q")
    Wt = _tiled(o_w, EPI_STORE)
    H = Wt.N
    slabs = torch.empty(Hkv, B * q_len, H, dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().sd_draft_attn_oproj(
        _p(q), q.shape[1], _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1],
        _p(seq_lens), pos_offset, _p(Wt.data), _p(slabs), _p(zero), 0 if zero is None else zero.numel(),
        B, q_len, Hq, Hkv, D, H, float(scale), max_ctx, _stream()), "sd_draft_attn_oproj")
    return slabs


def draft_mlp(h, add, norm_w, eps, up_w, down_km, acc, F: int, h_out=None) -> None:
    """sd_draft_mlp: acc [M, H] f32 += down(silu(gate(n)) * up(n)), n = RMSNorm(bf16(h + sum of the
    slabs add [n_add, M, H])); h_out receives bf16(h + sum)."""
    _need(h, torch.bfloat16, "h")
    _need(acc, torch.float32, "acc")
    M, H = h.shape
    Wu = _tiled(up_w, EPI_SILU_MUL)
    n_add = 0 if add is None else add.shape[0]
    _lib.check(_lib.load().sd_draft_mlp(
        _p(h), H, _p(add), n_add, _p(h_out), _p(norm_w), float(eps), _p(Wu.data), _p(down_km), _p(acc),
        M, H, F, _stream()), "sd_draft_mlp")


# ---------------------------------------------------------------- fused two-layer launches
def chain_state(hidden: int, device="cuda") -> torch.Tensor:
    """The device state of the fused launches below (sd_chain_state_init): tagged-word exchange
    buffer + launch counter.  One per model and per stream that may run passes concurrently."""
    lib = _lib.load()
    n = int(lib.sd_chain_state_bytes(hidden))
    st = torch.zeros(n + 256, dtype=torch.uint8, device=device)
    off = (-st.data_ptr()) % 256
    st = st[off:off + n]
    _lib.check(lib.sd_chain_state_init(_p(st), n, hidden, _stream()), "sd_chain_state_init")
    return st


def chain_supported(rows: int, hidden: int, q_dim: int, ffn: int) -> bool:
    return bool(_lib.load().sd_chain_supported(rows, hidden, q_dim, ffn))


def chain_errors(state: torch.Tensor) -> int:
    failed = C.c_int(0)
    _lib.check(_lib.load().sd_chain_errors(_p(state), C.byref(failed)), "sd_chain_errors")
    return failed.value


def chain_o_up(state, o_w: TiledWeight, h, norm_w, eps, up_w: TiledWeight, B, q_len, Hq, Hkv, D, *,
               attn=None, workspace=None, seq_lens=None, pos_offset=0, max_ctx=1, split_tokens=256,
               o_bias=None, up_bias=None, act=None) -> torch.Tensor:
    """h += o_w @ attention (rows `attn`, or merged from the partials in `workspace`);
    returns act = silu(gate) * up of RMSNorm(h).  h is updated in place."""
    _need(h, torch.bfloat16, "h")
    M, H = h.shape
    F = up_w.N
    if act is None:
        act = torch.empty(M, F, dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.load().sd_chain_o_up(
        _p(state), _p(o_w.data), _p(o_w.scale), _p(o_bias), _p(attn),
        0 if attn is None else attn.stride(0), _p(workspace),
        0 if workspace is None else workspace.numel(), _p(seq_lens), pos_offset, B, q_len, Hq, Hkv, D,
        max_ctx, split_tokens, _p(h), H, _p(norm_w), float(eps), _p(up_w.data), _p(up_w.scale),
        _p(up_bias), _p(act), act.stride(0), F, _stream()), "sd_chain_o_up")
    return act


def chain_down_qkv(state, down_w: TiledWeight, act, h, norm_w, eps, qkv_w: TiledWeight, rope_cs,
                   k_cache, v_cache, block_tables, seq_lens, pos_offset, B, q_len, Hq, Hkv, D, *,
                   down_bias=None, qkv_bias=None) -> torch.Tensor:
    """h += down_w @ act; returns the rotated q of RMSNorm(h) @ qkv_w (K/V appended to the
    cache).  h is updated in place."""
    _need(h, torch.bfloat16, "h")
    _need(act, torch.bfloat16, "act")
    M, H = h.shape
    q = torch.empty(M, Hq * D, dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.load().sd_chain_down_qkv(
        _p(state), _p(down_w.data), _p(down_w.scale), _p(down_bias), _p(act), act.stride(0),
        act.shape[1], _p(h), H, _p(norm_w), float(eps), _p(qkv_w.data), _p(qkv_w.scale),
        _p(qkv_bias), _p(q), Hq * D, _p(rope_cs), 0 if rope_cs is None else rope_cs.shape[0],
        _p(k_cache), _p(v_cache), _p(block_tables), block_tables.shape[1], _p(seq_lens), pos_offset,
        B, q_len, Hq, Hkv, D, _stream()), "sd_chain_down_qkv")
    return q
