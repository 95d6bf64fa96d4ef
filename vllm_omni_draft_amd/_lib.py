"""ctypes binding of libsd_amd.so (the C ABI declared in include/sd_amd.h).

There is no fallback: if the library is missing or a call fails, an exception
is raised.  Tensors cross the boundary as raw device pointers only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_NAME = "libsd_amd.so"
_lib: Optional[C.CDLL] = None

c_i32p = C.POINTER(C.c_int32)
vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int


class SdError(RuntimeError):
    pass


class LayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in (
        "attn_norm_w", "attn_norm_b", "qkv_w", "qkv_b", "o_w", "o_b", "mlp_norm_w", "mlp_norm_b",
        "up_w", "up_b", "down_w", "down_b", "qkv_s", "o_s", "up_s", "down_s", "down_w_km")]


class ModelDesc(C.Structure):
    _fields_ = [
        ("arch", i32), ("n_layers", i32), ("hidden", i32), ("n_heads", i32), ("n_kv_heads", i32),
        ("head_dim", i32), ("ffn", i32), ("vocab", i32), ("max_pos", i32), ("pos_bias", i32),
        ("norm_eps", C.c_float), ("attn_scale", C.c_float),
        ("embed", vp), ("embed_tiled", i32), ("pos_table", vp), ("rope_cs", vp), ("final_norm_w", vp),
        ("final_norm_b", vp), ("lm_head", vp), ("layers", C.POINTER(LayerWeights)),
        ("tp_rank", i32), ("allreduce", vp), ("allreduce_ctx", vp),
        ("weight_fp8", i32), ("lm_head_s", vp), ("lm_vocab", i32),
        ("mrope_section_t", i32), ("mrope_section_h", i32),
    ]


# int (*sd_allreduce_fn)(void* buf, size_t n_bf16, void* ctx, sd_stream_t stream)
ALLREDUCE_FN = C.CFUNCTYPE(i32, vp, C.c_size_t, vp, vp)


class KvCacheDesc(C.Structure):
    _fields_ = [("k_cache", vp), ("v_cache", vp), ("layer_stride_bytes", i64),
                ("block_tables", vp), ("max_blocks_per_seq", i32)]


class ForwardArgs(C.Structure):
    _fields_ = [("B", i32), ("q_len", i32), ("token_ids", vp), ("tok_stride", i64),
                ("tok_offset", i64), ("seq_lens", vp), ("pos_offset", i32), ("max_ctx", i32),
                ("split_tokens", i32), ("logits", vp), ("wide", i32),
                ("input_embeds", vp), ("rope_pos", vp), ("rope_delta", vp), ("tree_mask", vp),
                ("chain_state", vp), ("draft_fused", i32), ("read_ahead_mb", i32),
                ("wide_a8", i32), ("argmax_val", vp), ("argmax_idx", vp)]


class UniformRegion(C.Structure):
    _fields_ = [("offset", i64), ("n", i32), ("stride_j", i64), ("stride_b", i64)]


class RowAdds(C.Structure):
    _fields_ = [("add", vp), ("ld_add", i64), ("x_out", vp)]


class RopePositions(C.Structure):
    _fields_ = [("pos", vp), ("delta", vp), ("section_t", i32), ("section_h", i32)]


# name -> (restype, argtypes); every symbol include/sd_amd.h declares.
SIGNATURES = {
    "sd_version": (i32, []),
    "sd_last_error": (i32, [C.c_char_p, C.c_size_t]),
    "sd_device_info": (i32, [C.POINTER(i32), C.c_char_p, C.c_size_t]),
    "sd_rejection_sample": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "sd_typical_acceptance": (i32, [vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, i32, i32, i32, vp]),
    "sd_greedy_verify": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    "sd_argmax_rows": (i32, [vp, i64, vp, i32, i32, vp]),
    "sd_softmax_rows": (i32, [vp, i64, C.c_float, vp, i64, i32, i32, vp]),
    "sd_filter_probs_seq": (i32, [vp, i64, vp, vp, vp, i32, i32, i32, vp]),
    "sd_add_logit_bias": (i32, [vp, i64, vp, i64, vp, i32, i32, i32, vp]),
    "sd_token_logprobs": (i32, [vp, i64, vp, i32, i32, i32, vp, vp, vp, vp]),
    "sd_softmax_rows_seq": (i32, [vp, i64, vp, i32, vp, i64, i32, i32, vp, vp]),
    "sd_sample_rows": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "sd_rope_append": (i32, [vp, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sd_paged_attention_workspace_bytes": (C.c_size_t, [i32, i32, i32, i32, i32, i32]),
    "sd_paged_attention": (i32, [vp, i64, vp, vp, vp, i32, vp, i32, vp, i64, i32, i32, i32, i32,
                                 i32, C.c_float, i32, i32, vp, C.c_size_t, vp]),
    "sd_paged_attention_blocked": (i32, [vp, i64, vp, vp, vp, i32, vp, i32, vp, i64, i32, i32, i32,
                                         i32, i32, i32, C.c_float, i32, i32, vp, C.c_size_t, vp]),
    "sd_paged_attention_static": (i32, [vp, i64, vp, vp, vp, i32, vp, i32, vp, i64, i32, i32, i32,
                                        i32, i32, C.c_float, i32, i32, vp]),
    "sd_paged_attention_tree": (i32, [vp, i64, vp, vp, vp, i32, vp, i32, vp, vp, i64, i32, i32, i32,
                                      i32, i32, C.c_float, i32, i32, vp, C.c_size_t, vp]),
    "sd_map_tokens": (i32, [vp, i64, i32, vp, i32, i32, vp]),
    "sd_tree_prepare": (i32, [vp, i64, i32, vp, vp, vp, i32, vp, i32, i32, vp]),
    "sd_tree_accept_greedy": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp]),
    "sd_tree_accept_sample": (i32, [vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "sd_kv_compact": (i32, [vp, vp, i64, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "sd_linear": (i32, [vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "sd_norm_linear": (i32, [vp, vp, vp, i64, vp, vp, C.c_float, i32, vp, vp, i64, i32, i32, i32, i32, vp]),
    "sd_norm_qkv_rope": (i32, [vp, vp, vp, i64, vp, vp, C.c_float, i32, vp, vp, i64, vp, i32, vp, vp, vp,
                               i32, vp, i32, i32, i32, i32, i32, i32, i32,
                               C.POINTER(RopePositions), vp]),
    "sd_norm_linear_adds": (i32, [vp, vp, vp, i64, vp, vp, C.c_float, i32, vp, vp, i64, i32, i32, i32, i32,
                                  C.POINTER(RowAdds), vp]),
    "sd_norm_qkv_rope_adds": (i32, [vp, vp, vp, i64, vp, vp, C.c_float, i32, vp, vp, i64, vp, i32, vp, vp, vp,
                                    i32, vp, i32, i32, i32, i32, i32, i32, i32,
                                    C.POINTER(RopePositions), C.POINTER(RowAdds), vp]),
    "sd_draft_layer_supported": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "sd_draft_attn_oproj": (i32, [vp, i64, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32,
                                  i32, i32, C.c_float, i32, vp]),
    "sd_draft_mlp": (i32, [vp, i64, vp, i32, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, vp]),
    "sd_rmsnorm": (i32, [vp, i64, vp, vp, i64, i32, i32, C.c_float, vp]),
    "sd_layernorm": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, C.c_float, vp]),
    "sd_embed": (i32, [vp, i64, i64, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "sd_tiled_weight_elems": (C.c_size_t, [i32, i32, i32]),
    "sd_tile_weights": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sd_argmax_combine": (i32, [vp, vp, i32, i32, i32, i64, vp, vp, vp]),
    "sd_argmax_from_partials": (i32, [vp, vp, vp, i32, vp]),
    "sd_tp_allgather": (i32, [vp, vp, C.c_size_t, vp, vp]),
    "sd_tile_weights_fp8": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sd_model_create": (vp, [C.POINTER(ModelDesc)]),
    "sd_model_destroy": (None, [vp]),
    "sd_model_workspace_bytes": (C.c_size_t, [vp, i32, i32, i32]),
    "sd_model_forward": (i32, [vp, C.POINTER(KvCacheDesc), C.POINTER(ForwardArgs), vp,
                               C.c_size_t, vp]),
    "sd_store_proposal": (i32, [vp, vp, i32, i32, i32, vp]),
    "sd_debug_force_proposals": (i32, [vp, i32, vp, vp, i64, C.c_char_p, i32, vp, i32, i32, vp]),
    "sd_gather_proposals": (i32, [vp, vp, i32, i32, vp]),
    "sd_set_proposals": (i32, [vp, vp, i32, i32, vp]),
    "sd_advance_state": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, i32, i32, vp]),
    "sd_linear_wide_workspace_bytes": (C.c_size_t, [i32, i32, i32, i32]),
    "sd_linear_wide": (i32, [vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, C.c_size_t, vp]),
    "sd_qkv_rope_wide": (i32, [vp, vp, vp, i64, vp, vp, i64, vp, i32, vp, vp, vp, i32, vp, i32, i32,
                               i32, i32, i32, i32, i32, C.POINTER(RopePositions), vp, C.c_size_t, vp]),
    "sd_row_scales_fp8": (i32, [vp, i64, i32, i32, vp, vp]),
    "sd_topk_tokens": (i32, [vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, i32, vp]),
    "sd_norm_linear_argmax": (i32, [vp, vp, vp, i64, vp, vp, C.c_float, i32, vp, vp, i32, i32, i32, vp]),
    "sd_linear_wide_a8": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, C.c_size_t, vp]),
    "sd_qkv_rope_wide_a8": (i32, [vp, vp, vp, vp, i64, vp, vp, i64, vp, i32, vp, vp, vp, i32, vp, i32, i32,
                               i32, i32, i32, i32, i32, C.POINTER(RopePositions), vp, C.c_size_t, vp]),
    "sd_linear_attn_merge": (i32, [vp, vp, vp, C.c_size_t, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                   vp, vp, vp, i64, i32, i32, vp]),
    "sd_tp_unique_id": (i32, [vp]),
    "sd_tp_comm_create": (vp, [vp, i32, i32]),
    "sd_tp_comm_destroy": (None, [vp]),
    "sd_tp_allreduce": (i32, [vp, C.c_size_t, vp, vp]),
    "sd_sampling_scratch_bytes": (C.c_size_t, [i32]),
    "sd_softmax_rows_split": (i32, [vp, i64, C.c_float, vp, i64, i32, i32, vp, vp]),
    "sd_sample_rows_split": (i32, [vp, i64, vp, vp, i32, i32, vp, vp]),
    "sd_rejection_sample_split": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "sd_ipc_buffer_bytes": (C.c_size_t, [C.c_size_t]),
    "sd_ipc_buffer_create": (i32, [C.c_size_t, C.POINTER(vp), vp]),
    "sd_ipc_buffer_open": (i32, [vp, C.POINTER(vp)]),
    "sd_ipc_buffer_close": (i32, [vp]),
    "sd_ipc_buffer_destroy": (i32, [vp]),
    "sd_ipc_comm_create": (vp, [i32, i32, C.POINTER(vp), C.c_size_t]),
    "sd_ipc_comm_destroy": (None, [vp]),
    "sd_ipc_comm_errors": (i32, [vp, C.POINTER(i32)]),
    "sd_ipc_comm_set_timeout_ms": (i32, [vp, i32]),
    "sd_ipc_allreduce": (i32, [vp, C.c_size_t, vp, vp]),
    "sd_ipc_allgather": (i32, [vp, vp, C.c_size_t, vp, vp]),
    "sd_fill_uniforms": (i32, [vp, i32, C.c_uint64, vp, vp]),
    "sd_fill_uniforms_seq": (i32, [vp, vp, i32, vp, vp, i32, vp]),
    "sd_argmax_partial": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "sd_propose_from_partials": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, vp]),
    "sd_verify_from_partials": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, i32, i32,
                                      vp]),
    "sd_chain_state_bytes": (C.c_size_t, [i32]),
    "sd_chain_state_init": (i32, [vp, C.c_size_t, i32, vp]),
    "sd_chain_errors": (i32, [vp, C.POINTER(i32)]),
    "sd_chain_supported": (i32, [i32, i32, i32, i32]),
    "sd_chain_o_up": (i32, [vp, vp, vp, vp, vp, i64, vp, C.c_size_t, vp, i32, i32, i32, i32, i32, i32,
                            i32, i32, vp, i32, vp, C.c_float, vp, vp, vp, vp, i64, i32, vp]),
    "sd_chain_down_qkv": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, i32, vp, C.c_float, vp, vp, vp, vp,
                                i64, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sd_graph_begin": (i32, [vp]),
    "sd_graph_end": (i32, [vp, C.POINTER(vp)]),
    "sd_graph_launch": (i32, [vp, vp]),
    "sd_graph_destroy": (None, [vp]),
}


def backend() -> str:
    """"gpu" (libsd_amd.so, gfx950; the default) or "cpu" (libsd_cpu.so, the C++/OpenMP build of
    the same ABI for hosts without a GPU), chosen for the whole process by SD_AMD_BACKEND before
    the first call.  The choice is explicit: a missing GPU library is an error, never a reason
    to load the other one."""
    b = os.environ.get("SD_AMD_BACKEND", "gpu").lower()
    if b not in ("gpu", "cpu"):
        raise SdError(f"SD_AMD_BACKEND must be 'gpu' or 'cpu', not {b!r}")
    return b


def lib_path() -> str:
    """The in-tree library of the chosen backend; SD_AMD_LIB names another build of the GPU
    sources (the timestamp-instrumented one of `make stamps`, for experiments/)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if backend() == "cpu":
        return os.path.join(here, "libsd_cpu.so")
    override = os.environ.get("SD_AMD_LIB")
    if override:
        return os.path.abspath(override)
    return os.path.join(here, _LIB_NAME)


def load() -> C.CDLL:
    """Loads the in-tree library and declares every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SdError(f"{path} is missing: build it with `make` (hipcc --offload-arch=gfx950; "
                      "g++ for the CPU build); nothing falls back to another library or to PyTorch")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().sd_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SdError(f"{what} failed ({rc}): {last_error()}")
