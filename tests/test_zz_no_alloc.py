"""The allocation-free attention entry point (sd_paged_attention_static) against the entry
point it is a variant of (sd_paged_attention with a caller workspace): bit-identical output
on two calls in a row, the second with a smaller input, through the C ABI."""
import gc

import pytest
import torch

from tests.test_gpu_attention import _setup

pytestmark = pytest.mark.gpu


def _case(ops, B, ctx_lens, q_len, Hq, Hkv, D, split, seed):
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=seed)
    kc, vc, bt = kc.cuda(), vc.cuda(), bt.cuda()
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    max_ctx = max(ctx_lens) + q_len
    ref = ops.paged_attention(q, kc, vc, bt, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                              split)
    out = torch.full_like(ref, float("nan"))
    # Garbage of earlier tests (engines with reference cycles) must not be collected while the
    # allocator is being watched: a collection in between frees device memory and changes the count.
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        before = torch.cuda.memory_allocated()
        ops.paged_attention_static(q, kc, vc, bt, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5,
                                   max_ctx, split, out)
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated()
    finally:
        gc.enable()
    assert after == before                               # torch's allocator was not touched
    assert torch.equal(out, ref)


def test_static_scratch_attention_equals_workspace_attention_twice():
    from vllm_omni_draft_amd import ops
    # first call: 2 sequences, 6 splits of 128 keys, verify rows (k + 1 = 6)
    _case(ops, 2, [700, 300], 6, 32, 8, 128, 128, seed=5)
    # second call, smaller: one sequence, 3 splits, one query token; the array still holds
    # the first call's partials
    _case(ops, 1, [333], 1, 32, 8, 64, 128, seed=6)
    # single split: the array is not used at all
    _case(ops, 1, [40], 2, 4, 2, 64, 128, seed=7)


def test_static_scratch_attention_rejects_what_does_not_fit():
    from vllm_omni_draft_amd import _lib, ops
    B, q_len, Hq, Hkv, D, split = 64, 6, 32, 8, 128, 128
    max_ctx = 4096            # 64 * 8 * 32 splits * 32 rows * 130 * 4 bytes = 272 MB of partials
    q = torch.zeros(B * q_len, Hq * D, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros_like(q)
    bt = torch.zeros(B, max_ctx // 16, dtype=torch.int32, device="cuda")
    sl = torch.zeros(B, dtype=torch.int32, device="cuda")
    kc = torch.zeros(4, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.SdError, match="exceed the built-in scratch"):
        ops.paged_attention_static(q, kc, kc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                                   split, out)
