"""Tensor-parallel target on the GPU: the native forward with a two-rank group emulated in
one process (both shards on one card, a host callback as the collective), and the RCCL
group itself at world size 1."""
import threading

import pytest
import torch

from oracle.models import OracleConfig, random_state

pytestmark = pytest.mark.gpu


def _cfg_of(oc, **kw):
    from vllm_omni_draft_amd import config
    return config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        max_pos=oc.max_pos, norm_eps=oc.norm_eps, rope_theta=oc.rope_theta,
        tie_embeddings=oc.tie_embeddings, **kw)


class _TwoRankSum:
    """All-reduce of two virtual ranks that enqueue on one stream from two threads."""

    def __init__(self):
        self.barrier = threading.Barrier(2)
        self.slots = [None, None]
        self.models = [None, None]
        self.error = None

    def view(self, rank, ptr, n):
        ws = self.models[rank].workspace
        off = ptr - ws.data_ptr()
        assert 0 <= off and off + 2 * n <= ws.numel()
        return ws[off:off + 2 * n].view(torch.bfloat16)

    def make(self, rank):
        def fn(ptr, n, stream):
            try:
                self.slots[rank] = self.view(rank, ptr, n)
                self.barrier.wait(timeout=60)          # both partial products are enqueued
                total = self.slots[0] + self.slots[1]  # bf16 sum, as the collective does
                self.barrier.wait(timeout=60)          # both sums are enqueued before any write
                self.slots[rank].copy_(total)
                return 0
            except Exception as e:  # noqa: BLE001 - reported by the test thread
                self.error = e
                return -1
        return fn


@pytest.mark.parametrize("arch", ["llama", "opt"])
def test_two_rank_forward_matches_single_gpu(arch):
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    oc = OracleConfig(arch=arch, n_layers=2, hidden=256, n_heads=4, n_kv_heads=4 if arch == "opt" else 2,
                      head_dim=64, ffn=512, vocab=512, max_pos=256)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=7).items()}
    cfg = _cfg_of(oc)
    dev = "cuda"
    stream = torch.cuda.Stream()
    B, max_ctx = 2, 64
    tokens = torch.randint(0, oc.vocab, (B, 8), dtype=torch.int32, device=dev)
    zero = torch.zeros(B, dtype=torch.int32, device=dev)
    four = torch.full((B,), 4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def run(model, kv, logits):
        # a 4-token chunk per sequence, then a 3-token verify-style pass
        model.forward(kv, tokens, 8, 0, zero, 0, B, 4, None, max_ctx)
        model.forward(kv, tokens, 8, 4, four, 0, B, 3, logits, max_ctx)

    def build(c, st, group=None):
        w = weights.from_state_dict(c, st, dev)
        m = DecoderModel(w, max_ctx, 128, tp_group=group)
        kv = PagedKVCache(c, PagedKVCache.pages_for(c, B, max_ctx), B, max_ctx, dev)
        for b in range(B):
            kv.ensure_capacity(b, max_ctx)
        return m, kv

    with torch.cuda.stream(stream):
        single, kv1 = build(cfg, state)
        ref = torch.empty(B * 3, oc.vocab, dtype=torch.float32, device=dev)
        run(single, kv1, ref)

        coll = _TwoRankSum()
        ranks = []
        for r in range(2):
            local, st = weights.shard_state_dict(cfg, state, 2, r)
            m, kv = build(local, st, parallel.CallbackAllReduce(r, coll.make(r)))
            coll.models[r] = m
            ranks.append((m, kv, torch.empty(B * 3, oc.vocab, dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()

    def worker(r):
        with torch.cuda.stream(stream):
            run(*ranks[r])

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive()
    torch.cuda.synchronize()
    assert coll.error is None, coll.error
    # every rank ends with the same logits, close to the unsharded model's (bf16 partial sums)
    assert torch.equal(ranks[0][2], ranks[1][2])
    scale = ref.abs().max().item()
    torch.testing.assert_close(ranks[0][2], ref, rtol=0, atol=0.03 * scale)
    assert (ranks[0][2].argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.8


def test_rccl_group_of_one():
    from vllm_omni_draft_amd import parallel
    g = parallel.TensorParallelGroup(0, 1)
    x = torch.randn(5, 1000, device="cuda").to(torch.bfloat16)
    y = x.clone()
    g.all_reduce(y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    g.close()
