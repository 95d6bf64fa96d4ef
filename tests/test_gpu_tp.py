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


class _RankSum:
    """All-reduce / all-gather of n virtual ranks that enqueue on one stream from n threads."""

    def __init__(self, n=2):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.models = [None] * n
        self.error = None

    def view(self, rank, ptr, n):
        models = self.models[rank]
        for m in models if isinstance(models, (list, tuple)) else [models]:
            ws = m.workspace
            off = ptr - ws.data_ptr()
            if 0 <= off and off + 2 * n <= ws.numel():
                return ws[off:off + 2 * n].view(torch.bfloat16)
        raise AssertionError("the buffer is in none of the rank's workspaces")

    def make(self, rank):
        def fn(ptr, n, stream):
            try:
                self.slots[rank] = self.view(rank, ptr, n)
                self.barrier.wait(timeout=120)         # every partial product is enqueued
                total = self.slots[0].float()          # f32 sum in rank order, rounded once
                for q in range(1, self.n):
                    total = total + self.slots[q].float()
                total = total.to(torch.bfloat16)
                self.barrier.wait(timeout=120)         # every sum is enqueued before any write
                self.slots[rank].copy_(total)
                return 0
            except Exception as e:  # noqa: BLE001 - reported by the test thread
                self.error = e
                return -1
        return fn

    def make_gather(self, rank):
        """All-gather of the virtual ranks (same stream, same barrier discipline)."""
        def fn(send, recv):
            try:
                self.slots[rank] = send
                self.barrier.wait(timeout=120)
                every = torch.cat([self.slots[q].reshape(-1).view(torch.uint8) for q in range(self.n)])
                self.barrier.wait(timeout=120)
                recv.reshape(-1).view(torch.uint8).copy_(every)
            except Exception as e:  # noqa: BLE001
                self.error = e
                raise
        return fn


_TwoRankSum = _RankSum


def _run_ranks(n, worker):
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    torch.cuda.synchronize()


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


def test_distributed_argmax_equals_argmax_of_the_whole_row():
    """Slice maxima of vocabulary shards, gathered and combined, finish to the first argmax of
    the whole row (ties to the lowest global index, also across shard borders)."""
    from vllm_omni_draft_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(9)
    for n_ranks, V, R in ((2, 512, 3), (4, 50304, 6), (8, 128256, 6)):
        Vl = V // n_ranks
        assert Vl % 16 == 0 and Vl * n_ranks == V
        logits = torch.randn(R, V, generator=gen, device="cuda")
        logits[0, 5] = logits[0, Vl + 3] = 50.0            # tie across two shards: index 5 wins
        logits[1, V - 1] = 60.0                             # maximum in the last shard
        if R > 2:
            logits[2, Vl - 1] = logits[2, Vl] = 70.0        # tie at a shard border
        n = R * ops.ARGMAX_PARTS
        blocks = []
        for r in range(n_ranks):
            send = torch.empty(2 * n, dtype=torch.float32, device="cuda")
            ops.argmax_partial(logits[:, r * Vl:(r + 1) * Vl].contiguous(), send[:n].view(R, -1),
                               send[n:].view(torch.int32).view(R, -1))
            blocks.append(send)
        pv = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.float32, device="cuda")
        pi = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.int32, device="cuda")
        ops.argmax_combine(torch.cat(blocks), n_ranks, R, Vl, pv, pi)
        got = ops.argmax_from_partials(pv, pi, R)
        want = ops.argmax_rows(logits)
        assert torch.equal(got, want)
        assert got[0].item() == 5 and got[1].item() == V - 1
        if R > 2:
            assert got[2].item() == Vl - 1


@pytest.mark.parametrize("temperature", [0.0, 1.0])
def test_two_rank_engine_with_vocabulary_parallel_head(temperature):
    """Two virtual ranks (threads, one stream, host callbacks as collectives) run whole
    engines on sharded targets.  The vocabulary-parallel head (slice maxima exchanged, or rows
    gathered for sampling) must give the stream of the replicated head, rank for rank; greedy
    speculative decoding must equal greedy autoregressive decoding under the same sharding."""
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                      ffn=512, vocab=512, max_pos=256, tie_embeddings=True)
    dc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                      ffn=256, vocab=512, max_pos=256)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=7).items()}
    cfg = _cfg_of(oc)
    dw = weights.from_state_dict(_cfg_of(dc), {k: v.to(torch.bfloat16) for k, v in
                                               random_state(dc, seed=8).items()}, "cuda")
    stream = torch.cuda.Stream()
    prompts = [[5, 9, 200, 31, 7], list(range(40, 60))]

    def run(vocab_parallel, k):
        coll = _TwoRankSum()
        engines = []
        for r in range(2):
            local, st = weights.shard_state_dict(cfg, state, 2, r, vocab_parallel=vocab_parallel)
            assert local.head_rows == (256 if vocab_parallel else 512)
            tw = weights.from_state_dict(local, st, "cuda")
            group = parallel.CallbackAllReduce(r, coll.make(r), world_size=2,
                                               gather=coll.make_gather(r))
            eng = SpecEngine(tw, dw if k else None, EngineConfig(
                k=k, max_batch=2, max_seq_len=120, split_tokens=128, use_graph=False,
                temperature=temperature, seed=11), tp_group=group)
            eng.stream = stream
            coll.models[r] = eng.target
            engines.append(eng)
        torch.cuda.synchronize()
        outs = [None, None]

        def worker(r):
            eng = engines[r]
            toks = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
            while min(len(t) for t in toks) < 24:
                for t, new in zip(toks, eng.step()):
                    t.extend(new)
            outs[r] = [t[:24] for t in toks]

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
            assert not t.is_alive()
        torch.cuda.synchronize()
        assert coll.error is None, coll.error
        assert outs[0] == outs[1]          # the ranks agree without exchanging decisions
        return outs[0], engines[0].metrics()

    spec_vp, m = run(True, 3)
    spec_rep, _ = run(False, 3)
    assert spec_vp == spec_rep
    assert m["draft"] > 0 and m["emitted"] >= 24
    if temperature == 0.0:
        ar_vp, _ = run(True, 0)
        assert spec_vp == ar_vp


def test_rccl_group_of_one():
    from vllm_omni_draft_amd import parallel
    g = parallel.TensorParallelGroup(0, 1)
    x = torch.randn(5, 1000, device="cuda").to(torch.bfloat16)
    y = x.clone()
    g.all_reduce(y)
    send = torch.arange(300, dtype=torch.float32, device="cuda")
    recv = torch.zeros_like(send)
    g.all_gather(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(x, y) and torch.equal(send, recv)
    g.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_rccl_collectives_inside_the_step_graph(use_graph):
    """An RCCL group of one rank under a whole engine: the all-reduces of the row-parallel
    projections and the all-gather of the vocabulary-parallel head are enqueued (and, with
    use_graph, captured into the step's graph and replayed); the stream equals the one of the
    engine without a group."""
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                      ffn=512, vocab=512, max_pos=256, tie_embeddings=True)
    dc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                      ffn=256, vocab=512, max_pos=256)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=7).items()}
    cfg = _cfg_of(oc)
    dw = weights.from_state_dict(_cfg_of(dc), {k: v.to(torch.bfloat16) for k, v in
                                               random_state(dc, seed=8).items()}, "cuda")
    prompts = [[5, 9, 200, 31, 7], list(range(40, 60))]

    def run(group):
        if group is None:
            tw = weights.from_state_dict(cfg, state, "cuda")
        else:
            local, st = weights.shard_state_dict(cfg, state, 1, 0, vocab_parallel=True)
            assert local.lm_vocab == oc.vocab
            tw = weights.from_state_dict(local, st, "cuda")
        eng = SpecEngine(tw, dw, EngineConfig(k=3, max_batch=2, max_seq_len=120, split_tokens=128,
                                              use_graph=use_graph), tp_group=group)
        toks = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
        while min(len(t) for t in toks) < 30:
            for t, new in zip(toks, eng.step()):
                t.extend(new)
        return [t[:30] for t in toks], eng

    want, _ = run(None)
    group = parallel.TensorParallelGroup(0, 1)
    got, eng = run(group)
    assert got == want
    assert eng.vp and (not use_graph or eng._graphs)
    del eng
    group.close()


@pytest.mark.parametrize("draft_vp,temperature", [(False, 0.0), (True, 0.0), (True, 1.0)])
def test_two_rank_engine_with_sharded_draft(draft_vp, temperature):
    """Target and draft both sharded over two virtual ranks (the draft's LM head whole or
    split by vocabulary like the target's).  With the target's own weights as
    the draft every proposal is accepted (rows are computed alike in draft and verify passes);
    with another draft the stream still equals the autoregressive one; the ranks agree."""
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                      ffn=512, vocab=512, max_pos=256)
    dc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=2, head_dim=64,
                      ffn=256, vocab=512, max_pos=256)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=7).items()}
    dstate = {k: v.to(torch.bfloat16) for k, v in random_state(dc, seed=8).items()}
    cfg, dcfg = _cfg_of(oc), _cfg_of(dc)
    stream = torch.cuda.Stream()
    prompts = [[5, 9, 200, 31, 7], list(range(40, 60))]

    def run(k, draft, vp=draft_vp):
        coll = _TwoRankSum()
        engines = []
        for r in range(2):
            local, st = weights.shard_state_dict(cfg, state, 2, r, vocab_parallel=True)
            tw = weights.from_state_dict(local, st, "cuda")
            dw = None
            if k:
                dlocal, dst = weights.shard_state_dict(*draft, 2, r, vocab_parallel=vp)
                assert dlocal.head_rows == (256 if vp else 512)
                dw = weights.from_state_dict(dlocal, dst, "cuda")
            group = parallel.CallbackAllReduce(r, coll.make(r), world_size=2,
                                               gather=coll.make_gather(r))
            eng = SpecEngine(tw, dw, EngineConfig(k=k, max_batch=2, max_seq_len=120, split_tokens=128,
                                                  use_graph=False, temperature=temperature, seed=11),
                             tp_group=group, draft_tp=bool(k))
            eng.stream = stream
            coll.models[r] = [eng.target] + ([eng.draft] if k else [])
            engines.append(eng)
        torch.cuda.synchronize()
        outs = [None, None]

        def worker(r):
            eng = engines[r]
            toks = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
            while min(len(t) for t in toks) < 24:
                for t, new in zip(toks, eng.step()):
                    t.extend(new)
            outs[r] = [t[:24] for t in toks]

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
            assert not t.is_alive()
        torch.cuda.synchronize()
        assert coll.error is None, coll.error
        assert outs[0] == outs[1]
        return outs[0], engines[0].metrics()

    if temperature > 0.0:
        # sampling: the split head gathers the rows the whole head computes: same stream
        split, m = run(3, (dcfg, dstate))
        whole, _ = run(3, (dcfg, dstate), vp=False)
        assert split == whole and m["draft"] > 0
        return
    ar, _ = run(0, None)
    same, m = run(3, (cfg, state))
    assert same == ar and m["accepted"] == m["draft"] > 0
    other, m = run(3, (dcfg, dstate))
    assert other == ar and m["draft"] > 0
    with pytest.raises(ValueError):
        SpecEngine(weights.from_state_dict(cfg, state, "cuda"), None, EngineConfig(k=0), draft_tp=True)


# ---------------------------------------------------------------------------- the named tensor-parallel configuration
LLAMA3_70B_2L = OracleConfig(arch="llama", n_layers=2, hidden=8192, n_heads=64, n_kv_heads=8, head_dim=128,
                             ffn=28672, vocab=128256, rope_theta=500000.0,
                             rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                           "high_freq_factor": 4.0,
                                           "original_max_position_embeddings": 8192},
                             max_pos=512)
LLAMA3_8B_2L = OracleConfig(arch="llama", n_layers=2, hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128,
                            ffn=14336, vocab=128256, rope_theta=500000.0,
                            rope_scaling=LLAMA3_70B_2L.rope_scaling, max_pos=512)


@pytest.mark.timeout(900)
def test_eight_ranks_at_llama3_70b_width_against_the_oracle():
    """Llama-3-70B's layer shapes (two layers) sharded over EIGHT virtual ranks as the tensor-
    parallel configuration shards them: 8 query heads on 1 KV head per rank (the attention
    kernel's two-tile arm at k + 1 = 5 rows), ffn 28672 / 8, the LM head split into 8 vocabulary
    slices of 16032 rows, a Llama-3-8B-shaped draft replicated on every rank, k = 4.
    (1) The ranks' logit slices of a prompt chunk, decode rows and a 5-row verify pass, put side
    by side, against the oracle's logits of the unsharded model; (2) the distributed arg-max
    (slice maxima exchanged and combined) equals the arg-max of the whole row; (3) whole engines:
    the ranks emit the same streams, greedy speculative == greedy autoregressive, proposals
    are made and partly accepted."""
    from oracle.models import OracleLM
    from vllm_omni_draft_amd import ops, parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    N, k = 8, 4
    oc = LLAMA3_70B_2L
    state = {n_: v.to(torch.bfloat16) for n_, v in random_state(oc, seed=17).items()}
    cfg = _cfg_of(oc, rope_scaling=oc.rope_scaling)
    stream = torch.cuda.Stream()
    dev = "cuda"
    shards = []
    for r in range(N):
        local, st = weights.shard_state_dict(cfg, state, N, r, vocab_parallel=True)
        assert (local.n_heads, local.n_kv_heads, local.ffn, local.lm_vocab) == (8, 1, 3584, 16032)
        shards.append(weights.from_state_dict(local, st, dev))

    # ---- (1) + (2): forward passes of the eight shards
    toks = torch.randint(0, oc.vocab, (18,), generator=torch.Generator().manual_seed(3)).tolist()
    tokens = torch.tensor(toks, dtype=torch.int32, device=dev)
    chunks = [(0, 6), (6, 1), (7, 1), (8, 5), (13, 5)]         # prompt chunk, decode rows, verify passes
    max_ctx = 64
    coll = _RankSum(N)
    ranks = []
    with torch.cuda.stream(stream):
        for r in range(N):
            group = parallel.CallbackAllReduce(r, coll.make(r), world_size=N)
            m = DecoderModel(shards[r], max_ctx, 128, tp_group=group)
            kv = PagedKVCache(shards[r].cfg, 8, 1, max_ctx, dev)
            kv.ensure_capacity(0, max_ctx)
            coll.models[r] = m
            ranks.append((m, kv, [torch.empty(q, 16032, dtype=torch.float32, device=dev) for _, q in chunks]))
    torch.cuda.synchronize()
    zero = torch.zeros(1, dtype=torch.int32, device=dev)

    def forward(r):
        m, kv, outs = ranks[r]
        with torch.cuda.stream(stream):
            for (c0, q), lg in zip(chunks, outs):
                m.forward(kv, tokens, 0, c0, zero, c0, 1, q, lg, max_ctx)

    _run_ranks(N, forward)
    assert coll.error is None, coll.error
    got = torch.cat([torch.cat([ranks[r][2][i] for r in range(N)], dim=1) for i in range(len(chunks))]).cpu()
    lm = OracleLM(oc, {n_: v.float() for n_, v in state.items()}, bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.8
    del lm, ref
    # slice maxima of the last verify pass, gathered in rank order and combined
    R = chunks[-1][1]
    n = R * ops.ARGMAX_PARTS
    blocks = []
    for r in range(N):
        send = torch.empty(2 * n, dtype=torch.float32, device=dev)
        ops.argmax_partial(ranks[r][2][-1], send[:n].view(R, -1), send[n:].view(torch.int32).view(R, -1))
        blocks.append(send)
    pv = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.float32, device=dev)
    pi = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.int32, device=dev)
    ops.argmax_combine(torch.cat(blocks), N, R, 16032, pv, pi)
    assert ops.argmax_from_partials(pv, pi, R).cpu().tolist() == got[-R:].argmax(-1).tolist()
    del ranks

    # ---- (3) whole engines, the draft replicated
    dc = LLAMA3_8B_2L
    dw = weights.from_state_dict(_cfg_of(dc, rope_scaling=dc.rope_scaling),
                                 {n_: v.to(torch.bfloat16) for n_, v in random_state(dc, seed=18).items()}, dev)
    prompts = [toks[:9]]

    def run(kk):
        coll = _RankSum(N)
        engines = []
        for r in range(N):
            group = parallel.CallbackAllReduce(r, coll.make(r), world_size=N, gather=coll.make_gather(r))
            eng = SpecEngine(shards[r], dw if kk else None, EngineConfig(
                k=kk, max_batch=1, max_seq_len=80, split_tokens=128, use_graph=False), tp_group=group)
            eng.stream = stream
            coll.models[r] = eng.target
            engines.append(eng)
        torch.cuda.synchronize()
        outs = [None] * N

        def worker(r):
            eng = engines[r]
            t = [eng.add_request(0, prompts[0])]
            while len(t) < 20:
                t.extend(eng.step()[0])
            outs[r] = t[:20]

        _run_ranks(N, worker)
        assert coll.error is None, coll.error
        assert all(o == outs[0] for o in outs)      # the ranks agree without exchanging decisions
        return outs[0], engines[0].metrics()

    ar, _ = run(0)
    spec, m = run(k)
    assert spec == ar
    assert m["draft"] > 0 and m["draft"] % k == 0 and m["emitted"] >= 19


def test_a_timed_out_exchange_drops_the_steps_tokens():
    """A collective that reports a wait that gave up (IpcTensorParallelGroup.timeouts) makes the
    step raise instead of handing out tokens computed from partial sums that were never
    combined; the next healthy step goes on."""
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                      ffn=256, vocab=256, max_pos=128)
    cfg = _cfg_of(oc)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=2).items()}
    local, st = weights.shard_state_dict(cfg, state, 1, 0)
    group = parallel.CallbackAllReduce(0, lambda ptr, n, stream: 0, world_size=1)
    seen = {"n": 0}
    group.timeouts = lambda: seen["n"]
    eng = SpecEngine(weights.from_state_dict(local, st, "cuda"), None,
                     EngineConfig(k=0, max_batch=1, max_seq_len=64, split_tokens=128, use_graph=False),
                     tp_group=group)
    eng.add_request(0, [1, 2, 3])
    assert len(eng.step()[0]) == 1
    seen["n"] = 2
    with pytest.raises(RuntimeError, match="timed out"):
        eng.step()
    assert len(eng.step()[0]) == 1          # counted once; later steps are served
