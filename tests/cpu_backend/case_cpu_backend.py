"""Cases for the CPU build of the C ABI (libsd_cpu.so).  Run in a process of their own with
SD_AMD_BACKEND=cpu (tests/test_cpu_backend.py starts it): the backend is a per-process choice.
The checker is the CPU oracle, as for the GPU library."""
import numpy as np
import pytest
import torch

from oracle import sampling
from oracle.models import OracleConfig, OracleLM, random_state
from oracle.spec_decode import ar_generate


def _bf(state):
    return {k: v.to(torch.bfloat16).float() for k, v in state.items()}


def _product(oc, state):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        norm_eps=oc.norm_eps, rope_theta=oc.rope_theta, max_pos=oc.max_pos,
        tie_embeddings=oc.tie_embeddings)
    return weights.from_state_dict(cfg, state, "cpu")


def test_backend_is_the_cpu_library():
    from vllm_omni_draft_amd import _lib, ops
    assert _lib.backend() == "cpu" and _lib.lib_path().endswith("libsd_cpu.so")
    cus, name = ops.device_info()
    assert name == "cpu" and cus >= 1
    with pytest.raises(_lib.SdError):          # what the CPU build does not provide says so
        _lib.check(_lib.load().sd_graph_begin(None), "sd_graph_begin")


@pytest.mark.parametrize("arch", ["opt", "llama"])
def test_forward_logits_match_the_oracle(arch):
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    oc = (OracleConfig(arch="opt", n_layers=2, hidden=192, n_heads=3, n_kv_heads=3, head_dim=64, ffn=384,
                       vocab=272, max_pos=128, tie_embeddings=True) if arch == "opt" else
          OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64, ffn=256,
                       vocab=256, max_pos=128))
    st = _bf(random_state(oc, seed=3))
    w = _product(oc, st)
    model = DecoderModel(w, max_ctx=128, split_tokens=128)
    kv = PagedKVCache(w.cfg, 16, 1, 128, "cpu")
    tokens = [5, 9, 200, 31, 7, 7, 100, 42, 1, 2, 3]
    kv.ensure_capacity(0, len(tokens))
    tok = torch.tensor(tokens, dtype=torch.int32)
    zero = torch.zeros(1, dtype=torch.int32)
    outs, c0 = [], 0
    for q in (6, 1, 4):                      # prompt chunk, a decode token, a verify-like pass
        lg = torch.empty(q, oc.vocab, dtype=torch.float32)
        model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128)
        outs.append(lg)
        c0 += q
    got = torch.cat(outs)
    lm = OracleLM(oc, st, bf16_acts=True)
    ref = lm.forward(tokens, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale
    assert (got.argmax(dim=1) == ref.argmax(dim=1)).float().mean().item() > 0.8


def test_acceptance_entry_points_match_the_oracle():
    from vllm_omni_draft_amd import ops
    rng = np.random.default_rng(0)
    B, k, V = 5, 4, 64
    p = rng.dirichlet(np.ones(V) * 0.3, size=(B, k + 1)).astype(np.float32)
    q = rng.dirichlet(np.ones(V) * 0.3, size=(B, k)).astype(np.float32)
    q64 = q.astype(np.float64)
    ids = np.stack([[rng.choice(V, p=q64[b, i] / q64[b, i].sum()) for i in range(k)]
                    for b in range(B)]).astype(np.int64)
    bonus = rng.integers(0, V, size=B).astype(np.int64)
    ua, ur = rng.random((B, k), dtype=np.float32), rng.random((B, k + 1), dtype=np.float32)
    lens = np.array([4, 2, 0, 4, 3], dtype=np.int32)
    ref, ctr = sampling.rejection_sample(p, bonus, q, ids, ua, ur, lens)
    counters = torch.zeros(3, dtype=torch.int64)
    t = torch.from_numpy
    got = ops.rejection_sample(t(p), t(bonus), t(q), t(ids), t(ua), t(ur), t(lens), counters)
    assert np.array_equal(got.numpy(), ref) and counters.tolist() == list(ctr)
    tids = p.argmax(axis=2).astype(np.int64)
    gref, gctr = sampling.greedy_verify(tids, ids, lens)
    counters.zero_()
    assert np.array_equal(ops.greedy_verify(t(tids), t(ids), t(lens), counters).numpy(), gref)
    assert counters.tolist() == list(gctr)
    x = rng.standard_normal((6, 1000)).astype(np.float32)
    assert ops.argmax_rows(t(x)).tolist() == sampling.argmax_rows(x).tolist()
    pv = torch.empty(6, ops.ARGMAX_PARTS)
    pi = torch.empty(6, ops.ARGMAX_PARTS, dtype=torch.int32)
    ops.argmax_partial(t(x), pv, pi)
    assert ops.argmax_from_partials(pv, pi, 6).tolist() == sampling.argmax_rows(x).tolist()


def test_token_logprobs_match_the_oracle():
    from vllm_omni_draft_amd import ops
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((5, 1000)) * 3).astype(np.float32)
    x[1, 17] = x[1, 900] = x[1].max() + 1.0
    x[4, 3:] = -np.inf
    toks = np.array([int(x[0].argmax()), 900, -1, 123, 1], dtype=np.int32)
    ids, lps, ranks = ops.token_logprobs(torch.from_numpy(x), torch.from_numpy(toks), 4)
    w_ids, w_lps, w_ranks = sampling.token_logprobs(x, toks, 4)
    assert np.array_equal(ids.numpy(), w_ids) and np.array_equal(ranks.numpy(), w_ranks)
    fin = np.isfinite(w_lps)
    assert np.array_equal(np.isfinite(lps.numpy()), fin)
    assert np.abs(lps.numpy()[fin] - w_lps[fin]).max() < 2e-5


def test_config1_plumbing_opt125m_shapes_through_the_worker():
    """BASELINE.json configs[0]: facebook/opt-125m target + opt-125m draft (the same weights),
    k = 3, greedy, batch 8, one process, no GPU.  The draft IS the target, so every proposal is
    accepted: accept-len == k exactly (a known answer), and the streams are the greedy
    autoregressive streams of the CPU oracle."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    oc = OracleConfig(arch="opt", n_layers=12, hidden=768, n_heads=12, n_kv_heads=12, head_dim=64,
                      ffn=3072, vocab=50272, max_pos=256, tie_embeddings=True)
    st = _bf(random_state(oc, seed=125))
    w = _product(oc, st)
    worker = SpecDecodeWorker(SpecDecodeConfig(target=w, draft=w, num_speculative_tokens=3,
                                               max_num_seqs=8, max_model_len=96, split_tokens=128,
                                               rows="narrow"))
    worker.init_device()
    worker.load_model()
    g = torch.Generator().manual_seed(1234)
    prompts = {f"r{i}": torch.randint(0, oc.vocab, (int(n),), generator=g).tolist()
               for i, n in enumerate(torch.randint(4, 24, (8,), generator=g))}
    out = worker.execute_model(ExecuteModelRequest(
        [SequenceGroupMetadata(r, True, p) for r, p in prompts.items()]))
    got = {r: list(t) for r, t in zip(out.req_ids, out.sampled_token_ids)}
    for _ in range(3):
        out = worker.execute_model(ExecuteModelRequest(
            [SequenceGroupMetadata(r, False) for r in prompts]))
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            assert len(toks) == 4, (r, toks)            # k accepted + the bonus token
            got[r].extend(toks)
    m = worker.metrics.maybe_collect(force=True)
    assert m.accept_len == 3.0 and m.draft_acceptance_rate == 1.0
    lm = OracleLM(oc, st, bf16_acts=True)
    n_match = n_all = 0
    for r in list(prompts)[:3]:                         # the oracle is slow at this vocabulary
        ref = ar_generate(lm, prompts[r], len(got[r]))
        seq, cache = list(prompts[r]), lm.new_cache()
        logits = lm.forward(seq, 0, cache)[-1]
        for i, want in enumerate(ref):
            if got[r][i] != want:                       # only legitimate at a near-tie of the oracle
                top2 = torch.topk(logits, 2).values
                assert (top2[0] - top2[1]).item() / logits.abs().max().item() < 0.03, (r, i)
                break
            n_match += 1
            logits = lm.forward([want], len(seq), cache)[-1]
            seq.append(want)
        n_all += len(ref)
    assert n_match >= 0.6 * n_all, (n_match, n_all)


def test_prompts_prefilled_together_give_the_streams_of_one_at_a_time():
    """Batched prefill (add_requests) on the CPU build: ragged prompts, same streams as
    add_request prompt by prompt."""
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64, ffn=256,
                      vocab=256, max_pos=128)
    w = _product(oc, _bf(random_state(oc, seed=3)))
    d = _product(oc, _bf(random_state(oc, seed=4)))
    rng = np.random.default_rng(1)
    prompts = [[int(t) for t in rng.integers(0, oc.vocab, n)] for n in (4, 21, 1)]

    def run(batched):
        eng = SpecEngine(w, d, EngineConfig(k=2, max_batch=3, max_seq_len=64, split_tokens=128))
        outs = ([[f] for f in eng.add_requests(prompts)] if batched else
                [[eng.add_request(i, p)] for i, p in enumerate(prompts)])
        for _ in range(4):
            for o, new in zip(outs, eng.step()):
                o.extend(new)
        return [o[:5] for o in outs]

    assert run(True) == run(False)


def _policy_run(SpecEngine, EngineConfig, w, d, prompts, n_new, **kw):
    """Three sequences; the second leaves half-way and another prompt takes its place."""
    eng = SpecEngine(w, d, EngineConfig(max_batch=3, max_seq_len=140, split_tokens=128, **kw))
    outs = {i: [eng.add_request(i, p)] for i, p in enumerate(prompts[:3])}
    owner = [0, 1, 2]
    modes, swapped = [], False
    while min(len(outs[i]) for i in owner) < n_new:
        for slot, new in enumerate(eng.step()):
            outs[owner[slot]].extend(new)
        modes.append(eng._spec_disabled())
        if not swapped and len(outs[1]) >= n_new // 2:
            swapped = True
            eng.remove_request(1)                 # the last sequence moves into slot 1
            owner = [0, 2]
            outs[3] = [eng.add_request(2, prompts[3])]
            owner.append(3)
    return {i: o[:n_new] for i, o in outs.items()}, modes, eng


def test_the_speculation_policy_changes_no_greedy_stream():
    """spec_policy="auto" with a probe every 3 steps: the engine keeps switching between
    speculative steps and plain target steps that leave the draft out; the draft's cache is
    caught up in between (also for a one-token prompt and a sequence admitted mid-way).  Every
    stream is the autoregressive one, both kinds of step ran, and with draft == target nearly
    every proposal is accepted (the draft's cache is intact after every catch-up)."""
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64, ffn=256,
                      vocab=256, max_pos=160)
    w = _product(oc, _bf(random_state(oc, seed=3)))
    rng = np.random.default_rng(5)
    prompts = [[int(t) for t in rng.integers(0, oc.vocab, n)] for n in (9, 1, 30, 3)]
    want, _, _ = _policy_run(SpecEngine, EngineConfig, w, None, prompts, 40, k=0)
    got, modes, eng = _policy_run(SpecEngine, EngineConfig, w, w, prompts, 40, k=3, spec_policy="auto",
                                  policy_probe_every=3, policy_probe_steps=2)
    assert got[0] == want[0] and got[2] == want[2] and got[3][:20] == want[3][:20]
    assert got[1][:len(want[1]) // 2] == want[1][:len(want[1]) // 2]
    assert any(modes) and not all(modes)
    m = eng.metrics()
    assert m["draft"] > 0 and m["accepted"] >= 0.9 * m["draft"], m
    st = eng.policy_state()
    assert all(v["t_spec"] and v["t_ar"] for v in st.values() if v["t_spec"] is not None), st


def test_token_trees_on_the_cpu_build():
    """Multi-candidate verification through the CPU build: the host's tree construction and
    lookup equal the oracle's, and the stream is the autoregressive one."""
    from oracle import tree as otree
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64, ffn=256,
                      vocab=256, max_pos=160)
    w = _product(oc, _bf(random_state(oc, seed=3)))
    rng = np.random.default_rng(2)
    base = [int(t) for t in rng.integers(0, oc.vocab, 10)]
    prompt = base + base[:5] + [7, 8] + base[:6]

    def run(**kw):
        eng = SpecEngine(w, None, EngineConfig(max_batch=1, max_seq_len=100, split_tokens=128, **kw))
        out = [eng.add_request(0, prompt)]
        while len(out) < 24:
            out += eng.step()[0]
        return out[:24], eng

    ar, _ = run(k=0)
    got, eng = run(k=3, ngram_max=3, ngram_min=1, tree_candidates=3)
    assert got == ar and eng.metrics()["accepted"] > 0
    hist = prompt + got
    cands = eng._lookup_candidates(hist)
    assert cands == otree.lookup_candidates(hist, 3, 3, 1, 3)
    eng._upload_trees([cands])
    t, p, d = otree.build_tree(cands, 3, eng.tree_nodes)
    n = len(t)
    assert eng.tree_len[0].item() == n
    assert eng.tree_tok[0, 1:n].tolist() == t[1:] and eng.tree_parent[0, :n].tolist() == p
    assert eng.tree_depth[0, :n].tolist() == d
    assert eng.tree_mask[0, :n].tolist() == otree.ancestor_masks(p)


def test_draft_grown_trees_on_the_cpu_build():
    """draft_tree_branches: the draft's b best first tokens, each continued greedily to depth k
    by tree-masked draft passes, verified in one target pass.  The stream is the autoregressive
    one; with draft == target the first branch IS the target's continuation, so every step accepts
    k tokens (which needs the draft's cache compacted like the target's); the tokens of the tree
    equal what the draft model gives for each branch's prefix on its own (sd_topk_tokens picks
    the b largest in order, the level passes see their own branch only)."""
    from vllm_omni_draft_amd import ops
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    oc = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64, ffn=256,
                      vocab=256, max_pos=160)
    w = _product(oc, _bf(random_state(oc, seed=3)))
    d = _product(oc, _bf(random_state(oc, seed=4)))
    rng = np.random.default_rng(8)
    prompts = [[int(t) for t in rng.integers(0, oc.vocab, n)] for n in (7, 1, 19)]

    def run(draft, n_new=30, **kw):
        eng = SpecEngine(w, draft, EngineConfig(max_batch=3, max_seq_len=120, split_tokens=128, **kw))
        outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
        steps = 0
        while min(len(o) for o in outs) < n_new:
            for o, new in zip(outs, eng.step()):
                o.extend(new)
            steps += 1
        return [o[:n_new] for o in outs], steps, eng

    ar, _, _ = run(None, k=0)
    got, steps, eng = run(w, k=3, draft_tree_branches=3)
    assert eng.tree and eng.tree_nodes == 9 and got == ar
    m = eng.metrics()
    assert m["accepted"] == 3 * 3 * steps, m             # the first branch stands whole, every step
    got, steps_tree, eng = run(d, k=3, draft_tree_branches=3)
    assert got == ar
    _, steps_chain, _ = run(d, k=3)
    assert steps_tree <= steps_chain                      # branch 0 is the chain the draft proposes alone
    # the tree of the last step, branch by branch, against the draft run on that branch's prefix
    eng.prepare_step()
    root_len = eng.seq_lens[:3].tolist()
    eng._enqueue_step()
    tok = eng.tree_tok[:3].tolist()
    from vllm_omni_draft_amd.model_runner import DecoderModel, PagedKVCache
    model = DecoderModel(d, 160, 128)
    for b in range(3):
        # (the engine's own stream up to the root)
        prefix = (prompts[b] + eng.gen_tokens[b, :eng.gen_counts[b].item()].tolist())[:root_len[b]]
        assert prefix[-1] == tok[b][0]
        kv = PagedKVCache(oc, PagedKVCache.pages_for(oc, 1, 160), 1, 160, "cpu")
        kv.ensure_capacity(0, 160)

        def next_logits(seq):
            t = torch.tensor(seq, dtype=torch.int32)
            lg = torch.empty(len(seq), oc.vocab)
            model.forward(kv, t, 0, 0, torch.zeros(1, dtype=torch.int32), 0, 1, len(seq), lg, 160)
            return lg[-1]
        first = torch.sort(next_logits(prefix), descending=True, stable=True).indices[:3].tolist()
        assert [tok[b][1 + c] for c in range(3)] == first
        for c in range(3):
            seq = prefix + [first[c]]
            for depth in (2, 3):
                nxt = int(next_logits(seq).argmax())
                assert tok[b][1 + (depth - 1) * 3 + c] == nxt, (b, c, depth)
                seq.append(nxt)
    with pytest.raises(ValueError):
        run(w, k=3, draft_tree_branches=3, tree_candidates=2)
    with pytest.raises(ValueError):
        run(None, k=3, draft_tree_branches=3, ngram_max=2)
    with pytest.raises(ValueError):
        run(w, k=8, draft_tree_branches=4)               # 32 nodes: more than a verify pass holds


def test_tree_sampling_walk_matches_the_oracle():
    """sd_tree_accept_sample of the CPU build against the oracle (ids, rows and counters equal),
    and a sampling engine with prompt-lookup trees steps through it."""
    from oracle import tree as otree
    from vllm_omni_draft_amd import ops
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    rng = np.random.default_rng(9)
    B, k, N, V = 8, 3, 8, 64
    q = N + 1
    tok = torch.zeros(B, q, dtype=torch.int32)
    par = torch.zeros(B, q, dtype=torch.int32)
    lens = torch.ones(B, dtype=torch.int32)
    probs = rng.dirichlet(np.full(V, 0.05), size=(B, q)).astype(np.float32)
    trees = []
    for b in range(B):
        pool = rng.integers(0, V, 4)
        cands = [[int(t) for t in rng.choice(pool, rng.integers(1, k + 2))] for _ in range(rng.integers(0, 4))]
        t, p, _ = otree.build_tree(cands, k, N)
        if b == 3:      # a tree made elsewhere with the same token on two siblings (rows 1 and 2)
            t, p = [0, 7, 7, 8], [0, 0, 0, 0]
        trees.append((t, p))
        n = len(t)
        tok[b, :n], par[b, :n], lens[b] = torch.tensor(t), torch.tensor(p), n
        for r in range(1, n):
            if rng.random() < 0.7:
                probs[b, p[r], t[r]] += rng.choice([0.5, 2.0, 10.0])
    probs = (probs / probs.sum(axis=2, keepdims=True, dtype=np.float64)).astype(np.float32)
    ua, uf = rng.random((B, q), dtype=np.float32), rng.random(B, dtype=np.float32)
    # the duplicate sibling: token 7 is rejected at row 1; row 2 carries it again and must be
    # skipped whatever its uniform (its mass is gone); row 3 holds all that is left and stands
    probs[3, 0] = 0.0
    probs[3, 0, 7], probs[3, 0, 8] = 0.4, 0.6
    ua[3, 1], ua[3, 2], ua[3, 3] = 0.9, 0.0, 0.99
    out = torch.full((B, k + 1), -7, dtype=torch.int64)
    path = torch.full((B, k), -7, dtype=torch.int32)
    npath = torch.zeros(B, dtype=torch.int32)
    counters = torch.zeros(3, dtype=torch.int64)
    ops.tree_accept_sample(torch.from_numpy(probs).view(B * q, V), tok, par, lens, k, torch.from_numpy(ua),
                           torch.from_numpy(uf), out, path, npath, counters)
    acc = 0
    for b in range(B):
        t, p = trees[b]
        want_out, want_path = otree.accept_sample(probs[b], t, p, len(t), k, ua[b], uf[b])
        assert out[b].tolist() == want_out + [-1] * (k + 1 - len(want_out)), b
        assert path[b].tolist() == want_path + [-1] * (k - len(want_path)), b
        acc += len(want_path)
    assert path[3].tolist()[0] == 3 and out[3].tolist()[0] == 8
    assert counters[0].item() == acc and acc > 0
    oc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64, ffn=128,
                      vocab=32, max_pos=200)
    w = _product(oc, _bf(random_state(oc, seed=4)))
    eng = SpecEngine(w, None, EngineConfig(k=3, max_batch=2, max_seq_len=150, split_tokens=128,
                                           temperature=1.0, seed=3, ngram_max=2, ngram_min=1,
                                           tree_candidates=3))
    seqs = [[1, 2, 3, 1, 2], [5, 5, 5]]
    for b, s in enumerate(seqs):
        s.append(eng.add_request(b, s))
    for _ in range(20):
        for b, new in enumerate(eng.step()):
            assert 1 <= len(new) <= 4 and all(0 <= t < 32 for t in new)
            seqs[b].extend(new)
    assert eng.metrics()["emitted"] >= 40


@pytest.mark.parametrize("seed", [0, 2, 3, 105])
def test_random_v1_schedules_on_the_cpu_build(seed, monkeypatch):
    """The randomised V1 schedule of tests/test_gpu_fuzz.py (prompts whole and in pieces,
    preemption and return, finishes anywhere, log-probabilities) through the CPU build: every
    stream is the one the request gets alone."""
    import tests.test_gpu_fuzz as fuzz
    monkeypatch.setattr(fuzz, "_weights", lambda oc, s: _product(oc, _bf(random_state(oc, seed=s))))
    fuzz.test_random_v1_schedules_with_pieces_and_preemption(seed, False, "wide", 0.8 if seed == 3 else 0.0)


def test_logit_masks_on_the_cpu_build():
    from vllm_omni_draft_amd import ops
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    x = torch.randn(6, 40)
    bias = torch.zeros(3, 40)
    bias[1, ::2] = float("-inf")
    bias[2, 5] = 2.0
    on = torch.tensor([0, 1, 1], dtype=torch.int32)
    want = x.clone()
    want[2:4] += bias[1]
    want[4:6] += bias[2]
    ops.add_logit_bias(x, bias, on, rows_per_seq=2)
    assert torch.equal(x, want)
    oc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64, ffn=128,
                      vocab=64, max_pos=100)
    w = _product(oc, _bf(random_state(oc, seed=4)))
    eng = SpecEngine(w, w, EngineConfig(k=2, max_batch=1, max_seq_len=80, split_tokens=128))
    out = [eng.add_request(0, [1, 2, 3], bias={"allowed_token_ids": list(range(0, 64, 4))})]
    for _ in range(8):
        out += eng.step()[0]
    assert all(t % 4 == 0 for t in out)
