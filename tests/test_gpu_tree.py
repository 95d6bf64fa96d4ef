"""Token trees (multi-candidate verification) on the GPU against the CPU oracle: the masked
attention kernel, the acceptance walk, the cache compaction, a tree pass of a whole model
against chain passes of the same tokens, and the engine's streams against plain
autoregressive decoding."""
import numpy as np
import pytest
import torch

from oracle import tree as otree
from oracle.attention import unpack_pages
from oracle.models import OracleConfig, random_state
from tests.test_gpu_attention import _setup
from tests.test_gpu_model import MODELS, _bf, _engine, _generate, _product

pytestmark = pytest.mark.gpu


def _i64(masks):
    return torch.tensor([m - (1 << 64) if m >= (1 << 63) else m for m in masks], dtype=torch.int64)


TREES = [
    # parents of rows 1.. (row 0 = root)
    [0, 1, 2, 3, 4],                      # a chain: the tree kernel must equal the causal one
    [0, 0, 1, 1, 2, 3, 3, 5],             # branching at several depths
    [0, 0, 0, 0, 0, 0, 0],                # a star
    [0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14],   # five chains of three (15 nodes)
]


@pytest.mark.parametrize("Hq,Hkv,D,ctx_lens,split", [
    (4, 2, 64, [37], 128),
    (32, 8, 128, [300, 12], 128),         # several splits, ragged batch, partials + merge
    (32, 8, 128, [1100], 256),
    (12, 12, 64, [250, 255], 128),        # new tokens straddle a split and a page boundary
])
@pytest.mark.parametrize("parents", TREES, ids=["chain", "branchy", "star", "five-chains"])
def test_tree_attention_matches_oracle(Hq, Hkv, D, ctx_lens, split, parents):
    from vllm_omni_draft_amd import ops
    par = [0] + parents
    q_len, B = len(par), len(ctx_lens)
    if q_len * (Hq // Hkv) > 64:
        pytest.skip("more rows than one workgroup column holds")
    masks = otree.ancestor_masks(par)
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=sum(ctx_lens) + q_len)
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    tm = _i64(masks * B).cuda()
    max_ctx = max(ctx_lens) + q_len
    out = ops.paged_attention_tree(q, kc.cuda(), vc.cuda(), bt.cuda(), seq_lens, 0, tm, B, q_len, Hq,
                                   Hkv, D, D ** -0.5, max_ctx, split)
    torch.cuda.synchronize()
    out = out.float().cpu().view(B, q_len, Hq, D)
    for b in range(B):
        ref = otree.tree_attention(qs[b], ks[b], vs[b], ctx_lens[b], masks, D ** -0.5)
        torch.testing.assert_close(out[b], ref, rtol=2e-2, atol=2e-2)
    if parents == TREES[0]:   # a chain's masks are the causal mask: same bits as the plain kernel
        plain = ops.paged_attention(q, kc.cuda(), vc.cuda(), bt.cuda(), seq_lens, 0, B, q_len, Hq, Hkv,
                                    D, D ** -0.5, max_ctx, split)
        assert torch.equal(plain.float().cpu().view(B, q_len, Hq, D), out)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (4, 2, 64)])
@pytest.mark.parametrize("root", [5, 20, 29, 40, 250, 1044, 1050])
def test_tree_rows_equal_the_same_rows_laid_out_as_a_chain(Hq, Hkv, D, root):
    """Every root-to-leaf chain of a 15-node tree, with the new tokens crossing tile, page and
    split boundaries and (4 heads per KV head) two query tiles: a row of the tree pass equals,
    to two bf16 steps, the row of the plain kernel over a cache that holds the chain contiguously
    (only the summation order may differ).  With short contexts a single wrongly seen or missed
    key would be ten times that."""
    from vllm_omni_draft_amd import ops
    from tests.gpu_util import pack_cache_fast
    par = [0, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9, 0, 11, 12, 13, 14]
    masks, q_len, split = otree.ancestor_masks(par), 16, 256
    g = torch.Generator().manual_seed(root)
    n, pages = root + q_len, 2048 // 16
    bt = torch.randperm(pages, generator=g).to(torch.int32).view(1, -1)
    k = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(q_len, Hq, D, generator=g).to(torch.bfloat16)
    kc = torch.zeros(pages, Hkv, D * 16, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    pack_cache_fast(kc, vc, k, v, bt[0])
    sl = torch.tensor([root], dtype=torch.int32).cuda()
    out = ops.paged_attention_tree(q.reshape(q_len, Hq * D).cuda(), kc.cuda(), vc.cuda(), bt.cuda(), sl, 0,
                                   _i64(masks).cuda(), 1, q_len, Hq, Hkv, D, D ** -0.5, 2048, split)
    out = out.cpu().view(q_len, Hq, D).float()
    for chain in ([1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [11, 12, 13, 14, 15]):
        rows = [0] + chain
        k2, v2 = k.clone(), v.clone()
        for d, r in enumerate(rows):
            k2[root + d], v2[root + d] = k[root + r], v[root + r]
        kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
        pack_cache_fast(kc2, vc2, k2, v2, bt[0])
        ref = ops.paged_attention(q[rows].reshape(6, Hq * D).cuda(), kc2.cuda(), vc2.cuda(), bt.cuda(), sl, 0,
                                  1, 6, Hq, Hkv, D, D ** -0.5, 2048, split).cpu().view(6, Hq, D).float()
        torch.testing.assert_close(out[rows], ref, rtol=2 ** -7, atol=2 ** -9)


def test_tree_attention_rejects_bad_arguments():
    from vllm_omni_draft_amd import _lib, ops
    kc, vc, bt, ks, vs, qs = _setup(1, [20], 4, 4, 2, 64, seed=1)
    q = torch.cat(qs).reshape(4, 4 * 64).cuda()
    sl = torch.tensor([20], dtype=torch.int32).cuda()
    with pytest.raises(_lib.SdError):      # int32 masks
        ops.paged_attention_tree(q, kc.cuda(), vc.cuda(), bt.cuda(), sl, 0,
                                 torch.zeros(4, dtype=torch.int32).cuda(), 1, 4, 4, 2, 64, 0.125, 24, 128)
    with pytest.raises(_lib.SdError):      # too few masks
        ops.paged_attention_tree(q, kc.cuda(), vc.cuda(), bt.cuda(), sl, 0,
                                 torch.zeros(3, dtype=torch.int64).cuda(), 1, 4, 4, 2, 64, 0.125, 24, 128)


def test_accept_walk_and_prepare_match_oracle():
    from vllm_omni_draft_amd import ops
    rng = np.random.default_rng(3)
    B, k, N, V = 9, 4, 11, 7            # a small vocabulary makes matches and duplicates frequent
    q = N + 1
    tok = torch.zeros(B, q, dtype=torch.int32)
    par = torch.zeros(B, q, dtype=torch.int32)
    dep = torch.zeros(B, q, dtype=torch.int32)
    lens = torch.ones(B, dtype=torch.int32)
    tgt = torch.tensor(rng.integers(0, V, (B, q)), dtype=torch.int64)
    trees = []
    for b in range(B):
        cands = [list(rng.integers(0, V, rng.integers(1, k + 2))) for _ in range(rng.integers(0, 5))]
        if b == 0:
            cands = []                                     # no proposals at all
        if b == 1:                                         # the target follows one chain to the end
            cands = [[1, 2, 3, 4], [1, 2, 5, 5]]
        t, p, d = otree.build_tree(cands, k, N)
        trees.append((t, p))
        n = len(t)
        tok[b, :n], par[b, :n], dep[b, :n] = torch.tensor(t), torch.tensor(p), torch.tensor(d)
        lens[b] = n
        if b == 1:
            tgt[b, 0], tgt[b, 1], tgt[b, 2], tgt[b, 3], tgt[b, 4] = 1, 2, 3, 4, 6
    out = torch.full((B, k + 1), -7, dtype=torch.int64).cuda()
    path = torch.full((B, k), -7, dtype=torch.int32).cuda()
    npath = torch.zeros(B, dtype=torch.int32).cuda()
    counters = torch.zeros(3, dtype=torch.int64).cuda()
    ops.tree_accept_greedy(tgt.view(-1).cuda(), tok.cuda(), par.cuda(), lens.cuda(), k, out, path, npath,
                           counters)
    torch.cuda.synchronize()
    acc = emit = depth = 0
    for b in range(B):
        t, p = trees[b]
        want_path, want_out = otree.accept_greedy(tgt[b].tolist(), t, p, k)
        assert npath[b].item() == len(want_path)
        assert path[b].tolist() == want_path + [-1] * (k - len(want_path))
        assert out[b].tolist() == want_out + [-1] * (k + 1 - len(want_out))
        acc += len(want_path)
        emit += len(want_out)
        d = [0] * len(t)
        for r in range(1, len(t)):
            d[r] = d[p[r]] + 1
        depth += min(max(d), k)
    assert counters.tolist() == [acc, emit, depth]
    assert out[1].tolist() == [1, 2, 3, 4, 6]
    # prepare: the root token and the rows' rotary positions
    tok_buf = torch.tensor(rng.integers(0, 99, (B, k + 2)), dtype=torch.int32).cuda()
    seq_lens = torch.tensor(rng.integers(5, 500, B), dtype=torch.int32).cuda()
    d_tok, rope = tok.cuda(), torch.zeros(B * q, dtype=torch.int32).cuda()
    ops.tree_prepare(tok_buf, 1, d_tok, dep.cuda(), seq_lens, -1, rope)
    torch.cuda.synchronize()
    assert torch.equal(d_tok[:, 0], tok_buf[:, 1]) and torch.equal(d_tok[:, 1:].cpu(), tok[:, 1:])
    assert torch.equal(rope.view(B, q).cpu(), (seq_lens.cpu()[:, None] - 1 + dep))


@pytest.mark.parametrize("Hkv,D", [(2, 64), (8, 128)])
def test_kv_compact_moves_the_accepted_rows(Hkv, D):
    from vllm_omni_draft_amd import ops
    g = torch.Generator().manual_seed(5)
    B, L, k, pages_per_seq = 3, 2, 4, 6
    num_pages = B * pages_per_seq + 2
    bt = torch.randperm(num_pages, generator=g)[:B * pages_per_seq].view(B, pages_per_seq).to(torch.int32)
    kc = torch.randn(L, num_pages, Hkv, D * 16, generator=g).to(torch.bfloat16)
    vc = torch.randn(L, num_pages, Hkv, D * 16, generator=g).to(torch.bfloat16)
    seq_lens = torch.tensor([30, 16, 47], dtype=torch.int32)      # roots at slots 29, 15, 46
    path = torch.tensor([[2, 5, 9, -1], [1, 2, 3, 4], [-1, -1, -1, -1]], dtype=torch.int32)
    npath = torch.tensor([3, 4, 0], dtype=torch.int32)
    n_tok = [int(s) + 12 for s in seq_lens]
    before = [[unpack_pages(kc[l].view(torch.int16).numpy(), vc[l].view(torch.int16).numpy(), bt[b],
                            n_tok[b], Hkv, D) for b in range(B)] for l in range(L)]
    kd, vd = kc.cuda(), vc.cuda()
    ops.kv_compact(kd, vd, kd[0].numel() * 2, L, bt.cuda(), seq_lens.cuda(), -1, path.cuda(),
                   npath.cuda(), Hkv, D)
    torch.cuda.synchronize()
    for l in range(L):
        for b in range(B):
            k_new, v_new = unpack_pages(kd[l].cpu().view(torch.int16).numpy(),
                                        vd[l].cpu().view(torch.int16).numpy(), bt[b], n_tok[b], Hkv, D)
            k_old, v_old = before[l][b]
            want_k, want_v = k_old.copy(), v_old.copy()
            root = int(seq_lens[b]) - 1
            for j in range(int(npath[b])):
                want_k[root + j + 1] = k_old[root + int(path[b, j])]
                want_v[root + j + 1] = v_old[root + int(path[b, j])]
            assert np.array_equal(k_new, want_k) and np.array_equal(v_new, want_v)


def _chain_logits(model, kv_factory, prefix, chain, max_ctx):
    """Logits after the last token of prefix + chain, by plain passes over a fresh cache."""
    kv = kv_factory()
    toks = torch.tensor(prefix + chain, dtype=torch.int32).cuda()
    zero = torch.zeros(1, dtype=torch.int32).cuda()
    n = len(prefix) + len(chain)
    lg = torch.empty(1, model.cfg.vocab, dtype=torch.float32).cuda()
    if n > 1:
        model.forward(kv, toks, 0, 0, zero, 0, 1, n - 1, None, max_ctx)
    model.forward(kv, toks, 0, n - 1, zero, n - 1, 1, 1, lg, max_ctx)
    torch.cuda.synchronize()
    return lg[0].cpu()


@pytest.mark.parametrize("wide", [False, True])
def test_tree_pass_rows_equal_chain_passes(wide):
    """One tree pass of a model: every row's logits equal (to rounding: the ancestors sit at
    other cache slots than in a chain, so sums run in another order) the logits of the same
    root-to-node token path scored as a chain, and the arg-max tokens agree."""
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    oc = MODELS["llama"]
    w = _product(oc, _bf(random_state(oc, seed=21)))
    max_ctx = 128
    model = DecoderModel(w, max_ctx, 128, wide=wide)
    prefix = [5, 9, 200, 31, 7, 44, 3, 18, 90, 12, 77]
    cands = [[10, 11, 12], [10, 13], [20, 21, 22], [10, 11, 14]]
    toks, par, dep = otree.build_tree(cands, 3, 15)
    q = len(toks)
    masks = otree.ancestor_masks(par)

    def fresh():
        kv = PagedKVCache(w.cfg, 16, 1, max_ctx, "cuda")
        kv.ensure_capacity(0, max_ctx)
        return kv

    kv = fresh()
    n = len(prefix)
    ptoks = torch.tensor(prefix, dtype=torch.int32).cuda()
    zero = torch.zeros(1, dtype=torch.int32).cuda()
    model.forward(kv, ptoks, 0, 0, zero, 0, 1, n - 1, None, max_ctx)          # context: tokens 0..n-2
    tree_tok = torch.tensor([prefix[-1]] + toks[1:], dtype=torch.int32).cuda()
    rope = torch.tensor([n - 1 + d for d in dep], dtype=torch.int32).cuda()
    lens = torch.tensor([n], dtype=torch.int32).cuda()
    lg = torch.empty(q, w.cfg.vocab, dtype=torch.float32).cuda()
    model.forward(kv, tree_tok, q, 0, lens, -1, 1, q, lg, max_ctx, rope_pos=rope,
                  tree_mask=_i64(masks).cuda())
    torch.cuda.synchronize()
    lg = lg.cpu()
    for r in range(q):
        chain, cur = [], r
        while cur != 0:
            chain.append(toks[cur])
            cur = par[cur]
        ref = _chain_logits(model, fresh, prefix, chain[::-1], max_ctx)
        scale = ref.abs().max().item()
        assert (lg[r] - ref).abs().max().item() < 0.02 * scale, r
        assert lg[r].argmax().item() == ref.argmax().item(), r


def _repetitive_prompts(vocab, seed=2):
    rng = np.random.default_rng(seed)
    base = [int(t) for t in rng.integers(0, vocab, 12)]
    alt = base[:6] + [int(t) for t in rng.integers(0, vocab, 6)]
    return [base + alt + base[:7], alt * 2 + base + alt[:4], [3, 3, 3, 3]]


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("rows", ["narrow", "wide"])
def test_tree_engine_streams_equal_autoregressive(use_graph, rows):
    """Prompt-lookup trees (several earlier matches merged into one tree per sequence): the
    emitted streams are the target's autoregressive streams, steps accept at least as many
    tokens as the single-chain lookup, and after every step the cache holds the accepted path
    (what the next step's rows attend), which a wrong compaction would break."""
    oc = MODELS["llama"]
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    prompts = _repetitive_prompts(oc.vocab)
    n_new = 40
    ar = [_generate(_engine(tw, None, 0, rows=rows), [p], n_new)[0] for p in prompts]

    def run(eng):
        outs, steps = [[eng.add_request(i, p)] for i, p in enumerate(prompts)], 0
        while min(len(o) for o in outs) < n_new:
            for o, new in zip(outs, eng.step()):
                o.extend(new)
            steps += 1
        return [o[:n_new] for o in outs], steps

    chain = _engine(tw, None, 3, max_batch=3, ngram_max=3, ngram_min=1, use_graph=use_graph, rows=rows)
    got_chain, steps_chain = run(chain)
    assert got_chain == ar
    eng = _engine(tw, None, 3, max_batch=3, ngram_max=3, ngram_min=1, tree_candidates=4,
                  use_graph=use_graph, rows=rows)
    assert eng.tree and eng.tree_nodes == 12
    got, steps = run(eng)
    assert got == ar
    m = eng.metrics()
    # the chain is the tree's first branch: a tree step never accepts less from the same state
    assert steps <= steps_chain and m["accepted"] > 0 and m["emitted"] - m["accepted"] == 3 * steps


def test_external_candidate_sets_are_verified_as_a_tree():
    """Candidates from outside the engine: per sequence several chains.  With the target's own
    continuation hidden among wrong candidates every step accepts k tokens."""
    oc = MODELS["llama"]
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    prompt = [5, 9, 200, 31, 7]
    ar = _generate(_engine(tw, None, 0), [prompt], 41)[0]
    eng = _engine(tw, None, 3, external_proposals=True, tree_candidates=3)
    out = [eng.add_request(0, prompt)]
    steps = 0
    while len(out) < 37:
        nxt = ar[len(out):len(out) + 3]
        wrong = [(t + 1) % oc.vocab for t in nxt]
        out += eng.step(proposals=[[wrong, [nxt[0], wrong[1], nxt[2]], nxt]])[0]
        steps += 1
    assert out == ar[:len(out)] and len(out) == 1 + 4 * steps
    m = eng.metrics()
    assert m["accepted"] == 3 * steps and m["draft"] == 3 * steps
    # a flat list is one chain; no proposals: one token
    assert len(eng.step(proposals=[ar[len(out):len(out) + 3]])[0]) == 4
    assert len(eng.step(proposals=[[]])[0]) == 1


@pytest.mark.parametrize("V", [64, 1000, 128256])
def test_sampling_walk_matches_oracle(V):
    """sd_tree_accept_sample against the oracle's restatement of multi-step speculative sampling
    with point-mass proposals: emitted tokens, accepted rows and counters are equal bit for bit,
    over trees with several children per node, rows whose mass sits on the children (accepted),
    rows that reject every child (draw from what is left), chains accepted to the depth limit."""
    from vllm_omni_draft_amd import ops
    rng = np.random.default_rng(V)
    B, k, N = 12, 4, 11
    q = N + 1
    tok = torch.zeros(B, q, dtype=torch.int32)
    par = torch.zeros(B, q, dtype=torch.int32)
    lens = torch.ones(B, dtype=torch.int32)
    probs = rng.dirichlet(np.full(V, 0.05 if V < 2000 else 0.002), size=(B, q)).astype(np.float32)
    trees = []
    for b in range(B):
        pool = rng.integers(0, V, 5)
        cands = [[int(t) for t in rng.choice(pool, rng.integers(1, k + 2))] for _ in range(rng.integers(0, 5))]
        if b == 0:
            cands = []
        if b == 1:
            cands = [[1, 2, 3, 4], [1, 2, 5, 5], [9, 9]]
        t, p, _ = otree.build_tree(cands, k, N)
        if b == 3:      # a tree made elsewhere with the same token on two siblings (rows 1 and 2)
            t, p = [0, 7, 7, 8], [0, 0, 0, 0]
        trees.append((t, p))
        n = len(t)
        tok[b, :n], par[b, :n], lens[b] = torch.tensor(t), torch.tensor(p), n
        # put mass on the children of some rows, so that walks go deep
        for r in range(1, n):
            if rng.random() < 0.6:
                probs[b, p[r], t[r]] += rng.choice([0.5, 2.0, 10.0])
        if b == 1:                       # one chain stands to the depth limit, whatever the uniforms
            for row, x in ((0, 1), (1, 2), (2, 3), (3, 4)):
                probs[b, row] = 0.0
                probs[b, row, x] = 1.0
        if b == 2 and n > 1:             # the root's children hold ALL of its mass, the first one none
            kids = [r for r in range(1, n) if p[r] == 0]
            probs[b, 0] = 0.0
            for r in kids[1:] or kids:
                probs[b, 0, t[r]] = 1.0
    probs = (probs / probs.sum(axis=2, keepdims=True, dtype=np.float64)).astype(np.float32)
    ua = rng.random((B, q), dtype=np.float32)
    uf = rng.random(B, dtype=np.float32)
    # the duplicate sibling: token 7 is rejected at row 1; row 2 carries it again and must be
    # skipped whatever its uniform (its mass is gone); row 3 holds all that is left and stands
    probs[3, 0] = 0.0
    probs[3, 0, 7], probs[3, 0, 8] = 0.4, 0.6
    ua[3, 1], ua[3, 2], ua[3, 3] = 0.9, 0.0, 0.99
    out = torch.full((B, k + 1), -7, dtype=torch.int64).cuda()
    path = torch.full((B, k), -7, dtype=torch.int32).cuda()
    npath = torch.zeros(B, dtype=torch.int32).cuda()
    counters = torch.zeros(3, dtype=torch.int64).cuda()
    ops.tree_accept_sample(torch.from_numpy(probs).view(B * q, V).cuda(), tok.cuda(), par.cuda(), lens.cuda(),
                           k, torch.from_numpy(ua).cuda(), torch.from_numpy(uf).cuda(), out, path, npath,
                           counters)
    torch.cuda.synchronize()
    acc = emit = depth = 0
    for b in range(B):
        t, p = trees[b]
        want_out, want_path = otree.accept_sample(probs[b], t, p, len(t), k, ua[b], uf[b])
        assert out[b].tolist() == want_out + [-1] * (k + 1 - len(want_out)), b
        assert path[b].tolist() == want_path + [-1] * (k - len(want_path)), b
        assert npath[b].item() == len(want_path)
        acc, emit = acc + len(want_path), emit + len(want_out)
        d = [0] * len(t)
        for r in range(1, len(t)):
            d[r] = d[p[r]] + 1
        depth += min(max(d), k)
    assert counters.tolist() == [acc, emit, depth]
    assert out[1].tolist()[:4] == [1, 2, 3, 4] and npath[1].item() == 4
    assert path[3].tolist()[0] == 3 and out[3].tolist()[0] == 8      # the duplicate sibling was skipped
    assert acc >= 6                      # walks did descend beyond the forced chain


def test_tree_sampling_follows_the_target_distribution():
    """Prompt-lookup trees under sampling (T = 1): every emitted token, accepted candidate or final
    draw, is distributed as the ORACLE's target distribution given its predecessor (the
    tolerance stated for the sampling path), while candidates are accepted and rejected."""
    from tests.test_gpu_model import _check_transitions, _markov_pair, _transition_matrix
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    cfg, (ts, ds) = _markov_pair()
    P = _transition_matrix(cfg, ts, 1.0)
    Q = _transition_matrix(cfg, ds, 1.0)
    tw = _product(cfg, ts)
    B = 4
    eng = SpecEngine(tw, None, EngineConfig(k=3, max_batch=B, max_seq_len=900, split_tokens=128,
                                            temperature=1.0, seed=13, ngram_max=2, ngram_min=1,
                                            tree_candidates=4))
    assert eng.tree and eng.sampling
    counts = np.zeros((cfg.vocab, cfg.vocab))
    rounds = 0
    while counts.sum() < 21000:
        eng.reset()
        seqs = [[(5 * b + rounds) % cfg.vocab] for b in range(B)]
        for b in range(B):
            seqs[b].append(eng.add_request(b, seqs[b]))
        for _ in range(200):
            for b, new in enumerate(eng.step()):
                assert 1 <= len(new) <= 4 and all(0 <= t < cfg.vocab for t in new)
                seqs[b].extend(new)
        for sq in seqs:
            np.add.at(counts, (sq[:-1], sq[1:]), 1)
        rounds += 1
    m = eng.metrics()
    assert m["accepted"] > 500 and m["emitted"] - m["accepted"] > 500    # both paths ran
    _check_transitions(counts, P)
    with pytest.raises(AssertionError):
        _check_transitions(counts, Q)


def test_tree_mode_refuses_what_it_cannot_do():
    oc = MODELS["llama"]
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    with pytest.raises(ValueError):
        _engine(tw, tw, 3, tree_candidates=2)                             # a draft model
    with pytest.raises(ValueError):                                       # typical acceptance walks chains
        _engine(tw, None, 3, ngram_max=2, tree_candidates=2, temperature=0.7,
                acceptance_method="typical_acceptance_sampler")
    ocfg = OracleConfig(**{**vars(MODELS["opt"])})
    ow = _product(ocfg, _bf(random_state(ocfg, seed=3)))
    with pytest.raises(ValueError):
        _engine(ow, None, 3, ngram_max=2, tree_candidates=2)              # learned positions


# --------------------------------------------------------------------------- draft-grown trees
@pytest.mark.parametrize("V", [50, 1000, 128256])
def test_topk_tokens_against_a_stable_sort(V):
    """sd_topk_tokens: the b largest entries of strided rows, largest first, equal values in
    index order (what a stable descending sort gives), written as int32 tokens at a column of a
    strided table; b = 1 is the arg-max with the lowest index among equals."""
    from vllm_omni_draft_amd import ops
    g = torch.Generator().manual_seed(V)
    R = 11
    x = torch.randn(R, V, generator=g)
    x[1] = torch.round(x[1] * 2) / 2                    # many exact ties
    x[2] = 0.0                                          # all equal: tokens 0..b-1
    x[3, V - 1], x[3, 0] = 9.0, 9.0
    x[4] = float("-inf")
    x[4, V // 2] = -3.0                                 # one finite entry, then -inf ties
    xd = x.cuda()
    for b, off, step in ((1, 0, 1), (3, 1, 2), (8, 2, 3), (4, 0, 5)):
        n = (R - off + step - 1) // step
        out = torch.full((n, 12), -5, dtype=torch.int32).cuda()
        ops.topk_tokens(xd, b, out, 2, row_offset=off, row_step=step)
        torch.cuda.synchronize()
        want = torch.sort(x[off::step], dim=1, descending=True, stable=True).indices[:, :b]
        assert out[:, 2:2 + b].cpu().tolist() == want.tolist(), (b, off, step)
        assert (out[:, :2] == -5).all() and (out[:, 2 + b:] == -5).all()
    # several consecutive rows per sequence in one launch (a level of a draft-grown tree: every
    # branch's arg-max): sequence s takes rows 1 + 4 s + r, r < 3, two tokens each
    out = torch.full((2, 12), -5, dtype=torch.int32).cuda()
    ops.topk_tokens(xd, 2, out, 1, row_offset=1, row_step=4, rows=3, B=2)
    torch.cuda.synchronize()
    want = torch.sort(x, dim=1, descending=True, stable=True).indices[:, :2]
    for s in range(2):
        assert out[s, 1:7].cpu().tolist() == want[1 + 4 * s:4 + 4 * s].reshape(-1).tolist()
        assert out[s, 0].item() == -5 and (out[s, 7:] == -5).all()
    with pytest.raises(Exception):
        ops.topk_tokens(xd, ops.TOPK_TOKENS_MAX + 1, torch.zeros(R, 12, dtype=torch.int32).cuda(), 0)
    with pytest.raises(Exception):
        ops.topk_tokens(xd, 1, torch.zeros(R, 12, dtype=torch.int32).cuda(), 0, row_step=2, rows=3)   # rows overlap


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("rows", ["narrow", "wide"])
def test_draft_grown_trees_streams_equal_autoregressive(use_graph, rows):
    """draft_tree_branches: b branches of depth k grown by the draft model and verified in one
    target pass.  Streams are the autoregressive ones; a tree step never accepts less than the
    chain the same draft proposes alone (branch 0 is that chain), so it takes no more steps; with
    draft == target branch 0 is the target's own continuation and every step accepts k tokens,
    which also needs the draft's cache compacted to the accepted path."""
    oc = MODELS["llama"]
    dc = OracleConfig(**{**vars(oc), "n_layers": 1})
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    dw = _product(dc, _bf(random_state(dc, seed=23)))
    rng = np.random.default_rng(4)
    prompts = [[int(t) for t in rng.integers(0, oc.vocab, n)] for n in (9, 1, 33)]
    n_new = 40
    ar = [_generate(_engine(tw, None, 0, rows=rows), [p], n_new)[0] for p in prompts]

    def run(eng):
        outs, steps = [[eng.add_request(i, p)] for i, p in enumerate(prompts)], 0
        while min(len(o) for o in outs) < n_new:
            for o, new in zip(outs, eng.step()):
                o.extend(new)
            steps += 1
        return [o[:n_new] for o in outs], steps

    kw = dict(max_batch=3, use_graph=use_graph, rows=rows, draft_rows="same")
    got_chain, steps_chain = run(_engine(tw, dw, 3, **kw))
    assert got_chain == ar
    eng = _engine(tw, dw, 3, draft_tree_branches=4, **kw)
    assert eng.tree and eng.tree_nodes == 12
    got, steps = run(eng)
    assert got == ar and steps <= steps_chain
    m = eng.metrics()
    assert m["emitted"] - m["accepted"] == 3 * steps
    same = _engine(tw, tw, 3, draft_tree_branches=3, **kw)
    got, steps = run(same)
    assert got == ar and same.metrics()["accepted"] == 9 * steps


def test_draft_grown_tree_holds_each_branchs_own_continuation():
    """The tree a step verifies, node by node: depth 1 holds the draft's b most likely next
    tokens in order, and every deeper node is the draft's arg-max after its own branch's prefix
    (computed here by chain passes of the draft over a fresh cache), for sequences whose roots
    sit at different cache positions."""
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    oc = MODELS["llama"]
    dc = OracleConfig(**{**vars(oc), "n_layers": 1})
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    dw = _product(dc, _bf(random_state(dc, seed=23)))
    prompts = [[5, 9, 200, 31, 7], list(range(40, 57))]
    nb, k = 4, 3
    eng = _engine(tw, dw, k, max_batch=2, draft_tree_branches=nb)
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    for _ in range(5):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    eng.prepare_step()
    root_len = eng.seq_lens[:2].tolist()
    with torch.cuda.stream(eng.stream):
        eng._enqueue_step()
    torch.cuda.synchronize()
    tok = eng.tree_tok[:2].tolist()
    model = DecoderModel(dw, 200, 128, wide=True)

    def fresh():
        kv = PagedKVCache(dc, PagedKVCache.pages_for(dc, 1, 200), 1, 200, "cuda")
        kv.ensure_capacity(0, 200)
        return kv

    def next_logits(seq):
        return _chain_logits(model, fresh, seq, [], 200)

    for b in range(2):
        prefix = (prompts[b] + outs[b])[:root_len[b]]
        assert prefix[-1] == tok[b][0]
        lg = next_logits(prefix)
        first = torch.sort(lg, descending=True, stable=True).indices[:nb].tolist()
        assert tok[b][1:1 + nb] == first
        for c in range(nb):
            seq = prefix + [first[c]]
            for depth in range(2, k + 1):
                lg = next_logits(seq)
                got = tok[b][1 + (depth - 1) * nb + c]
                # (a chain pass and a tree pass may differ in the last bf16 step: accept a
                # runner-up that is as good to that precision)
                assert got == int(lg.argmax()) or lg[got] >= lg.max() - 0.02 * lg.abs().max(), (b, c, depth)
                seq.append(got)


def test_draft_grown_tree_sampling_follows_the_target_distribution():
    """Draft-grown trees under sampling (T = 1): the nodes are deterministic functions of the
    draft, i.e. point-mass proposals; every emitted token is distributed as the ORACLE's
    target distribution given its predecessor (the tolerance stated for the sampling path)."""
    from tests.test_gpu_model import _check_transitions, _markov_pair, _transition_matrix
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    cfg, (ts, ds) = _markov_pair()
    P = _transition_matrix(cfg, ts, 1.0)
    Q = _transition_matrix(cfg, ds, 1.0)
    B = 4
    eng = SpecEngine(_product(cfg, ts), _product(cfg, ds),
                     EngineConfig(k=3, max_batch=B, max_seq_len=900, split_tokens=128, temperature=1.0,
                                  seed=17, draft_tree_branches=4))
    assert eng.tree and eng.sampling and eng.tree_nodes == 12
    counts = np.zeros((cfg.vocab, cfg.vocab))
    rounds = 0
    while counts.sum() < 21000:
        eng.reset()
        seqs = [[(5 * b + rounds) % cfg.vocab] for b in range(B)]
        for b in range(B):
            seqs[b].append(eng.add_request(b, seqs[b]))
        for _ in range(200):
            for b, new in enumerate(eng.step()):
                assert 1 <= len(new) <= 4 and all(0 <= t < cfg.vocab for t in new)
                seqs[b].extend(new)
        for sq in seqs:
            np.add.at(counts, (sq[:-1], sq[1:]), 1)
        rounds += 1
    m = eng.metrics()
    assert m["accepted"] > 500 and m["emitted"] - m["accepted"] > 500
    _check_transitions(counts, P)
    with pytest.raises(AssertionError):
        _check_transitions(counts, Q)


def test_draft_grown_trees_refuse_what_they_cannot_do():
    oc = MODELS["llama"]
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    with pytest.raises(ValueError):
        _engine(tw, None, 3, ngram_max=2, draft_tree_branches=3)          # no draft model
    with pytest.raises(ValueError):
        _engine(tw, tw, 3, draft_tree_branches=9)                         # more than the top-b kernel ranks
    with pytest.raises(ValueError):
        _engine(tw, tw, 8, draft_tree_branches=8)                         # more nodes than a verify pass holds
    ocfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": oc.vocab})
    ow = _product(ocfg, _bf(random_state(ocfg, seed=3)))
    with pytest.raises(ValueError):
        _engine(tw, ow, 3, draft_tree_branches=3)                         # a draft with learned positions


@pytest.mark.parametrize("seed", [11, 12])
def test_draft_grown_trees_with_sequences_joining_and_leaving(seed):
    """Slots are reused while a draft-tree engine runs (a sequence leaves, the last slot moves
    down, another prompt joins, also with a one-token prompt): every stream stays the
    autoregressive one.  The tree's topology is the same for every slot; what must follow a
    sequence to its new slot is both caches."""
    oc = MODELS["llama"]
    dc = OracleConfig(**{**vars(oc), "n_layers": 1})
    tw = _product(oc, _bf(random_state(oc, seed=21)))
    dw = _product(dc, _bf(random_state(dc, seed=23)))
    rng = np.random.default_rng(seed)
    lens = (12, 5, 1, 20, 3) if seed == 11 else tuple(int(n) for n in rng.integers(1, 40, 5))
    prompts = [[int(t) for t in rng.integers(0, oc.vocab, n)] for n in lens]
    ar = [_generate(_engine(tw, None, 0), [p], 30)[0] for p in prompts]
    eng = _engine(tw, dw, int(rng.integers(1, 5)) if seed != 11 else 3, max_batch=3,
                  draft_tree_branches=int(rng.integers(2, 5)) if seed != 11 else 3, use_graph=bool(seed % 2))
    live = {}                       # slot -> [prompt index, tokens so far]
    for i in range(3):
        live[i] = [i, [eng.add_request(i, prompts[i])]]
    waiting, done = [3, 4], {}
    while live:
        outs = eng.step()
        for slot in sorted(live):
            live[slot][1].extend(outs[slot])
        for slot in sorted(live, reverse=True):
            idx, toks = live[slot]
            if len(toks) >= 30 or (idx == 1 and len(toks) >= 9):     # (one leaves early)
                done[idx] = toks
                last = max(live)
                eng.remove_request(slot)
                if slot != last:
                    live[slot] = live[last]
                del live[last]
                if waiting:
                    j = waiting.pop(0)
                    s = eng.active
                    live[s] = [j, [eng.add_request(s, prompts[j])]]
    for idx, toks in done.items():
        n = min(len(toks), 30)
        assert toks[:n] == ar[idx][:n], idx
    assert len(done) == 5
