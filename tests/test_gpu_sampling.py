"""HIP acceptance / sampling kernels against the CPU oracle (bit-exact ids)."""
import os

import numpy as np
import pytest
import torch

from oracle import sampling

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_library_loads_on_gpu():
    cus, name = _ops().device_info()
    assert cus > 0 and "gfx" in name


def _scratch(ops, split, rows):
    """None (one workgroup per row) or scratch for the split forms."""
    return ops.sampling_scratch(rows) if split else None


@pytest.mark.parametrize("split", [False, True])
def test_rejection_golden(split):
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "rejection_cases.npz"))
    for i in range(int(g["n_cases"])):
        pl = g[f"c{i}_proposal_lens"]
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.rejection_sample(
            cu(g[f"c{i}_target_probs"]), cu(g[f"c{i}_bonus_ids"]), cu(g[f"c{i}_draft_probs"]),
            cu(g[f"c{i}_draft_ids"]), cu(g[f"c{i}_u_accept"]), cu(g[f"c{i}_u_recover"]),
            None if pl.size == 0 else cu(pl), counters,
            scratch=_scratch(ops, split, g[f"c{i}_target_probs"].shape[0]))
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"], err_msg=f"case {i}")
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])


def test_typical_acceptance_golden():
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "typical_cases.npz"))
    for i in range(int(g["n_cases"])):
        pl = g[f"c{i}_proposal_lens"]
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.typical_acceptance(cu(g[f"c{i}_target_probs"]), cu(g[f"c{i}_bonus_ids"]),
                                     cu(g[f"c{i}_draft_ids"]), 0.09, 0.3,
                                     None if pl.size == 0 else cu(pl), counters)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"], err_msg=f"case {i}")
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])


@pytest.mark.parametrize("B,k,V,scale", [(1, 5, 128256, 4.0), (4, 3, 50272, 6.0), (8, 5, 151936, 8.0)])
def test_typical_acceptance_full_vocab(B, k, V, scale):
    """Full vocabulary rows from the library's own softmax; sequences whose outcome hangs on a
    margin below 1e-6 (rounding of the entropy sum) are left out of the comparison."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(B * 100 + k)
    tl = torch.randn(B * (k + 1), V, generator=gen, device="cuda") * scale
    p = ops.softmax_rows(tl).view(B, k + 1, V)
    top = p[:, :k].argmax(-1)
    rnd = torch.randint(0, V, (B, k), generator=gen, device="cuda")
    ids = torch.where(torch.rand(B, k, generator=gen, device="cuda") < 0.7, top, rnd)
    bonus = torch.randint(0, V, (B,), generator=gen, device="cuda")
    lens = torch.randint(0, k + 1, (B,), generator=gen, device="cuda").to(torch.int32)
    for pl in (None, lens):
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.typical_acceptance(p, bonus, ids, 0.09, 0.3, pl, counters).cpu().numpy()
        checked = 0
        for b in range(B):
            margins = []
            want, _ = sampling.typical_acceptance(
                p[b:b + 1].cpu().numpy(), bonus[b:b + 1].cpu().numpy(), ids[b:b + 1].cpu().numpy(),
                0.09, 0.3, None if pl is None else pl[b:b + 1].cpu().numpy(), margin_out=margins)
            if margins and min(margins) < 1e-6:
                continue
            np.testing.assert_array_equal(out[b], want[0], err_msg=f"sequence {b}")
            checked += 1
        assert checked >= max(1, B - 1)
        acc, emitted, draft = counters.tolist()
        assert emitted == acc + B and draft == (B * k if pl is None else int(lens.sum()))
        assert emitted == int((out >= 0).sum())


def test_typical_acceptance_rejects_bad_arguments():
    ops = _ops()
    from vllm_omni_draft_amd import _lib
    p = torch.full((1, 2, 6), 1 / 6, device="cuda")
    with pytest.raises(_lib.SdError):
        ops.typical_acceptance(p, torch.zeros(1, dtype=torch.int64, device="cuda"),
                               torch.zeros(1, 1, dtype=torch.int64, device="cuda"))   # V % 4 != 0
    p = torch.full((1, 2, 8), 1 / 8, device="cuda")
    with pytest.raises(_lib.SdError):
        ops.typical_acceptance(p, torch.zeros(1, dtype=torch.int64, device="cuda"),
                               torch.zeros(1, 3, dtype=torch.int64, device="cuda"))   # k mismatch


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("B,k,V", [(1, 5, 128256), (4, 3, 50272), (32, 5, 151936), (3, 1, 64)])
def test_rejection_random_full_vocab(B, k, V, split):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(B * 1000 + k)
    tl = torch.randn(B * (k + 1), V, generator=gen, device="cuda") * 3
    dl = tl.view(B, k + 1, V)[:, :k].reshape(B * k, V) + torch.randn(B * k, V, generator=gen, device="cuda")
    sc = _scratch(ops, split, B * (k + 1))
    p = ops.softmax_rows(tl, scratch=sc).view(B, k + 1, V)
    q = ops.softmax_rows(dl.contiguous(), scratch=sc).view(B, k, V)
    ud = torch.rand(B * k, generator=gen, device="cuda")
    ids = ops.sample_rows(q.view(B * k, V), ud, scratch=sc).view(B, k)
    ref_ids = sampling.sample_rows(q.view(B * k, V).cpu().numpy(), ud.cpu().numpy())
    np.testing.assert_array_equal(ids.view(-1).cpu().numpy(), ref_ids)
    ua = torch.rand(B, k, generator=gen, device="cuda")
    ur = torch.rand(B, k + 1, generator=gen, device="cuda")
    bonus = torch.randint(0, V, (B,), generator=gen, device="cuda")
    lens = torch.randint(0, k + 1, (B,), generator=gen, device="cuda").to(torch.int32)
    for pl in (None, lens):
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.rejection_sample(p, bonus, q, ids, ua, ur, pl, counters, scratch=sc)
        ref, cnt = sampling.rejection_sample(p.cpu().numpy(), bonus.cpu().numpy(), q.cpu().numpy(),
                                             ids.cpu().numpy(), ua.cpu().numpy(), ur.cpu().numpy(),
                                             None if pl is None else pl.cpu().numpy())
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        np.testing.assert_array_equal(counters.cpu().numpy(), cnt)


def test_greedy_golden_and_argmax():
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "greedy_cases.npz"))
    for i in range(int(g["n_cases"])):
        logits = g[f"c{i}_logits"]
        B, k1, V = logits.shape
        t_ids = ops.argmax_rows(cu(logits).view(B * k1, V)).view(B, k1)
        np.testing.assert_array_equal(t_ids.cpu().numpy(), g[f"c{i}_target_ids"])
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.greedy_verify(t_ids, cu(g[f"c{i}_draft_ids"]), None, counters)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"])
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])


def test_greedy_ragged_and_ties():
    ops = _ops()
    t = torch.tensor([[5, 6, 7, 8], [5, 6, 7, 8], [1, 2, 3, 4]], device="cuda")
    d = torch.tensor([[5, 6, 7], [5, 9, 7], [1, 2, 3]], device="cuda")
    lens = torch.tensor([1, 3, 0], dtype=torch.int32, device="cuda")
    out = ops.greedy_verify(t, d, lens)
    ref, _ = sampling.greedy_verify(t.cpu().numpy(), d.cpu().numpy(), lens.cpu().numpy())
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # argmax ties resolve to the smallest index, also across threads / waves
    x = torch.zeros(2, 128256, device="cuda")
    x[0, [70000, 123, 99999]] = 3.0
    x[1, 128255] = 1.0
    assert ops.argmax_rows(x).tolist() == [123, 128255]


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("V", [64, 50272, 128256])
def test_softmax_and_sample_rows(V, split):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(V)
    x = torch.randn(6, V, generator=gen, device="cuda") * 4
    sc = _scratch(ops, split, 6)
    for T in (1.0, 0.7):
        p = ops.softmax_rows(x, T, scratch=sc)
        ref = sampling.softmax_rows(x.cpu().numpy(), T)
        np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=2e-5, atol=1e-9)
    u = torch.rand(6, generator=gen, device="cuda")
    ids = ops.sample_rows(p, u, scratch=sc)
    np.testing.assert_array_equal(ids.cpu().numpy(), sampling.sample_rows(p.cpu().numpy(), u.cpu().numpy()))
    # strided rows (every other row) and the extreme uniforms
    ids2 = ops.sample_rows(p, torch.tensor([0.0, 0.99999994, 0.5], device="cuda"), row_offset=1, row_step=2,
                           scratch=sc)
    ref2 = sampling.sample_rows(p.cpu().numpy()[1::2], [0.0, 0.99999994, 0.5])
    np.testing.assert_array_equal(ids2.cpu().numpy(), ref2)


def test_bad_arguments_raise():
    from vllm_omni_draft_amd._lib import SdError
    ops = _ops()
    p = torch.full((1, 2, 6), 1 / 6, device="cuda")          # V % 4 != 0
    with pytest.raises(SdError):
        ops.rejection_sample(p, torch.zeros(1, dtype=torch.int64, device="cuda"), p[:, :1].contiguous(),
                             torch.zeros(1, 1, dtype=torch.int64, device="cuda"),
                             torch.zeros(1, 1, device="cuda"), torch.zeros(1, 2, device="cuda"))
    with pytest.raises(SdError):                              # CPU tensors are refused: no CPU path
        ops.argmax_rows(torch.zeros(2, 8))


def _partials(ops, logits):
    R = logits.shape[0]
    pv = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.float32, device="cuda")
    pi = torch.empty(R, ops.ARGMAX_PARTS, dtype=torch.int32, device="cuda")
    ops.argmax_partial(logits, pv, pi)
    return pv, pi


def test_fused_greedy_glue_matches_golden_and_oracle():
    """argmax partials -> proposal store / verification + state update, against the
    golden greedy cases and the oracle's greedy_verify, with the step record in pinned
    host memory."""
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "greedy_cases.npz"))
    for i in range(int(g["n_cases"])):
        logits = g[f"c{i}_logits"]
        B, k1, V = logits.shape
        k = k1 - 1
        draft = g[f"c{i}_draft_ids"]
        pv, pi = _partials(ops, cu(logits).view(B * k1, V))
        tok_buf = torch.zeros(B, k + 2, dtype=torch.int32, device="cuda")
        tok_buf[:, 0] = 11
        tok_buf[:, 1] = 12
        tok_buf[:, 2:] = cu(draft).to(torch.int32)
        seq_lens = torch.full((B,), 40, dtype=torch.int32, device="cuda")
        gen = torch.zeros(B, 32, dtype=torch.int32, device="cuda")
        cnt = torch.full((B,), 3, dtype=torch.int32, device="cuda")
        out = torch.zeros(B, k1, dtype=torch.int64, device="cuda")
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        rec = torch.full((B, k + 3), -7, dtype=torch.int32).pin_memory()
        ops.verify_from_partials(pv, pi, tok_buf, out, counters, seq_lens, gen, cnt, k, rec)
        torch.cuda.synchronize()
        ref = g[f"c{i}_out_ids"]
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])
        n_emit = (ref >= 0).sum(axis=1)
        np.testing.assert_array_equal(rec[:, 0].numpy(), n_emit)
        np.testing.assert_array_equal(rec[:, 1].numpy(), 40 + n_emit)
        np.testing.assert_array_equal(rec[:, 2:].numpy(), ref)
        np.testing.assert_array_equal(seq_lens.cpu().numpy(), 40 + n_emit)
        np.testing.assert_array_equal(cnt.cpu().numpy(), 3 + n_emit)
        for b in range(B):
            toks = ref[b, :n_emit[b]]
            np.testing.assert_array_equal(gen[b, 3:3 + n_emit[b]].cpu().numpy(), toks)
            prev = toks[-2] if n_emit[b] >= 2 else 12
            assert tok_buf[b, :2].tolist() == [prev, toks[-1]]


@pytest.mark.parametrize("V", [64, 50272, 128256, 151936])
def test_argmax_partials_ties_and_strided_rows(V):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(V)
    x = torch.randn(6, V, generator=gen, device="cuda")
    x[0, [V - 1, 3, V // 2]] = 9.0          # ties across slices: smallest index wins
    x[1, V - 1] = 9.0                       # maximum in the last (short) slice
    x[2] = -float("inf")                    # no finite entry: index 0
    pv, pi = _partials(ops, x)
    ref = sampling.argmax_rows(x.cpu().numpy())
    # draft side: rows 1, 3, 5 (the q_len = 2 first pass reads every other row)
    tok_buf = torch.zeros(3, 7, dtype=torch.int32, device="cuda")
    ops.propose_from_partials(pv, pi, 1, 2, tok_buf, 5, 4)
    assert tok_buf[:, 6].tolist() == [int(ref[1]), int(ref[3]), int(ref[5])]
    assert tok_buf[:, :6].abs().sum().item() == 0
    # k = 0 verification is plain decoding: every row emits its argmax
    seq_lens = torch.full((6,), 9, dtype=torch.int32, device="cuda")
    tb = torch.zeros(6, 2, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(6, dtype=torch.int32, device="cuda")
    rec = torch.zeros(6, 3, dtype=torch.int32).pin_memory()
    ops.verify_from_partials(pv, pi, tb, None, None, seq_lens, None, cnt, 0, rec)
    torch.cuda.synchronize()
    assert rec[:, 2].tolist() == [int(v) for v in ref]
    assert rec[:, 0].tolist() == [1] * 6 and seq_lens.tolist() == [10] * 6
    assert tb[:, 1].tolist() == [int(v) for v in ref]


def test_store_proposal_is_the_same_in_both_paths():
    ops = _ops()
    B, k, V = 8, 5, 4096
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, V, generator=gen, device="cuda")
    pv, pi = _partials(ops, x)
    for step in range(k):
        a = torch.zeros(B, k + 2, dtype=torch.int32, device="cuda")
        b = torch.zeros(B, k + 2, dtype=torch.int32, device="cuda")
        ops.store_proposal(ops.argmax_rows(x), a, k, step)
        ops.propose_from_partials(pv, pi, 0, 1, b, k, step)
        assert torch.equal(a, b)
        assert torch.equal(a[:, 2 + step].long(), x.argmax(dim=1))


def test_debug_force_proposals_follows_its_schedule():
    """The benchmark control: proposals 0..a-1 become the true tokens, proposal a a token that
    is not the true one, later ones are left alone; a walks cyclically through the pattern."""
    ops = _ops()
    B, k, V = 4, 5, 1000
    gen = torch.Generator(device="cuda").manual_seed(6)
    truth = torch.randint(0, V, (B, 64), generator=gen, device="cuda").to(torch.int32)
    seq_lens = torch.tensor([10, 11, 12, 13], dtype=torch.int32, device="cuda")
    ctr = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device="cuda")   # sequences out of phase
    pattern = [5, 0, 3, 1, 4]
    for rnd in range(7):
        tb = torch.full((B, k + 2), -7, dtype=torch.int32, device="cuda")
        ops.debug_force_proposals(tb, k, seq_lens, truth, pattern, ctr, V)
        torch.cuda.synchronize()
        for b in range(B):
            a = pattern[(b + rnd) % len(pattern)]
            n = int(seq_lens[b])
            for i in range(k):
                want = int(truth[b, n + i])
                got = int(tb[b, 2 + i])
                if i < a:
                    assert got == want
                elif i == a:
                    assert got == (want + 1) % V
                else:
                    assert got == -7
        assert tb[:, :2].eq(-7).all()
    assert ctr.tolist() == [7, 8, 9, 10]


def test_fill_uniforms_is_uniform_and_advances():
    ops = _ops()
    u = torch.zeros(20000, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.fill_uniforms(u, 7, counter)
    a = u.clone()
    ops.fill_uniforms(u, 7, counter)
    assert counter.item() == 2 and not torch.equal(a, u)
    for t in (a, u):
        assert 0.0 <= t.min().item() and t.max().item() < 1.0
        assert abs(t.mean().item() - 0.5) < 0.01
        hist = torch.histc(t, bins=10, min=0, max=1)
        assert (hist - 2000).abs().max().item() < 250
    counter.zero_()
    ops.fill_uniforms(u, 7, counter)          # same (seed, counter) -> same numbers
    assert torch.equal(a, u)


def test_per_sequence_uniform_streams_match_the_host_hash():
    """sd_fill_uniforms_seq: sequence b's j-th number of a step is hash(seed_b, counter_b, j) at
    the place its region says; counters advance by one; seeds above 2**63 work."""
    from vllm_omni_draft_amd import ops
    B, k = 5, 3
    sizes = [k * B, B * k, B * (k + 1), B]
    out = torch.full((sum(sizes) + 7,), -1.0, dtype=torch.float32).cuda()
    regions = [(0, k, B, 1), (k * B, k, 1, k), (2 * k * B, k + 1, 1, k + 1), (2 * k * B + B * (k + 1), 1, 0, 1)]
    seeds_u = [0, 1, 12345678901234567890, 2 ** 64 - 1, 42]
    seeds = torch.tensor([s - 2 ** 64 if s >= 2 ** 63 else s for s in seeds_u], dtype=torch.int64).cuda()
    counters = torch.tensor([0, 5, 7, 2 ** 40, 3], dtype=torch.int64).cuda()
    c0 = counters.tolist()
    n_active = 4                                   # the last sequence is not in the batch
    for step in range(2):
        ops.fill_uniforms_seq(out, ops.uniform_regions(regions), seeds, counters, n_active)
        torch.cuda.synchronize()
        got = out.cpu().tolist()
        for b in range(n_active):
            j = 0
            for off, n, sj, sb in regions:
                for i in range(n):
                    want = np.float32(ops.uniform24(seeds_u[b], c0[b] + step, j))
                    assert np.float32(got[off + i * sj + b * sb]) == want, (b, j)
                    j += 1
        assert counters.tolist() == [c + step + 1 for c in c0[:n_active]] + c0[n_active:]
    assert all(v == -1.0 for v in got[sum(sizes):])                # nothing past the regions
    vals = [got[off + i * sj + b * sb] for off, n, sj, sb in regions for i in range(n) for b in range(n_active)]
    assert all(0.0 <= v < 1.0 for v in vals) and len(set(vals)) > len(vals) - 3


@pytest.mark.parametrize("split", [False, True])
def test_softmax_with_one_temperature_per_sequence(split):
    """Row r is softened by temperatures[r // rows_per_seq]; a temperature of 0 puts all mass on
    the row's first maximum (ties included), in both launch forms."""
    from vllm_omni_draft_amd import ops
    g = torch.Generator().manual_seed(3)
    n_seq, per, V = 5, 3, 50272
    logits = torch.randn(n_seq * per, V, generator=g) * 3.0
    logits[4, 100] = logits[4, 40000] = logits[4].max() + 1.0        # a tie inside a greedy sequence
    logits[5, V - 1] = logits[5].max() + 2.0                          # maximum in the last slice
    temps = torch.tensor([0.7, 0.0, 1.0, 2.5, 0.0])
    scratch = ops.sampling_scratch(n_seq * per, "cuda") if split else None
    out = ops.softmax_rows(logits.cuda(), temps.cuda(), rows_per_seq=per, scratch=scratch).cpu()
    for r in range(n_seq * per):
        T = temps[r // per].item()
        if T > 0:
            ref = torch.softmax(logits[r].double() / T, dim=0).float()
            torch.testing.assert_close(out[r], ref, rtol=2e-5, atol=1e-9)
        else:
            want = torch.zeros(V)
            want[int(logits[r].argmax())] = 1.0                       # torch returns the first maximum
            assert torch.equal(out[r], want), r
    assert out[4, 100] == 1.0 and out[4, 40000] == 0.0 and out[5, V - 1] == 1.0
    # strided rows (a draft pass's "last row of every sequence") with one row per sequence
    sub = ops.softmax_rows(logits.cuda(), temps.cuda(), row_offset=2, row_step=3, scratch=scratch).cpu()
    for b in range(n_seq):
        assert torch.equal(sub[b], out[3 * b + 2]) or temps[b] != temps[(3 * b + 2) // per]


@pytest.mark.parametrize("V", [1000, 50272, 128256, 151936])
def test_top_k_top_p_min_p_filter_matches_oracle(V):
    """sd_filter_probs_seq against the oracle's restatement of upstream's top-k / min-p / top-p, per
    sequence parameters, three row-storage tiers (registers, LDS, global: the largest vocabulary),
    ties at the cut-off, filters off, and a greedy (one-hot) row."""
    from vllm_omni_draft_amd import ops
    rng = np.random.default_rng(V)
    params = [dict(top_k=50), dict(top_p=0.9), dict(min_p=0.05), dict(top_k=20, top_p=0.8),
              dict(top_k=1000, top_p=0.95, min_p=0.01), dict(), dict(top_k=1), dict(top_p=0.3)]
    per = 2
    rows = []
    for i in range(len(params) * per):
        logits = rng.normal(size=V) * (2.0 + i % 3)
        p = np.exp(logits - logits.max())
        rows.append((p / p.sum()).astype(np.float32))
    rows[0][5] = rows[0][7] = np.sort(rows[0])[::-1][49]          # a tie exactly at the top-k cut-off
    rows[-1][:] = 0.0
    rows[-1][V // 3] = 1.0                                         # a greedy request's row
    probs = torch.tensor(np.stack(rows)).cuda()
    tk = torch.tensor([q.get("top_k", 0) for q in params], dtype=torch.int32).cuda()
    tp = torch.tensor([q.get("top_p", 1.0) for q in params], dtype=torch.float32).cuda()
    mp = torch.tensor([q.get("min_p", 0.0) for q in params], dtype=torch.float32).cuda()
    ops.filter_probs(probs, tk, tp, mp, rows_per_seq=per)
    torch.cuda.synchronize()
    got = probs.cpu().numpy()
    for r, row in enumerate(rows):
        q = params[r // per]
        if not q:
            assert np.array_equal(got[r], row)                     # untouched
            continue
        # f32 mass sums on the device: the nucleus may end one value earlier or later when the
        # cumulative mass lands within rounding of the bound
        cands = [sampling.filter_probs(row, q.get("top_k", 0), q.get("top_p", 1.0) + d, q.get("min_p", 0.0))
                 for d in ((0.0, -2e-6, 2e-6) if "top_p" in q else (0.0,))]
        assert any(np.array_equal(got[r] > 0, c > 0) and np.allclose(got[r], c, rtol=2e-5, atol=1e-9)
                   for c in cands), (r, q, (got[r] > 0).sum(), [(c > 0).sum() for c in cands])
        assert abs(got[r].sum() - 1.0) < 1e-5


@pytest.mark.parametrize("V", [1000, 50272, 128256, 151936])
def test_token_logprobs_match_oracle(V):
    """sd_token_logprobs against the oracle (float64 log-softmax): chosen token's log-probability
    and rank, the top entries in (value, index) order, with exact ties in the top, a skipped row,
    the most and the least likely token chosen, and a row with -inf entries."""
    ops, DEV = _ops(), "cuda"
    g = torch.Generator().manual_seed(V)
    R, n_top = 7, 5
    x = (torch.randn(R, V, generator=g) * 3.0).float()
    x[1, 17] = x[1, 900] = x[1].max() + 1.0          # a tie at the top: lower index first
    x[2, 5] = x[2, 6] = x[2, 7] = x[2].max() + 0.5   # three equal
    x[5, 10:] = float("-inf")                        # ten finite entries
    x[6, 3:] = float("-inf")                         # fewer finite entries than n_top
    toks = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    toks[0] = int(x[0].argmax())
    toks[1] = 900
    toks[3] = -1                                     # skipped
    toks[4] = int(x[4].argmin())
    toks[5] = 4
    toks[6] = 1
    ids, lps, ranks = ops.token_logprobs(x.to(DEV), toks.to(DEV), n_top)
    ids, lps, ranks = ids.cpu().numpy(), lps.cpu().numpy().astype(np.float64), ranks.cpu().numpy()
    w_ids, w_lps, w_ranks = sampling.token_logprobs(x.numpy(), toks.numpy(), n_top)
    w_lps[3], w_ids[3] = -np.inf, -1                 # the skipped row keeps the wrapper's fill
    np.testing.assert_array_equal(ids, w_ids)
    np.testing.assert_array_equal(ranks, w_ranks)
    finite = np.isfinite(w_lps)
    assert (np.isfinite(lps) == finite).all()
    # tolerance: float32 logsumexp (f64 sum of f32 exponentials) against the float64 oracle
    np.testing.assert_allclose(lps[finite], w_lps[finite], rtol=0, atol=2e-5)
    assert ranks[0] == 1 and ranks[1] == 2 and ranks[4] == V


def test_token_logprobs_without_top_entries():
    ops, DEV = _ops(), "cuda"
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 4096, generator=g).float()
    toks = torch.tensor([0, 4095, 77, 1234], dtype=torch.int32)
    ids, lps, ranks = ops.token_logprobs(x.to(DEV), toks.to(DEV), 0)
    want = torch.log_softmax(x, dim=-1)[torch.arange(4), toks.long()]
    assert ids.shape == (4, 1) and ids[:, 0].cpu().tolist() == toks.tolist()
    torch.testing.assert_close(lps[:, 0].cpu(), want, rtol=0, atol=2e-5)
