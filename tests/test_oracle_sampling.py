"""CPU checks of the acceptance oracle: golden vectors, the independent
transformers implementation, distribution preservation, edge cases."""
import os

import numpy as np
import pytest
import torch

from oracle import sampling

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _rand_probs(rng, *shape):
    x = rng.standard_normal(shape).astype(np.float32) * 2.0
    return sampling.softmax_rows(x)


def test_golden_rejection_cases():
    g = np.load(os.path.join(GOLDEN, "rejection_cases.npz"))
    n = int(g["n_cases"])
    assert n >= 8
    for i in range(n):
        pl = g[f"c{i}_proposal_lens"]
        out, cnt = sampling.rejection_sample(
            g[f"c{i}_target_probs"], g[f"c{i}_bonus_ids"], g[f"c{i}_draft_probs"],
            g[f"c{i}_draft_ids"], g[f"c{i}_u_accept"], g[f"c{i}_u_recover"],
            None if pl.size == 0 else pl)
        np.testing.assert_array_equal(out, g[f"c{i}_out_ids"])
        np.testing.assert_array_equal(cnt, g[f"c{i}_counters"])


def test_golden_greedy_cases():
    g = np.load(os.path.join(GOLDEN, "greedy_cases.npz"))
    for i in range(int(g["n_cases"])):
        t_ids = sampling.argmax_rows(g[f"c{i}_logits"])
        np.testing.assert_array_equal(t_ids, g[f"c{i}_target_ids"])
        out, cnt = sampling.greedy_verify(t_ids, g[f"c{i}_draft_ids"])
        np.testing.assert_array_equal(out, g[f"c{i}_out_ids"])
        np.testing.assert_array_equal(cnt, g[f"c{i}_counters"])


def test_golden_typical_cases():
    g = np.load(os.path.join(GOLDEN, "typical_cases.npz"))
    n = int(g["n_cases"])
    assert n >= 8
    for i in range(n):
        pl = g[f"c{i}_proposal_lens"]
        out, cnt = sampling.typical_acceptance(g[f"c{i}_target_probs"], g[f"c{i}_bonus_ids"],
                                               g[f"c{i}_draft_ids"], 0.09, 0.3,
                                               None if pl.size == 0 else pl)
        np.testing.assert_array_equal(out, g[f"c{i}_out_ids"])
        np.testing.assert_array_equal(cnt, g[f"c{i}_counters"])


def test_typical_acceptance_known_answers():
    """Hand-computed thresholds.  One-hot rows: H = 0, threshold = min(cap, alpha) -> only the
    hot token stands.  Uniform rows over V = 64: H = -log(1/64 + 1e-5) ~ 4.158, threshold
    0.3 * exp(-H) ~ 0.00469 < 1/64 -> every token stands."""
    V, k = 64, 3
    hot = np.zeros((1, k + 1, V), np.float32)
    for i, t in enumerate([7, 8, 9, 10]):
        hot[0, i, t] = 1.0
    out, cnt = sampling.typical_acceptance(hot, [33], [[7, 8, 9]])
    assert out.tolist() == [[7, 8, 9, 33]] and cnt.tolist() == [3, 4, 3]      # all stand + bonus
    out, cnt = sampling.typical_acceptance(hot, [33], [[7, 5, 9]])
    assert out.tolist() == [[7, 8, -1, -1]] and cnt.tolist() == [1, 2, 3]     # argmax replaces x_1
    out, cnt = sampling.typical_acceptance(hot, [33], [[7, 8, 9]], proposal_lens=[2])
    assert out.tolist() == [[7, 8, 9, -1]] and cnt.tolist() == [2, 3, 2]      # short: argmax p_L
    flat = np.full((1, k + 1, V), 1.0 / V, np.float32)
    out, cnt = sampling.typical_acceptance(flat, [5], [[1, 2, 3]])
    assert out.tolist() == [[1, 2, 3, 5]] and cnt.tolist() == [3, 4, 3]
    # the cap rules when alpha * exp(-H) is above it: 0.05 stands under cap 0.04, falls under 0.09
    row = np.zeros(V, np.float32)
    row[0], row[1] = 0.95, 0.05
    two = np.tile(row, (1, 2, 1))
    assert sampling.typical_acceptance(two, [9], [[1]], 0.04, 0.3)[0].tolist() == [[1, 9]]
    assert sampling.typical_acceptance(two, [9], [[1]], 0.09, 0.3)[0].tolist() == [[0, -1]]
    # out-of-range proposal ids never stand
    assert sampling.typical_acceptance(two, [9], [[V + 3]])[0].tolist() == [[0, -1]]


def test_greedy_known_answers():
    t = np.array([[5, 6, 7, 8]])
    out, cnt = sampling.greedy_verify(t, np.array([[5, 6, 7]]))
    assert out.tolist() == [[5, 6, 7, 8]] and cnt.tolist() == [3, 4, 3]      # all accepted + bonus
    out, cnt = sampling.greedy_verify(t, np.array([[9, 6, 7]]))
    assert out.tolist() == [[5, -1, -1, -1]] and cnt.tolist() == [0, 1, 3]   # first rejected
    out, cnt = sampling.greedy_verify(t, np.array([[5, 9, 7]]))
    assert out.tolist() == [[5, 6, -1, -1]] and cnt.tolist() == [1, 2, 3]    # mid rejection
    out, cnt = sampling.greedy_verify(t, np.array([[5, 6, 7]]), proposal_lens=[1])
    assert out.tolist() == [[5, 6, -1, -1]] and cnt.tolist() == [1, 2, 1]    # ragged length
    out, cnt = sampling.greedy_verify(t, np.array([[5, 6, 7]]), proposal_lens=[0])
    assert out.tolist() == [[5, -1, -1, -1]] and cnt.tolist() == [0, 1, 0]


def test_rejection_edge_cases():
    V, k = 8, 2
    p = np.zeros((1, k + 1, V), np.float32)
    q = np.zeros((1, k, V), np.float32)
    p[0, :, 3] = 1.0
    q[0, :, 3] = 1.0
    ua = np.full((1, k), 0.999, np.float32)
    ur = np.full((1, k + 1), 0.5, np.float32)
    # p == q: always accepted, bonus emitted
    out, cnt = sampling.rejection_sample(p, [7], q, [[3, 3]], ua, ur)
    assert out.tolist() == [[3, 3, 7]] and cnt.tolist() == [2, 3, 2]
    # q(x) == 0 < p(x): accepted (ratio is +inf)
    q2 = q.copy(); q2[0, 0] = 0; q2[0, 0, 1] = 1.0
    out, _ = sampling.rejection_sample(p, [7], q2, [[3, 3]], ua, ur)
    assert out.tolist() == [[3, 3, 7]]
    # p(x) == 0: always rejected, even with u == 0; recovered token comes from max(p-q, 0)
    out, cnt = sampling.rejection_sample(p, [7], q2, [[1, 3]], np.zeros((1, k), np.float32), ur)
    assert out.tolist() == [[3, -1, -1]] and cnt.tolist() == [0, 1, 2]
    # out-of-vocabulary proposal is rejected
    out, _ = sampling.rejection_sample(p, [7], q, [[99, 3]], ua, ur)
    assert out.tolist() == [[3, -1, -1]]
    # k == 1
    out, _ = sampling.rejection_sample(p[:, :2], [6], q[:, :1], [[3]], ua[:, :1], ur[:, :2])
    assert out.tolist() == [[3, 6]]


def test_inverse_cdf_properties():
    w = np.array([0, 0.25, 0, 0.5, 0.25, 0], np.float32)
    assert sampling.inverse_cdf(w, 0.0) == 1
    assert sampling.inverse_cdf(w, 0.2499) == 1
    assert sampling.inverse_cdf(w, 0.25) == 3          # running sum must EXCEED u * total
    assert sampling.inverse_cdf(w, 0.7499) == 3
    assert sampling.inverse_cdf(w, 0.75) == 4
    assert sampling.inverse_cdf(w, np.nextafter(np.float32(1), np.float32(0))) == 4
    assert sampling.inverse_cdf(np.zeros(4, np.float32), 0.3, fallback=np.array([0, 2, 2, 1])) == 1


def test_matches_transformers_speculative_sampling(monkeypatch):
    """Batch-1 cross-check against transformers' `_speculative_sampling`
    (generation/utils.py:3992) with its RNG replaced by explicit uniforms."""
    from transformers.generation import utils as gu

    rng = np.random.default_rng(11)
    V, k = 48, 4
    n_checked = {"reject": 0, "all": 0}
    for case in range(40):
        t_logits = rng.standard_normal((1, k + 1, V)).astype(np.float32) * 2
        # every few cases make the draft close to the target so that all k are accepted
        d_logits = t_logits[:, :k] + rng.standard_normal((1, k, V)).astype(np.float32) * \
            (0.05 if case % 4 == 0 else 1.5)
        p = sampling.softmax_rows(t_logits[0])[None]
        q = sampling.softmax_rows(d_logits[0])[None]
        ids = np.array([[sampling.inverse_cdf(q[0, i], rng.random()) for i in range(k)]])
        ua = rng.random((1, k), dtype=np.float32)
        ur = rng.random((1, k + 1), dtype=np.float32)
        bonus = sampling.inverse_cdf(p[0, k], ur[0, k])

        seen = {}

        def fake_rand_like(x, *a, **kw):
            return torch.from_numpy(ua[0].copy()).to(x.dtype)

        def fake_multinomial(probs, num_samples=1, **kw):
            seen["p_prime"] = probs.detach().clone()
            n = seen["n"]
            tok = sampling.inverse_cdf(probs[0].numpy(), ur[0, n])
            return torch.tensor([[tok]])

        monkeypatch.setattr(torch, "rand_like", fake_rand_like)
        monkeypatch.setattr(torch, "multinomial", fake_multinomial)
        # number of matches is needed by the fake multinomial to pick its uniform
        out, cnt = sampling.rejection_sample(p, [bonus], q, ids, ua, ur)
        seen["n"] = int(cnt[0])
        cand = torch.cat([torch.zeros(1, 3, dtype=torch.long), torch.from_numpy(ids)], dim=1)
        valid, n_matches = gu._speculative_sampling(
            cand, torch.from_numpy(d_logits), k, torch.from_numpy(t_logits), False)
        monkeypatch.undo()
        # transformers accepts on r <= ratio, the oracle on u*q < p: equal away from ties
        ratio = p[0, np.arange(k), ids[0]] / q[0, np.arange(k), ids[0]]
        if np.any(np.abs(ua[0] - ratio) < 1e-5):
            continue
        assert int(n_matches) == int(cnt[0])
        mine = [int(t) for t in out[0] if t >= 0]
        assert valid[0].tolist() == mine
        if cnt[0] < k:
            resid = np.maximum(p[0, cnt[0]] - q[0, cnt[0]], 0)
            np.testing.assert_allclose(seen["p_prime"][0].numpy(), resid / resid.sum(), rtol=1e-4,
                                       atol=1e-6)
            n_checked["reject"] += 1
        else:
            n_checked["all"] += 1
    assert n_checked["reject"] >= 10 and n_checked["all"] >= 3


@pytest.mark.slow
def test_output_distribution_equals_target():
    """Lossless property: the first emitted token is distributed as p_0 whatever q is."""
    rng = np.random.default_rng(5)
    V, k, N = 12, 3, 60000
    p = _rand_probs(rng, 1, k + 1, V)
    q = _rand_probs(rng, 1, k, V)
    counts = np.zeros(V)
    P = np.repeat(p, N, axis=0)
    Q = np.repeat(q, N, axis=0)
    cdf = np.cumsum(q[0].astype(np.float64), axis=-1)
    ids = np.stack([np.minimum((rng.random(N)[:, None] * cdf[i, -1] >= cdf[i][None]).sum(1), V - 1)
                    for i in range(k)], axis=1)
    ua = rng.random((N, k), dtype=np.float32)
    ur = rng.random((N, k + 1), dtype=np.float32)
    bonus = np.zeros(N, np.int64)
    out, cnt = sampling.rejection_sample(P, bonus, Q, ids, ua, ur)
    first = out[:, 0]
    counts = np.bincount(first, minlength=V).astype(np.float64)
    expected = p[0, 0].astype(np.float64) * N
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 45.0, chi2     # 11 dof: P(chi2 > 45) < 1e-5
    # expected acceptance of position 0 is sum(min(p, q))
    # a rejected token has p < q, hence zero residual mass: out[:, 0] == x_0 iff accepted;
    # the acceptance probability of position 0 is sum(min(p_0, q_0))
    rate = (out[:, 0] == ids[:, 0]).mean()
    assert abs(rate - np.minimum(p[0, 0], q[0, 0]).sum()) < 0.01
    assert cnt[1] == cnt[0] + N and cnt[2] == N * k
