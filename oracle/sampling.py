"""Acceptance rules of speculative decoding, restated on the CPU (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Algorithm 1 of Leviathan et al. 2023 (arXiv 2211.17192); the same rule
is implemented for batch 1 in
/usr/local/lib/python3.10/dist-packages/transformers/generation/utils.py:3992-4059
(accept `r <= p/q` at :4027-4029, residual `clamp(p-q, 0)/sum` at :4045-4052,
bonus token when all are accepted at :4053-4054).  The reference repository
holds no implementation (/root/reference/pyproject.toml:31 names the external
dependency that does).

Conventions fixed here so that results are deterministic and portable:
  * uniforms are explicit inputs in [0, 1);
  * accept x iff  u * q(x) < p(x)  (one f32 multiply, one compare);
  * a token is drawn from non-negative weights w by inverse CDF: the first
    index t whose float64 running sum exceeds u * sum(w); if rounding leaves
    none, the last index with w > 0; if sum(w) == 0, the first argmax of p.
"""
from __future__ import annotations

import numpy as np


def inverse_cdf(weights: np.ndarray, u: float, fallback: np.ndarray | None = None) -> int:
    w = np.asarray(weights, dtype=np.float32)
    c = np.cumsum(w.astype(np.float64))
    total = c[-1]
    if not total > 0.0:
        src = w if fallback is None else np.asarray(fallback, dtype=np.float32)
        return int(np.argmax(src))
    thr = np.float64(np.float32(u)) * total
    idx = int(np.searchsorted(c, thr, side="right"))
    if idx >= w.shape[0]:
        idx = int(np.nonzero(w > 0)[0][-1])
    return idx


def rejection_sample(target_probs, bonus_ids, draft_probs, draft_ids, u_accept, u_recover,
                     proposal_lens=None):
    """Returns (out_ids [B, k+1] int64 with -1 padding, counters [accepted, emitted, draft])."""
    p = np.asarray(target_probs, dtype=np.float32)
    q = np.asarray(draft_probs, dtype=np.float32)
    x = np.asarray(draft_ids, dtype=np.int64)
    ua = np.asarray(u_accept, dtype=np.float32)
    ur = np.asarray(u_recover, dtype=np.float32)
    bonus = np.asarray(bonus_ids, dtype=np.int64).reshape(-1)
    B, k1, V = p.shape
    k = k1 - 1
    out = np.full((B, k + 1), -1, dtype=np.int64)
    counters = np.zeros(3, dtype=np.int64)
    for b in range(B):
        L = k if proposal_lens is None else int(min(max(int(proposal_lens[b]), 0), k))
        j = 0
        while j < L:
            tok = int(x[b, j])
            ok = False
            if 0 <= tok < V:
                uq = np.float32(ua[b, j]) * np.float32(q[b, j, tok])  # f32 product
                ok = bool(uq < p[b, j, tok])
            if not ok:
                break
            out[b, j] = tok
            j += 1
        if j == L and L == k:
            out[b, j] = bonus[b]
        elif j < L:
            resid = np.maximum(p[b, j] - q[b, j], np.float32(0))
            out[b, j] = inverse_cdf(resid, ur[b, j], fallback=p[b, j])
        else:  # every proposal of a short sequence accepted: draw from p_L
            out[b, j] = inverse_cdf(p[b, j], ur[b, j])
        counters += (j, j + 1, L)
    return out, counters


def greedy_verify(target_ids, draft_ids, proposal_lens=None):
    t = np.asarray(target_ids, dtype=np.int64)
    x = np.asarray(draft_ids, dtype=np.int64)
    B, k1 = t.shape
    k = k1 - 1
    out = np.full((B, k + 1), -1, dtype=np.int64)
    counters = np.zeros(3, dtype=np.int64)
    for b in range(B):
        L = k if proposal_lens is None else int(min(max(int(proposal_lens[b]), 0), k))
        j = 0
        while j < L and x[b, j] == t[b, j]:
            out[b, j] = x[b, j]
            j += 1
        out[b, j] = t[b, j]
        counters += (j, j + 1, L)
    return out, counters


def typical_acceptance(target_probs, bonus_ids, draft_ids, posterior_threshold: float = 0.09,
                       posterior_alpha: float = 0.3, proposal_lens=None, margin_out=None):
    """Typical acceptance (Cai et al. 2024, "Medusa", section 3.3.1; upstream vLLM's
    TypicalAcceptanceSampler, absent from the reference tree): proposal x_i is accepted iff
        p_i(x_i) > min(posterior_threshold, posterior_alpha * exp(-H(p_i))),
    H(p) = -sum p * log(p + 1e-5) (f32 terms, f64 sum); the token at the first rejected position
    is argmax p_j; the bonus token follows k accepted proposals.  No uniforms: deterministic.
    `margin_out`, if a list, receives |p_i(x_i) - threshold| of every tested position, so that a
    comparison with another implementation can skip cases decided by rounding."""
    p = np.asarray(target_probs, dtype=np.float32)
    x = np.asarray(draft_ids, dtype=np.int64)
    bonus = np.asarray(bonus_ids, dtype=np.int64).reshape(-1)
    B, k1, V = p.shape
    k = k1 - 1
    out = np.full((B, k + 1), -1, dtype=np.int64)
    counters = np.zeros(3, dtype=np.int64)
    for b in range(B):
        L = k if proposal_lens is None else int(min(max(int(proposal_lens[b]), 0), k))
        j = 0
        while j < L:
            tok = int(x[b, j])
            row = p[b, j]
            terms = row * np.log(row + np.float32(1e-5))
            entropy = -terms.astype(np.float64).sum()
            thr = np.float32(min(np.float64(np.float32(posterior_threshold)),
                                 np.float64(np.float32(posterior_alpha)) * np.exp(-entropy)))
            cand = row[tok] if 0 <= tok < V else np.float32(-1.0)
            if margin_out is not None:
                margin_out.append(abs(float(cand) - float(thr)))
            if not cand > thr:
                break
            out[b, j] = tok
            j += 1
        if j == L and L == k:
            out[b, j] = bonus[b]
        else:
            out[b, j] = int(np.argmax(p[b, j]))
        counters += (j, j + 1, L)
    return out, counters


def argmax_rows(logits) -> np.ndarray:
    return np.argmax(np.asarray(logits, dtype=np.float32), axis=-1).astype(np.int64)


def softmax_rows(logits, temperature: float = 1.0) -> np.ndarray:
    z = np.asarray(logits, dtype=np.float32) * np.float32(1.0 / temperature)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z.astype(np.float32))
    s = e.astype(np.float64).sum(axis=-1, keepdims=True)
    return (e * (1.0 / s).astype(np.float32)).astype(np.float32)


def sample_rows(probs, uniforms) -> np.ndarray:
    p = np.asarray(probs, dtype=np.float32)
    return np.array([inverse_cdf(p[r], uniforms[r]) for r in range(p.shape[0])], dtype=np.int64)


def filter_probs(p: np.ndarray, top_k: int = 0, top_p: float = 1.0, min_p: float = 0.0) -> np.ndarray:
    """top-k, min-p, then top-p on what is left, renormalised; one probability row (float64 in the
    arithmetic).  Follows vllm/v1/sample/ops/topk_topp_sampler.py apply_top_k_top_p (top-k keeps
    everything not below the k-th largest value; top-p drops the low tail whose cumulative mass
    is at most 1 - top_p) and the sampler's min-p rule (keep p >= min_p * max p), with the one
    stated difference that entries EQUAL to a cut-off value always stay together (upstream's
    sort puts ties in an arbitrary order).  The reference tree does not contain the sampler."""
    p = np.asarray(p, dtype=np.float64)
    V = p.size
    keep = p > 0
    if 0 < top_k < V:
        keep &= p >= np.sort(p)[::-1][top_k - 1]
    if min_p > 0:
        keep &= p >= np.float32(min_p) * np.float32(p.max())
    if top_p < 1.0:
        base = p[keep].sum()
        vals = np.unique(p[keep])[::-1]                  # distinct kept values, largest first
        acc, cut = 0.0, vals[-1]
        for v in vals:
            acc += p[keep & (p == v)].sum()
            if acc >= top_p * base:
                cut = v
                break
        keep &= p >= cut
    out = np.where(keep, p, 0.0)
    return out / out.sum()


def token_logprobs(logits: np.ndarray, tokens, n_top: int = 0):
    """Log-probability and rank of tokens[r] under raw logits row r, plus the n_top most likely
    entries of the row.  Follows upstream's sampler (vllm/v1/sample/sampler.py compute_logprobs /
    gather_logprobs: log_softmax of the raw logits in float32, top-n by value, rank = number of
    entries >= the chosen token's), the layout of ModelRunnerOutput.logprobs at
    /root/reference/src/doc/engine_core.md:326-329 (column 0 the sampled token, then the top
    max_num_logprobs).  Ties among the top entries are ordered by lower index (upstream's topk
    leaves that order unspecified).  float64 arithmetic.
    Returns (ids [R, n_top+1] int64, logprobs [R, n_top+1] float64, ranks [R] int64)."""
    x = np.asarray(logits, dtype=np.float64)
    R, V = x.shape
    mx = x.max(axis=1, keepdims=True)
    lp = x - (mx + np.log(np.exp(x - mx).sum(axis=1, keepdims=True)))
    ids = np.full((R, n_top + 1), -1, dtype=np.int64)
    out = np.full((R, n_top + 1), -np.inf)
    ranks = np.zeros(R, dtype=np.int64)
    for r in range(R):
        t = int(tokens[r])
        if not 0 <= t < V:
            continue
        ids[r, 0], out[r, 0] = t, lp[r, t]
        ranks[r] = int((x[r] >= x[r, t]).sum())
        order = np.lexsort((np.arange(V), -x[r]))[:n_top]      # value descending, index ascending
        order = order[np.isfinite(x[r, order])]
        ids[r, 1:1 + order.size] = order
        out[r, 1:1 + order.size] = lp[r, order]
    return ids, out, ranks
