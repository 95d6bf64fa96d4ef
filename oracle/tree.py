"""Token trees (multi-candidate verification) restated on the CPU.  TEST INFRASTRUCTURE ONLY.

The reference repository contains no tree verification (nor any other speculative-decoding
code); the semantics restated here are those of tree attention as published for SpecInfer
(Miao et al., 2023, section 4) and Medusa (Cai et al., 2024, section 2.1.2): candidate
continuations are merged into a prefix tree, every tree node attends the context and its own
ancestors, and under greedy decoding the accepted path is the longest root path whose tokens
are the target model's own arg-max continuation.  Parity unpinned: no fixture of the reference
covers it; the invariant the tests pin instead is that the emitted stream is the target's
autoregressive stream and that every row's logits equal those of the same tokens scored as a
chain.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def build_tree(candidates: Sequence[Sequence[int]], k: int, max_nodes: int
               ) -> Tuple[List[int], List[int], List[int]]:
    """Prefix tree of the candidate chains (each cut to k tokens), nodes numbered from 1 in
    order of first appearance; row 0 is the root.  Returns (tokens, parents, depths) with
    entry 0 belonging to the root (token 0 stands in for it)."""
    tokens, parents, depths = [0], [0], [0]
    for chain in candidates:
        cur = 0
        for d, tok in enumerate(list(chain)[:k]):
            found = None
            for row in range(1, len(tokens)):
                if parents[row] == cur and tokens[row] == tok:
                    found = row
                    break
            if found is None:
                if len(tokens) - 1 >= max_nodes:
                    break
                tokens.append(int(tok))
                parents.append(cur)
                depths.append(d + 1)
                found = len(tokens) - 1
            cur = found
    return tokens, parents, depths


def ancestor_masks(parents: Sequence[int]) -> List[int]:
    """Bit j of entry t: row t attends row j (itself and every ancestor up to the root)."""
    masks = []
    for t in range(len(parents)):
        m, cur = 1 << t, t
        while cur != 0:
            cur = parents[cur]
            m |= 1 << cur
        masks.append(m)
    return masks


def accept_greedy(target_ids: Sequence[int], tokens: Sequence[int], parents: Sequence[int],
                  k: int) -> Tuple[List[int], List[int]]:
    """target_ids[t]: the target's arg-max token after row t.  Returns (accepted rows, emitted
    tokens): descend from the root while a child carries the target's token (lowest row first)
    and fewer than k rows are accepted; the last emitted token is the target's own."""
    cur, path, out = 0, [], []
    while len(path) < k:
        want = int(target_ids[cur])
        nxt = next((r for r in range(1, len(tokens)) if parents[r] == cur and tokens[r] == want), None)
        if nxt is None:
            break
        path.append(nxt)
        out.append(want)
        cur = nxt
    out.append(int(target_ids[cur]))
    return path, out


def tree_attention(q: torch.Tensor, k_all: torch.Tensor, v_all: torch.Tensor, n_ctx: int,
                   masks: Sequence[int], scale: float) -> torch.Tensor:
    """q [T, Hq, D]: the tree's rows; k_all, v_all [n_ctx + T, Hkv, D]: the context followed
    by the rows' own keys/values in row order.  Row t sees keys 0..n_ctx-1 and the new key j
    iff bit j of masks[t].  Returns [T, Hq, D] float32."""
    q, k, v = q.float(), k_all.float(), v_all.float()
    T, Hq, D = q.shape
    G = Hq // k.shape[1]
    out = torch.zeros(T, Hq, D)
    for t in range(T):
        keep = list(range(n_ctx)) + [n_ctx + j for j in range(T) if (masks[t] >> j) & 1]
        for h in range(Hq):
            s = (k[keep, h // G] @ q[t, h]) * scale
            out[t, h] = torch.softmax(s, dim=0) @ v[keep, h // G]
    return out


def lookup_candidates(history: Sequence[int], k: int, ngram_max: int, ngram_min: int,
                      n_candidates: int) -> List[List[int]]:
    """Prompt-lookup candidates: for n = ngram_max .. ngram_min, every earlier occurrence of the
    last n tokens (most recent first) contributes the up-to-k tokens that followed it; distinct
    continuations only, at most n_candidates."""
    h = list(history)
    out: List[List[int]] = []
    for n in range(min(ngram_max, len(h) - 1), ngram_min - 1, -1):
        pat = h[len(h) - n:]
        for e in range(len(h) - 2, n - 2, -1):          # match end (inclusive), latest first
            if h[e + 1 - n:e + 1] == pat:
                cont = h[e + 1:e + 1 + k]
                if cont and cont not in out:
                    out.append(cont)
                    if len(out) >= n_candidates:
                        return out
    return out


def accept_sample(probs, tokens: Sequence[int], parents: Sequence[int], length: int, k: int,
                  u_accept: Sequence[float], u_final: float) -> Tuple[List[int], List[int]]:
    """Acceptance walk under sampling for candidates that carry no draft distribution (prompt
    lookup, proposals made outside): SpecInfer's multi-step speculative sampling (Miao et al.,
    2023, section 4.3, Algorithm 2) with a point mass as every candidate's proposal
    distribution.  With q = one-hot(x): the acceptance probability min(1, r(x) / q(x)) is r(x)
    and the residual norm(max(r - q, 0)) is r with x removed, so at a node with target
    distribution r the children are tried in row order, child x stands with probability
    r(x) / mass(r), a rejected child's token leaves r, and when no child stands the emitted token
    is drawn from what is left of r.  probs: [rows, V] float32 (the target's distribution after
    every tree row); u_accept: one uniform per row.  float64 arithmetic on float32 inputs, the
    mass written as 1 - (sum of the removed entries).  Returns (emitted tokens, accepted rows).
    Parity unpinned (no fixture anywhere covers it): the tests pin the emitted distribution
    against the target's instead."""
    import numpy as np
    from oracle.sampling import inverse_cdf
    p = np.asarray(probs, dtype=np.float32)
    cur, out, path = 0, [], []
    while True:
        removed, excl, nxt = np.float64(0.0), [], -1
        if len(path) < k:
            for c in range(1, length):
                if parents[c] != cur:
                    continue
                x = int(tokens[c])
                if x in excl:       # a rejected sibling carried this token: its mass is gone
                    continue
                px = np.float64(p[cur, x])
                if np.float64(np.float32(u_accept[c])) * (np.float64(1.0) - removed) < px:
                    nxt = c
                    break
                removed += px
                excl.append(x)
        if nxt < 0:
            r = p[cur].astype(np.float64)
            r[excl] = 0.0
            out.append(int(inverse_cdf(r, float(np.float32(u_final)), fallback=p[cur])))
            return out, path
        out.append(int(tokens[nxt]))
        path.append(nxt)
        cur = nxt
