"""Autoregressive and speculative generation loops on the CPU oracle models.
TEST INFRASTRUCTURE ONLY.

One speculative step (Leviathan et al. 2023, Alg. 1; the driver loop has the
shape of transformers' `_assisted_decoding`,
transformers/generation/utils.py:3562): the draft proposes k tokens one at a
time, the target scores the k proposals plus the bonus position in ONE pass,
the accepted prefix plus one more token is emitted, and both caches are rolled
back to the emitted length.  The reference repository contains no such loop
(/root/reference/src/doc/engine_core.md:117-122 only gives the
schedule -> execute_model -> update_from_output step contract it must fit).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import sampling
from .models import OracleLM


def ar_generate(model: OracleLM, prompt: List[int], n_new: int, temperature: float = 0.0,
                uniforms: Optional[np.ndarray] = None) -> List[int]:
    """Greedy (temperature 0) or sampled continuation of `prompt`."""
    cache = model.new_cache()
    logits = model.forward(prompt, 0, cache)[-1]
    out: List[int] = []
    for i in range(n_new):
        if temperature == 0.0:
            tok = int(torch.argmax(logits))
        else:
            p = sampling.softmax_rows(logits.numpy()[None], temperature)[0]
            tok = sampling.inverse_cdf(p, float(uniforms[i]))
        out.append(tok)
        if i + 1 < n_new:
            logits = model.forward([tok], len(prompt) + i, cache)[-1]
    return out


class SpecStats:
    def __init__(self):
        self.steps = 0
        self.accepted = 0
        self.emitted = 0
        self.draft = 0

    @property
    def accept_len(self) -> float:
        return self.accepted / max(self.steps, 1)


def spec_generate(target: OracleLM, draft: OracleLM, prompt: List[int], n_new: int, k: int,
                  temperature: float = 0.0, rng: Optional[np.random.Generator] = None,
                  propose_hook: Optional[Callable[[int, int, int], int]] = None
                  ) -> Tuple[List[int], SpecStats]:
    """Speculative continuation.  Greedy when temperature == 0, else draft
    tokens are drawn from q and accepted by modified rejection sampling with
    uniforms from `rng`.  `propose_hook(step, position, token)` may replace a
    proposal (used to test with adversarial drafts)."""
    stats = SpecStats()
    seq = list(prompt)
    t_cache, d_cache = target.new_cache(), draft.new_cache()
    # Prefill both models on all but the last prompt token; the first generated
    # token comes from the target like any other step's bonus/recovered token.
    first_logits = target.forward(seq, 0, t_cache)[-1]
    draft.forward(seq, 0, d_cache)
    if temperature == 0.0:
        seq.append(int(torch.argmax(first_logits)))
    else:
        p0 = sampling.softmax_rows(first_logits.numpy()[None], temperature)[0]
        seq.append(sampling.inverse_cdf(p0, float(rng.random(dtype=np.float32))))
    n_prompt = len(prompt)
    while len(seq) - n_prompt < n_new:
        n = len(seq)
        # caches hold tokens 0..n-2 at most; the draft may lag by one (bonus case)
        t_cache.truncate(n - 1)
        d_cache.truncate(min(len(d_cache), n - 1))
        proposals: List[int] = []
        q_rows = []
        feed_from = len(d_cache)
        cur = seq[feed_from:]
        for i in range(k):
            lg = draft.forward(cur, feed_from, d_cache)[-1]
            if temperature == 0.0:
                tok = int(torch.argmax(lg))
            else:
                q = sampling.softmax_rows(lg.numpy()[None], temperature)[0]
                tok = sampling.inverse_cdf(q, float(rng.random(dtype=np.float32)))
                q_rows.append(q)
            if propose_hook is not None:
                tok = propose_hook(i, n + i, tok)
            proposals.append(tok)
            feed_from = n + i
            cur = [tok]
        t_logits = target.forward([seq[-1]] + proposals, n - 1, t_cache)  # [k+1, V]
        if temperature == 0.0:
            t_ids = sampling.argmax_rows(t_logits.numpy())[None]
            out, cnt = sampling.greedy_verify(t_ids, np.array([proposals]))
        else:
            p = sampling.softmax_rows(t_logits.numpy(), temperature)[None]
            bonus = sampling.inverse_cdf(p[0, k], float(rng.random(dtype=np.float32)))
            ua = rng.random((1, k), dtype=np.float32)
            ur = rng.random((1, k + 1), dtype=np.float32)
            out, cnt = sampling.rejection_sample(p, [bonus], np.stack(q_rows)[None],
                                                 np.array([proposals]), ua, ur)
        emitted = [int(t) for t in out[0] if t >= 0]
        seq.extend(emitted)
        stats.steps += 1
        stats.accepted += int(cnt[0])
        stats.emitted += int(cnt[1])
        stats.draft += int(cnt[2])
    return seq[n_prompt:n_prompt + n_new], stats
