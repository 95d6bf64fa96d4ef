"""Generates the committed golden vectors from the CPU oracle.

    python tests/golden/gen_golden.py

The reference repository holds no fixtures for this path (its tests assert no
tensor values), so these vectors pin the oracle against itself over time and
give the HIP kernels fixed inputs with known answers; the oracle in turn is
cross-checked against `transformers` in tests/test_oracle_*.py.
Acceptance cases are tie-guarded: no uniform lies within 1e-4 (relative) of
its acceptance threshold or of an inverse-CDF boundary, so f32 reassociation
on another device cannot flip an answer.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sampling  # noqa: E402
from oracle.models import OracleConfig, OracleLM, random_state  # noqa: E402
from oracle.spec_decode import ar_generate, spec_generate  # noqa: E402


def guarded(p, q, ids, ua, ur, lens):
    """True if every decision of this case has a safe margin."""
    B, k1, V = p.shape
    k = k1 - 1
    for b in range(B):
        L = k if lens is None else int(lens[b])
        j = 0
        while j < L:
            x = ids[b, j]
            a, c = float(ua[b, j]) * float(q[b, j, x]), float(p[b, j, x])
            if abs(a - c) <= 1e-4 * max(a, c, 1e-30):
                return False
            if not a < c:
                break
            j += 1
        if j == L and L == k:
            continue
        w = np.maximum(p[b, j] - q[b, j], 0) if j < L else p[b, j]
        c = np.cumsum(w.astype(np.float64))
        if c[-1] <= 0:
            return False
        thr = float(ur[b, j]) * c[-1]
        if np.min(np.abs(c - thr)) <= 1e-6 * c[-1]:
            return False
    return True


def rejection_cases():
    rng = np.random.default_rng(2024)
    out = {}
    n = 0
    specs = [(1, 1, 64, None), (1, 3, 64, None), (4, 3, 64, None), (4, 5, 64, None),
             (32, 5, 64, None), (2, 5, 2048, None), (4, 3, 64, "ragged"), (3, 5, 256, "close"),
             (4, 5, 64, "qzero"), (2, 1, 2048, "close")]
    for (B, k, V, kind) in specs:
        while True:
            tl = rng.standard_normal((B, k + 1, V)).astype(np.float32) * 2
            noise = 0.1 if kind == "close" else 1.0
            dl = tl[:, :k] + rng.standard_normal((B, k, V)).astype(np.float32) * noise
            p = np.stack([sampling.softmax_rows(tl[b]) for b in range(B)])
            q = np.stack([sampling.softmax_rows(dl[b]) for b in range(B)])
            ids = np.array([[sampling.inverse_cdf(q[b, i], rng.random()) for i in range(k)]
                            for b in range(B)], dtype=np.int64)
            if kind == "qzero":  # proposals the draft itself gave zero mass
                for b in range(B):
                    q[b, 0, ids[b, 0]] = 0.0
            lens = rng.integers(0, k + 1, size=B).astype(np.int32) if kind == "ragged" else None
            ua = rng.random((B, k), dtype=np.float32)
            ur = rng.random((B, k + 1), dtype=np.float32)
            bonus = rng.integers(0, V, size=B).astype(np.int64)
            if guarded(p, q, ids, ua, ur, lens):
                break
        o, c = sampling.rejection_sample(p, bonus, q, ids, ua, ur, lens)
        pre = f"c{n}_"
        out.update({pre + "target_probs": p, pre + "draft_probs": q, pre + "draft_ids": ids,
                    pre + "bonus_ids": bonus, pre + "u_accept": ua, pre + "u_recover": ur,
                    pre + "proposal_lens": np.zeros(0, np.int32) if lens is None else lens,
                    pre + "out_ids": o, pre + "counters": c})
        n += 1
    out["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "rejection_cases.npz"), **out)


def typical_cases():
    """Typical-acceptance cases; every tested position is decided by more than 1e-4."""
    rng = np.random.default_rng(4242)
    out = {}
    n = 0
    specs = [(1, 1, 64, None, 2.0), (4, 3, 64, None, 2.0), (4, 5, 256, None, 0.5),
             (32, 5, 64, None, 1.0), (2, 5, 2048, None, 1.0), (4, 3, 64, "ragged", 2.0),
             (3, 5, 512, "peaked", 6.0), (3, 4, 128, "flat", 0.05)]
    for (B, k, V, kind, scale) in specs:
        while True:
            tl = rng.standard_normal((B, k + 1, V)).astype(np.float32) * scale
            p = np.stack([sampling.softmax_rows(tl[b]) for b in range(B)])
            # proposals: the target's own top tokens half of the time, random otherwise
            ids = np.where(rng.random((B, k)) < 0.5, np.argmax(p[:, :k], axis=-1),
                           rng.integers(0, V, size=(B, k))).astype(np.int64)
            lens = rng.integers(0, k + 1, size=B).astype(np.int32) if kind == "ragged" else None
            bonus = rng.integers(0, V, size=B).astype(np.int64)
            margins = []
            o, c = sampling.typical_acceptance(p, bonus, ids, 0.09, 0.3, lens, margin_out=margins)
            if not margins or min(margins) > 1e-4:
                break
        pre = f"c{n}_"
        out.update({pre + "target_probs": p, pre + "draft_ids": ids, pre + "bonus_ids": bonus,
                    pre + "proposal_lens": np.zeros(0, np.int32) if lens is None else lens,
                    pre + "out_ids": o, pre + "counters": c})
        n += 1
    out["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "typical_cases.npz"), **out)


def greedy_cases():
    rng = np.random.default_rng(7)
    out = {}
    n = 0
    for (B, k, V) in [(1, 1, 64), (1, 3, 64), (4, 3, 64), (4, 5, 512), (32, 5, 64), (2, 5, 4096)]:
        while True:
            logits = rng.standard_normal((B, k + 1, V)).astype(np.float32)
            top2 = np.sort(logits, axis=-1)[..., -2:]
            if np.min(top2[..., 1] - top2[..., 0]) > 1e-3:   # tie-margin guard
                break
        t_ids = sampling.argmax_rows(logits)
        d_ids = t_ids[:, :k].copy()
        for b in range(B):   # all-accept, first-position reject, mid reject, ...
            mode = b % 4
            if mode == 1:
                d_ids[b, 0] = (d_ids[b, 0] + 1) % V
            elif mode == 2 and k > 1:
                d_ids[b, k // 2] = (d_ids[b, k // 2] + 1) % V
            elif mode == 3:
                d_ids[b, k - 1] = (d_ids[b, k - 1] + 3) % V
        o, c = sampling.greedy_verify(t_ids, d_ids)
        pre = f"c{n}_"
        out.update({pre + "logits": logits, pre + "target_ids": t_ids, pre + "draft_ids": d_ids,
                    pre + "out_ids": o, pre + "counters": c})
        n += 1
    out["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "greedy_cases.npz"), **out)


TINY_TARGET = dict(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64,
                   ffn=256, vocab=256, norm_eps=1e-5, rope_theta=10000.0, max_pos=256)
TINY_DRAFT = dict(arch="llama", n_layers=1, hidden=64, n_heads=2, n_kv_heads=1, head_dim=64,
                  ffn=128, vocab=256, norm_eps=1e-5, rope_theta=10000.0, max_pos=256)


def e2e_tiny():
    """2-layer target + 1-layer draft, seeded weights rounded to bf16, 8 prompts x
    32 new tokens, k = 3: the autoregressive stream and the speculative stream
    (identical by construction) with the acceptance counts."""
    tc, dc = OracleConfig(**TINY_TARGET), OracleConfig(**TINY_DRAFT)
    bf = lambda st: {k: v.to(torch.bfloat16).float() for k, v in st.items()}
    target = OracleLM(tc, bf(random_state(tc, seed=100)), bf16_acts=True)
    draft = OracleLM(dc, bf(random_state(dc, seed=200)), bf16_acts=True)
    rng = np.random.default_rng(1234)
    cases = []
    for i in range(8):
        prompt = rng.integers(0, tc.vocab, size=int(rng.integers(3, 20))).tolist()
        ar = ar_generate(target, prompt, 32)
        sp, st = spec_generate(target, draft, prompt, 32, k=3)
        assert sp == ar, "greedy speculative stream differs from the autoregressive stream"
        cases.append({"prompt": prompt, "tokens": ar, "steps": st.steps, "accepted": st.accepted})
    with open(os.path.join(HERE, "e2e_tiny.json"), "w") as f:
        json.dump({"target": TINY_TARGET, "draft": TINY_DRAFT, "target_seed": 100, "draft_seed": 200,
                   "k": 3, "cases": cases}, f)


if __name__ == "__main__":
    rejection_cases()
    typical_cases()
    greedy_cases()
    e2e_tiny()
    print("golden vectors written to", HERE)
