"""Properties of the oracle's speculative loop that hold by construction."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.models import OracleConfig, OracleLM, random_state
from oracle.spec_decode import ar_generate, spec_generate

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _pair(seed_t=1, seed_d=2):
    tc = OracleConfig(arch="llama", n_layers=2, hidden=64, n_heads=2, n_kv_heads=1, head_dim=32,
                      ffn=96, vocab=50, max_pos=128)
    dc = OracleConfig(arch="opt", n_layers=1, hidden=32, n_heads=1, n_kv_heads=1, head_dim=32,
                      ffn=48, vocab=50, max_pos=128, tie_embeddings=True)
    return (OracleLM(tc, random_state(tc, seed_t)), OracleLM(dc, random_state(dc, seed_d)))


@pytest.mark.parametrize("k", [1, 3, 5])
def test_greedy_spec_equals_ar_for_any_draft(k):
    target, draft = _pair()
    prompt = [3, 14, 15, 9, 2, 6]
    ar = ar_generate(target, prompt, 24)
    sp, st = spec_generate(target, draft, prompt, 24, k=k)
    assert sp == ar
    assert st.emitted == st.accepted + st.steps and st.draft == k * st.steps


def test_adversarial_and_perfect_proposals():
    target, draft = _pair()
    prompt = [1, 2, 3]
    ar = ar_generate(target, prompt, 20)
    truth = prompt + ar
    # always-wrong proposals: one token per step, still the AR stream
    sp, st = spec_generate(target, draft, prompt, 20, k=4,
                           propose_hook=lambda i, pos, tok: (truth[pos] + 1) % 50 if pos < len(truth) else 0)
    assert sp == ar and st.accepted == 0 and st.steps == 19
    # perfect proposals: k+1 tokens per step (known answer: accept-len == k)
    sp, st = spec_generate(target, draft, prompt, 21, k=4,
                           propose_hook=lambda i, pos, tok: truth[pos] if pos < len(truth) else tok)
    assert sp == ar_generate(target, prompt, 21)
    assert st.accepted >= 4 * (st.steps - 1)


def test_draft_equal_to_target_accepts_everything():
    target, _ = _pair()
    sp, st = spec_generate(target, target, [5, 4, 3, 2], 16, k=3)
    assert st.accepted == 3 * st.steps                     # accept-len == k exactly
    assert sp == ar_generate(target, [5, 4, 3, 2], 16)


def test_sampling_spec_runs_and_is_seed_deterministic():
    target, draft = _pair()
    a, _ = spec_generate(target, draft, [7, 8, 9], 12, k=3, temperature=1.0,
                         rng=np.random.default_rng(0))
    b, _ = spec_generate(target, draft, [7, 8, 9], 12, k=3, temperature=1.0,
                         rng=np.random.default_rng(0))
    assert a == b and len(a) == 12 and all(0 <= t < 50 for t in a)


def test_golden_e2e_tiny():
    with open(os.path.join(GOLDEN, "e2e_tiny.json")) as f:
        g = json.load(f)
    tc, dc = OracleConfig(**g["target"]), OracleConfig(**g["draft"])
    bf = lambda st: {k: v.to(torch.bfloat16).float() for k, v in st.items()}
    target = OracleLM(tc, bf(random_state(tc, g["target_seed"])), bf16_acts=True)
    draft = OracleLM(dc, bf(random_state(dc, g["draft_seed"])), bf16_acts=True)
    for case in g["cases"][:3]:
        sp, st = spec_generate(target, draft, case["prompt"], len(case["tokens"]), k=g["k"])
        assert sp == case["tokens"]
        assert (st.steps, st.accepted) == (case["steps"], case["accepted"])
