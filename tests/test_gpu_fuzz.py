"""Randomised schedules on the GPU: requests join and leave a running batch at random steps.
Whatever the schedule, every request's greedy stream must equal the stream the same request
gets alone from an autoregressive engine of the same kernel family (rows are independent of
their batch neighbours bit for bit; speculation is lossless)."""
import random

import pytest
import torch

from oracle.models import OracleConfig, random_state

pytestmark = pytest.mark.gpu

TARGET = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=8, n_kv_heads=2, head_dim=64,
                      ffn=512, vocab=512, max_pos=512)
DRAFT = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                     ffn=256, vocab=512, max_pos=512, tie_embeddings=True)


def _weights(oc, seed):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        norm_eps=oc.norm_eps, rope_theta=oc.rope_theta, max_pos=oc.max_pos,
        tie_embeddings=oc.tie_embeddings)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=seed).items()}
    return weights.from_state_dict(cfg, state, "cuda")


@pytest.mark.parametrize("rows,max_batch,k,use_graph,seed", [
    ("narrow", 2, 5, True, 0), ("narrow", 2, 3, False, 1), ("wide", 6, 3, True, 2),
    ("wide", 6, 5, False, 3), ("wide", 20, 2, True, 4)])
def test_random_join_leave_schedule(rows, max_batch, k, use_graph, seed):
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    rnd = random.Random(seed)
    tw, dw = _weights(TARGET, 50), _weights(DRAFT, 51)
    n_req, n_new = 3 * max_batch if max_batch <= 6 else 30, 28
    prompts = [[rnd.randrange(TARGET.vocab) for _ in range(rnd.randint(1, 70))] for _ in range(n_req)]

    def cfg(kk, mb):
        return EngineConfig(k=kk, max_batch=mb, max_seq_len=140, split_tokens=128, use_graph=use_graph,
                            rows=rows)

    # every request alone, autoregressive, same kernel family
    solo = SpecEngine(tw, None, cfg(0, 1))
    want = []
    for p in prompts:
        solo.reset()
        out = [solo.add_request(0, p)]
        while len(out) < n_new:
            out.extend(solo.step()[0])
        want.append(out[:n_new])
    del solo

    eng = SpecEngine(tw, dw, cfg(k, max_batch))
    waiting = list(range(n_req))
    running = []           # request index per slot
    got = {i: [] for i in range(n_req)}
    steps = 0
    while waiting or running:
        # admit a random number of waiting requests into the free slots
        while waiting and len(running) < max_batch and (not running or rnd.random() < 0.6):
            i = waiting.pop(rnd.randrange(len(waiting)))
            got[i].append(eng.add_request(len(running), prompts[i]))
            running.append(i)
        for slot, new in enumerate(eng.step()):
            got[running[slot]].extend(new)
        steps += 1
        # finished requests leave (highest slot first: the last sequence moves into the hole)
        for slot in sorted((s for s, i in enumerate(running) if len(got[i]) >= n_new), reverse=True):
            eng.remove_request(slot)
            last = len(running) - 1
            if slot != last:
                running[slot] = running[last]
            running.pop()
        assert steps < 2000
    for i in range(n_req):
        assert got[i][:n_new] == want[i], i
    m = eng.metrics()
    assert m["draft"] > 0 and m["emitted"] >= m["accepted"]
