"""Token trees against a single chain of proposals, verify side only (no draft model):

    python experiments/bench_tree.py [model] [prompt_len] [steps]

The target's own greedy stream is generated first.  Every step the "proposer" hands the engine
candidate chains of k tokens built from that stream: candidate c carries the true continuation
with probability P[c] (independently per step, first hit wins), every other candidate is wrong
from its first token.  A chain engine sees candidate 0 only; the tree engine sees all of them in
one verify pass.  Prints ms per step and tokens per step for both.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

P = [0.45, 0.25, 0.15]     # chance that candidate c is the true continuation (first hit wins)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
    n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    k = 5
    dev = torch.device("cuda", 0)
    cfg = config.preset(name)
    w = weights.random_init(cfg, seed=1, device=dev)
    rng = np.random.default_rng(0)
    prompt = [int(t) for t in rng.integers(0, cfg.vocab, n_prompt)]
    n_new = steps * (k + 1) + 8

    def engine(**kw):
        return SpecEngine(w, None, EngineConfig(k=kw.pop("k", k), max_batch=1,
                                                max_seq_len=n_prompt + n_new + 16, **kw))

    ar_eng = engine(k=0)
    truth = [ar_eng.add_request(0, prompt)]
    t0 = time.perf_counter()
    while len(truth) < n_new:
        truth += ar_eng.step()[0]
    ar_ms = (time.perf_counter() - t0) * 1e3 / (len(truth) - 1)
    del ar_eng
    print(f"{name} prompt {n_prompt}: autoregressive {ar_ms:.3f} ms/token", flush=True)

    def candidates(pos, draw):
        nxt = truth[pos:pos + k]
        out, hit = [], False
        for c, p in enumerate(P):
            if not hit and draw[c] < p:
                out.append(list(nxt))
                hit = True
            else:
                out.append([(t + 1 + c) % cfg.vocab for t in nxt])
        return out

    draws = np.random.default_rng(1).random((steps + 8, len(P)))
    for label, kw, take in (("chain (candidate 0 only)", dict(external_proposals=True), 1),
                            ("tree of 3 candidates", dict(external_proposals=True, tree_candidates=3), 3),
                            ("tree cut to 11 nodes", dict(external_proposals=True, tree_candidates=3, tree_nodes=11), 3),
                            ("tree cut to 7 nodes", dict(external_proposals=True, tree_candidates=3, tree_nodes=7), 3)):
        eng = engine(**kw)
        out = [eng.add_request(0, prompt)]
        for i in range(4):                       # warm-up: graph capture
            c = candidates(len(out), draws[steps + i])
            out += eng.step(proposals=[c if take > 1 else c[0]])[0]
        torch.cuda.synchronize()
        # Tokens per step are counted while the stream is the autoregressive one: a tree row keeps
        # its ancestors at other cache slots than a chain, sums run in another order, and at a
        # near-tie of the target's logits (frequent with random weights) the tree engine may emit
        # the other token; from there on `truth` is no longer its continuation.
        t0 = time.perf_counter()
        toks = on_steps = 0
        on_truth = out == truth[:len(out)]
        for i in range(steps):
            c = candidates(len(out), draws[i])
            new = eng.step(proposals=[c if take > 1 else c[0]])[0]
            out += new
            if on_truth and out == truth[:len(out)]:
                toks += len(new)
                on_steps += 1
            else:
                on_truth = False
        dt = time.perf_counter() - t0
        ms = dt * 1e3 / steps
        per = toks / max(on_steps, 1)
        print(f"  {label:26s} {ms:.3f} ms/step, {per:.2f} tokens/step over the {on_steps} steps on the "
              f"autoregressive stream, {per / ms * 1e3:.1f} tok/s ({per / ms * ar_ms:.2f}x autoregressive), "
              f"verify rows {eng.tree_nodes + 1 if eng.tree else k + 1}", flush=True)
        del eng


if __name__ == "__main__":
    main()
