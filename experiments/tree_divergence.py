"""Step by step: root-row logits of a tree engine against the autoregressive logits at the same
position (bit-exact while the cached context is the same), to find the step whose cache update
first makes them differ."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine

name, n_prompt, k = "llama-3-8b", 1024, 5
dev = torch.device("cuda", 0)
cfg = config.preset(name)
w = weights.random_init(cfg, seed=1, device=dev)
rng = np.random.default_rng(0)
prompt = [int(t) for t in rng.integers(0, cfg.vocab, n_prompt)]
n_new = 120
def engine(**kw):
    return SpecEngine(w, None, EngineConfig(k=kw.pop("k", k), max_batch=1, max_seq_len=n_prompt + n_new + 32,
                                            use_graph=False, **kw))
ar = engine(k=0)
truth = [ar.add_request(0, prompt)]
ar_logits = []          # ar_logits[i]: logits after truth[i] (predicting truth[i+1])
while len(truth) < n_new:
    truth += ar.step()[0]
    ar_logits.append(ar.t_logits[0].clone())
del ar
eng = engine(external_proposals=True, tree_candidates=3)
out = [eng.add_request(0, prompt)]
draws = np.random.default_rng(1).random((400, 3))
i = 0
prev = None
while len(out) < n_new - 8:
    nxt = truth[len(out):len(out) + k]
    c = [[(t + 1 + j) % cfg.vocab for t in nxt] for j in range(3)]
    hit = next((j for j in range(3) if draws[i % 400][j] < [0.45, 0.25, 0.15][j]), None)
    if hit is not None:
        c[hit] = list(nxt)
    root = n_prompt + len(out) - 1
    new = eng.step(proposals=[c])[0]
    rows = [0] + [r for r in eng.tree_path[0].tolist() if r >= 0]
    diffs = [(eng.t_logits[r] - ar_logits[len(out) - 1 + d]).abs().max().item() for d, r in enumerate(rows)]
    print(f"step {i}: root slot {root} hit {hit} path {rows[1:]} emitted {len(new)} diffs {['%.4f' % x for x in diffs]}")
    if any(x > 0 for x in diffs) or out + new != truth[:len(out) + len(new)]:
        print("previous step:", prev)
        break
    prev = (root, hit, rows[1:], len(new))
    out += new
    i += 1
