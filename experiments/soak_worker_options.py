"""Soak run of the randomised V1 schedule (prompts in pieces, preemption and return, early
finishes, log-probabilities) with worker options the test suite runs it without: the
speculate-or-not policy, trees grown by the draft model, speculation switched off above a batch
size, fp8 weights, prompt-lookup proposals as chains and as trees.  Greedy: every stream must stay the autoregressive one.
    python experiments/soak_worker_options.py [first_seed] [n_seeds]
"""
import sys
import traceback

sys.path.insert(0, ".")
import tests.test_gpu_fuzz as fz  # noqa: E402
from vllm_omni_draft_amd import spec_decode_worker as sw  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
plain_init = sw.SpecDecodeConfig.__init__
plain_weights = fz._weights


def with_options(**opts):
    def init(self, *a, **k):
        k.update(opts)
        plain_init(self, *a, **k)
    return init


def fp8_weights(oc, seed):
    import torch
    from oracle.models import random_state
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab, norm_eps=oc.norm_eps,
        rope_theta=oc.rope_theta, max_pos=oc.max_pos, tie_embeddings=oc.tie_embeddings)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=seed).items()}
    return weights.from_state_dict(cfg, state, "cuda", fp8=True)


variants = {
    "policy auto": (dict(speculative_policy="auto"), plain_weights),
    "draft tree": (dict(speculative_draft_tree_branches=3), plain_weights),
    "off above 1 sequence": (dict(speculative_disable_by_batch_size=1), plain_weights),
    "fp8 weights": ({}, fp8_weights),
    "prompt lookup": (dict(draft=None, ngram_prompt_lookup_max=3, ngram_prompt_lookup_min=1), plain_weights),
    "lookup trees": (dict(draft=None, ngram_prompt_lookup_max=3, ngram_prompt_lookup_min=1,
                          speculative_tree_candidates=3), plain_weights),
    "policy auto + off above 2": (dict(speculative_policy="auto", speculative_disable_by_batch_size=2), plain_weights),
}
ok = bad = 0
for seed in range(first, first + count):
    for name, (opts, wfn) in variants.items():
        sw.SpecDecodeConfig.__init__ = with_options(**opts)
        fz._weights = wfn
        for rows, graph in (("narrow", True), ("wide", False)):
            try:
                fz.test_random_v1_schedules_with_pieces_and_preemption(seed, graph, rows, 0.0)
                ok += 1
            except Exception:
                bad += 1
                print("FAIL", name, seed, rows, graph, flush=True)
                traceback.print_exc()
    print("seed", seed, "ok", ok, "bad", bad, flush=True)
sw.SpecDecodeConfig.__init__ = plain_init
fz._weights = plain_weights
print("TOTAL ok", ok, "bad", bad)
