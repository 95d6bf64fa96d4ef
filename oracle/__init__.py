"""CPU oracle for the speculative-decoding hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  The product
package (``vllm_omni_draft_amd``) never imports it and has no CPU fallback.

Parity status: UNPINNED BY THE REFERENCE.  The reference tree has no
speculative-decoding code, tests or golden vectors (its only pytest files,
/root/reference/src/CI_test/tests/test_omni_llm.py:233-543 and
/root/reference/src/CI_test/tests/test_end2end.py:81-157, mock the engine and
assert no tensor values), and the dependency that holds the algorithm
(``vllm>=0.2.0``, /root/reference/pyproject.toml:31) is not installed.  The
oracle therefore restates the published algorithm (Leviathan et al. 2023,
arXiv 2211.17192, Algorithm 1) and is pinned instead by
  * the defining invariant: greedy speculative decoding emits exactly the
    greedy autoregressive stream of the target model, for any draft;
  * an independent batch-1 implementation that ships in this image,
    ``transformers.generation.utils._speculative_sampling``;
  * ``transformers`` Llama / Qwen2 / OPT forward passes on seeded random
    weights for the model arithmetic.
"""
