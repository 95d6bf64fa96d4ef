"""Speculative decoding end to end through the worker and its scheduler.

    python examples/generate.py                                  # Llama-3-8B + 3.2-1B shapes, random weights
    python examples/generate.py --target-dir /models/Llama-3-8B --draft-dir /models/Llama-3.2-1B \\
                                --prompt-ids 128000 9906 1917

With --target-dir / --draft-dir the models are read from local Hugging Face checkpoints
(config.json + *.safetensors); without them, seeded random weights of the named presets stand in
(the output is then meaningless, the timing is not).  Tokenisation is the caller's: prompts are
token ids.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler  # noqa: E402
from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker  # noqa: E402


def load(model_dir, preset, vocab=None, seed=0, fp8=False):
    if model_dir:
        cfg = config.load_hf_config(model_dir)
        return weights.from_safetensors_dir(cfg, model_dir, fp8=fp8)
    cfg = config.preset(preset, **({"vocab": vocab} if vocab else {}))
    return weights.random_init(cfg, seed=seed, fp8=fp8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target-dir")
    ap.add_argument("--draft-dir")
    ap.add_argument("--target", default="llama-3-8b", help="preset name when no --target-dir is given")
    ap.add_argument("--draft", default="llama-3.2-1b")
    ap.add_argument("--k", type=int, default=5, help="proposals per step")
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--max-num-seqs", type=int, default=4)
    ap.add_argument("--fp8", action="store_true", help="quantise both models' matrices to fp8 at load")
    ap.add_argument("--prompt-ids", type=int, nargs="*", help="one prompt as token ids (default: four random ones)")
    args = ap.parse_args()

    target = load(args.target_dir, args.target, fp8=args.fp8)
    draft = load(args.draft_dir, args.draft, vocab=target.cfg.vocab, seed=1, fp8=args.fp8)
    worker = SpecDecodeWorker(SpecDecodeConfig(target=target, draft=draft, num_speculative_tokens=args.k,
                                               max_num_seqs=args.max_num_seqs, max_model_len=2048))
    worker.init_device()
    worker.load_model()
    sched = SpecDecodeScheduler.for_worker(worker)

    if args.prompt_ids:
        prompts = [list(args.prompt_ids)]
    else:
        g = torch.Generator().manual_seed(0)
        prompts = [torch.randint(0, target.cfg.vocab, (n,), generator=g).tolist() for n in (12, 200, 57, 1000)]
    sched.generate(prompts[:1], max_tokens=4)            # warm-up: graphs of the common shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reqs = [Request(f"r{i}", p, max_tokens=args.max_tokens) for i, p in enumerate(prompts)]
    for r in reqs:
        sched.add_request(r)
    while sched.has_unfinished():
        sched.step()
    sched.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(len(r.output_token_ids) for r in reqs)
    m = worker.engine.metrics()
    for r in reqs:
        print(f"{r.request_id}: prompt of {len(r.prompt_token_ids)} tokens -> {r.output_token_ids[:12]}"
              f"{' ...' if len(r.output_token_ids) > 12 else ''} ({r.finish_reason})")
    print(f"{n} tokens in {dt * 1e3:.1f} ms ({n / dt:.0f} tokens/s, prefill included); "
          f"draft acceptance {m['accepted'] / max(m['draft'], 1):.2f}")


if __name__ == "__main__":
    main()
