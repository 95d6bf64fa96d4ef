"""Soak run of the kernel tests over random shapes the test suite does not list: the paged
attention against the oracle (random batch, ragged contexts, new-token counts, head layouts,
split sizes; random token trees for the masked variant), the <= 16-row and the 17..256-row linear kernels against the rounding reference
(random M, N, K, every epilogue, bf16 and fp8 weights), and the row-count invariance of the
wide family.
    python experiments/soak_kernels.py [seconds]
"""
import random
import sys
import time
import traceback

import torch

sys.path.insert(0, ".")
import tests.test_gpu_attention as ta  # noqa: E402
import tests.test_gpu_linear as tl  # noqa: E402
import tests.test_gpu_tree as tt  # noqa: E402
from vllm_omni_draft_amd import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rnd = random.Random(2024)
t_end = time.time() + budget
n_ok = n_bad = 0


def attempt(name, fn, *args):
    global n_ok, n_bad
    try:
        fn(*args)
        n_ok += 1
    except Exception:
        n_bad += 1
        print("FAIL", name, args, flush=True)
        traceback.print_exc()


def attention_case():
    D = rnd.choice([64, 128])
    Hkv = rnd.choice([1, 2, 4, 8])
    G = rnd.choice([1, 2, 4, 7, 8])
    q_len = rnd.randint(1, max(1, min(16, 64 // G)))
    B = rnd.randint(1, 4)
    ctx = [rnd.choice([0, 1, 15, 16, 17, rnd.randint(0, 1500)]) for _ in range(B)]
    split = rnd.choice([128, 256, 512])
    attempt("attention", ta.test_paged_attention_matches_oracle, B, ctx, q_len, Hkv * G, Hkv, D, split)


def narrow_case():
    M = rnd.randint(1, 16)
    N = rnd.choice([16, 48, 64, 130, 257, 1000, 2048, rnd.randint(1, 3000)])
    K = 32 * rnd.randint(1, 200)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W = W2[:N].contiguous()
    bias = (torch.randn(2 * N, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)

    def run():
        for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_RELU, ops.EPI_F32, ops.EPI_GELU_TANH):
            b = rnd.choice([None, bias[:N].contiguous()])
            got = ops.linear(W, x, b, res if epi == ops.EPI_RESIDUAL else None, epi).float()
            torch.testing.assert_close(got, tl._ref_linear(W, x, b, res, epi, ops), rtol=2e-2, atol=2e-2)
        got = ops.linear(W2, x, bias, None, ops.EPI_SILU_MUL).float()
        torch.testing.assert_close(got, tl._ref_linear(W2, x, bias, None, ops.EPI_SILU_MUL, ops), rtol=3e-2, atol=3e-2)
    attempt(f"narrow M={M} N={N} K={K}", run)


def wide_case():
    M = rnd.randint(17, 256)
    big = rnd.random() < 0.3          # a matrix of >= 1024 tiles: the four-wave kernel's cases
    N = rnd.choice([16400, 20000, 33000]) if big else rnd.choice([64, 130, 257, 1024, rnd.randint(1, 4000)])
    K = (256 if big else 32) * rnd.randint(1, 8 if big else 120)
    fp8 = rnd.random() < 0.3 and K % 64 == 0
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = (torch.randn(N, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    if fp8:
        q, sc = ops.quantize_fp8(W)
        Wt = ops.tile_weights(q, ops.TILE_PLAIN, scale=sc)
        Wref = (q.float() * sc[:, None]).to(torch.bfloat16) if hasattr(q, "float") else W
    else:
        Wt, Wref = ops.tile_weights(W), W

    def run():
        for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_F32):
            got = ops.linear_wide(Wt, x, bias, res if epi == ops.EPI_RESIDUAL else None, epi)
            if not fp8:
                torch.testing.assert_close(got.float(), tl._ref_linear(Wref, x, bias, res, epi, ops), rtol=2e-2, atol=2e-2)
            m0 = rnd.randint(0, M - 1)
            n = rnd.randint(1, min(64, M - m0))
            part = ops.linear_wide(Wt, x[m0:m0 + n].contiguous(), bias,
                                   res[m0:m0 + n].contiguous() if epi == ops.EPI_RESIDUAL else None, epi)
            assert torch.equal(part, got[m0:m0 + n]), ("rows depend on the pass", epi, m0, n)
    attempt(f"wide M={M} N={N} K={K} fp8={fp8}", run)


def tree_attention_case():
    D = rnd.choice([64, 128])
    Hkv = rnd.choice([1, 2, 8])
    G = rnd.choice([1, 2, 4])
    n = rnd.randint(1, max(1, min(15, 64 // G - 1)))
    parents = [rnd.randint(0, i) for i in range(n)]          # row i + 1 hangs under a row <= i
    B = rnd.randint(1, 3)
    ctx = [rnd.choice([1, 15, 16, 17, rnd.randint(1, 1200)]) for _ in range(B)]
    attempt("tree attention", tt.test_tree_attention_matches_oracle, Hkv * G, Hkv, D, ctx, rnd.choice([128, 256]), parents)


while time.time() < t_end:
    attention_case()
    tree_attention_case()
    narrow_case()
    wide_case()
    if (n_ok + n_bad) % 30 == 0:
        print("cases", n_ok + n_bad, "bad", n_bad, flush=True)
print("TOTAL ok", n_ok, "bad", n_bad)
