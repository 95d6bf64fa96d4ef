"""HIP acceptance / sampling kernels against the CPU oracle (bit-exact ids)."""
import os

import numpy as np
import pytest
import torch

from oracle import sampling

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_library_loads_on_gpu():
    cus, name = _ops().device_info()
    assert cus > 0 and "gfx" in name


def test_rejection_golden():
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "rejection_cases.npz"))
    for i in range(int(g["n_cases"])):
        pl = g[f"c{i}_proposal_lens"]
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.rejection_sample(
            cu(g[f"c{i}_target_probs"]), cu(g[f"c{i}_bonus_ids"]), cu(g[f"c{i}_draft_probs"]),
            cu(g[f"c{i}_draft_ids"]), cu(g[f"c{i}_u_accept"]), cu(g[f"c{i}_u_recover"]),
            None if pl.size == 0 else cu(pl), counters)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"], err_msg=f"case {i}")
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])


@pytest.mark.parametrize("B,k,V", [(1, 5, 128256), (4, 3, 50272), (32, 5, 151936), (3, 1, 64)])
def test_rejection_random_full_vocab(B, k, V):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(B * 1000 + k)
    tl = torch.randn(B * (k + 1), V, generator=gen, device="cuda") * 3
    dl = tl.view(B, k + 1, V)[:, :k].reshape(B * k, V) + torch.randn(B * k, V, generator=gen, device="cuda")
    p = ops.softmax_rows(tl).view(B, k + 1, V)
    q = ops.softmax_rows(dl.contiguous()).view(B, k, V)
    ud = torch.rand(B * k, generator=gen, device="cuda")
    ids = ops.sample_rows(q.view(B * k, V), ud).view(B, k)
    ua = torch.rand(B, k, generator=gen, device="cuda")
    ur = torch.rand(B, k + 1, generator=gen, device="cuda")
    bonus = torch.randint(0, V, (B,), generator=gen, device="cuda")
    lens = torch.randint(0, k + 1, (B,), generator=gen, device="cuda").to(torch.int32)
    for pl in (None, lens):
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.rejection_sample(p, bonus, q, ids, ua, ur, pl, counters)
        ref, cnt = sampling.rejection_sample(p.cpu().numpy(), bonus.cpu().numpy(), q.cpu().numpy(),
                                             ids.cpu().numpy(), ua.cpu().numpy(), ur.cpu().numpy(),
                                             None if pl is None else pl.cpu().numpy())
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        np.testing.assert_array_equal(counters.cpu().numpy(), cnt)


def test_greedy_golden_and_argmax():
    ops = _ops()
    g = np.load(os.path.join(GOLDEN, "greedy_cases.npz"))
    for i in range(int(g["n_cases"])):
        logits = g[f"c{i}_logits"]
        B, k1, V = logits.shape
        t_ids = ops.argmax_rows(cu(logits).view(B * k1, V)).view(B, k1)
        np.testing.assert_array_equal(t_ids.cpu().numpy(), g[f"c{i}_target_ids"])
        counters = torch.zeros(3, dtype=torch.int64, device="cuda")
        out = ops.greedy_verify(t_ids, cu(g[f"c{i}_draft_ids"]), None, counters)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"])
        np.testing.assert_array_equal(counters.cpu().numpy(), g[f"c{i}_counters"])


def test_greedy_ragged_and_ties():
    ops = _ops()
    t = torch.tensor([[5, 6, 7, 8], [5, 6, 7, 8], [1, 2, 3, 4]], device="cuda")
    d = torch.tensor([[5, 6, 7], [5, 9, 7], [1, 2, 3]], device="cuda")
    lens = torch.tensor([1, 3, 0], dtype=torch.int32, device="cuda")
    out = ops.greedy_verify(t, d, lens)
    ref, _ = sampling.greedy_verify(t.cpu().numpy(), d.cpu().numpy(), lens.cpu().numpy())
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # argmax ties resolve to the smallest index, also across threads / waves
    x = torch.zeros(2, 128256, device="cuda")
    x[0, [70000, 123, 99999]] = 3.0
    x[1, 128255] = 1.0
    assert ops.argmax_rows(x).tolist() == [123, 128255]


@pytest.mark.parametrize("V", [64, 50272, 128256])
def test_softmax_and_sample_rows(V):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(V)
    x = torch.randn(6, V, generator=gen, device="cuda") * 4
    for T in (1.0, 0.7):
        p = ops.softmax_rows(x, T)
        ref = sampling.softmax_rows(x.cpu().numpy(), T)
        np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=2e-5, atol=1e-9)
    u = torch.rand(6, generator=gen, device="cuda")
    ids = ops.sample_rows(p, u)
    np.testing.assert_array_equal(ids.cpu().numpy(), sampling.sample_rows(p.cpu().numpy(), u.cpu().numpy()))
    # strided rows (every other row) and the extreme uniforms
    ids2 = ops.sample_rows(p, torch.tensor([0.0, 0.99999994, 0.5], device="cuda"), row_offset=1, row_step=2)
    ref2 = sampling.sample_rows(p.cpu().numpy()[1::2], [0.0, 0.99999994, 0.5])
    np.testing.assert_array_equal(ids2.cpu().numpy(), ref2)


def test_bad_arguments_raise():
    from vllm_omni_draft_amd._lib import SdError
    ops = _ops()
    p = torch.full((1, 2, 6), 1 / 6, device="cuda")          # V % 4 != 0
    with pytest.raises(SdError):
        ops.rejection_sample(p, torch.zeros(1, dtype=torch.int64, device="cuda"), p[:, :1].contiguous(),
                             torch.zeros(1, 1, dtype=torch.int64, device="cuda"),
                             torch.zeros(1, 1, device="cuda"), torch.zeros(1, 2, device="cuda"))
    with pytest.raises(SdError):                              # CPU tensors are refused: no CPU path
        ops.argmax_rows(torch.zeros(2, 8))
