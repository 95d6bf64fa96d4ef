"""Weight-streaming linear / norm / embedding kernels against fp32 torch math,
and the row-independence (batch-invariance) property that greedy parity with
autoregressive decoding rests on."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


def rb(x):
    return x.to(torch.bfloat16).float()


def _ref_linear(W, x, bias, residual, epi, ops):
    y = x.float() @ W.float().t()
    if epi == ops.EPI_SILU_MUL:
        N = W.shape[0] // 2
        g, u = y[:, :N], y[:, N:]
        if bias is not None:
            g, u = g + bias[:N].float(), u + bias[N:].float()
        return rb(rb(torch.nn.functional.silu(rb(g))) * rb(u))
    if bias is not None:
        y = y + bias.float()
    if epi == ops.EPI_F32:
        return y
    if epi == ops.EPI_RESIDUAL:
        return rb(rb(y) + residual.float())
    if epi == ops.EPI_RELU:
        return rb(torch.relu(rb(y)))
    if epi == ops.EPI_GELU_TANH:
        return rb(torch.nn.functional.gelu(rb(y), approximate="tanh"))
    return rb(y)


@pytest.mark.parametrize("M", [1, 2, 5, 6, 8, 13, 16])
@pytest.mark.parametrize("N,K", [(64, 512), (130, 768), (1024, 4096), (257, 896), (96, 14336)])
def test_linear_epilogues(M, N, K):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=gen, device="cuda")).to(torch.bfloat16)
    bias = (torch.randn(2 * N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=gen, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W = W2[:N].contiguous()
    for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_RELU, ops.EPI_F32, ops.EPI_GELU_TANH):
        for b in (None, bias[:N].contiguous()):
            got = ops.linear(W, x, b, res if epi == ops.EPI_RESIDUAL else None, epi).float()
            ref = _ref_linear(W, x, b, res, epi, ops)
            torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    for b in (None, bias):
        got = ops.linear(W2, x, b, None, ops.EPI_SILU_MUL).float()
        ref = _ref_linear(W2, x, b, None, ops.EPI_SILU_MUL, ops)
        torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)


def test_linear_f32_accuracy():
    """f32 output vs float64 reference: accumulation error only."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(0)
    W = (torch.randn(512, 4096, generator=gen, device="cuda") * 0.02).to(torch.bfloat16)
    x = torch.randn(6, 4096, generator=gen, device="cuda").to(torch.bfloat16)
    got = ops.linear(W, x, None, None, ops.EPI_F32).double()
    ref = x.double() @ W.double().t()
    assert (got - ref).abs().max().item() < 2e-5 * 4096 ** 0.5


@pytest.mark.parametrize("epi_name", ["EPI_STORE", "EPI_F32", "EPI_SILU_MUL", "EPI_RESIDUAL"])
def test_linear_rows_are_independent(epi_name):
    """Row m of an M-row call is bit-identical to the same row computed alone."""
    ops = _ops()
    epi = getattr(ops, epi_name)
    gen = torch.Generator(device="cuda").manual_seed(5)
    K, N = (14336, 384) if epi == ops.EPI_RESIDUAL else (2048, 384)   # 14336: x is chunked at M >= 5
    W = (torch.randn(2 * N if epi == ops.EPI_SILU_MUL else N, K, generator=gen, device="cuda") * 0.03
         ).to(torch.bfloat16)
    x = torch.randn(16, K, generator=gen, device="cuda").to(torch.bfloat16)
    res = torch.randn(16, N, generator=gen, device="cuda").to(torch.bfloat16)
    W = ops.tile_weights(W, ops.TILE_GATED if epi == ops.EPI_SILU_MUL else ops.TILE_PLAIN)
    full = ops.linear(W, x, None, res if epi == ops.EPI_RESIDUAL else None, epi)
    for M in (1, 2, 6, 11):
        for m0 in (0, 16 - M):
            part = ops.linear(W, x[m0:m0 + M].contiguous(), None,
                              res[m0:m0 + M].contiguous() if epi == ops.EPI_RESIDUAL else None, epi)
            assert torch.equal(part, full[m0:m0 + M]), (M, m0)


@pytest.mark.parametrize("H", [64, 768, 896, 4096, 8192])
def test_norms(H):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(H)
    x = (torch.randn(7, H, generator=gen, device="cuda") * 3).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    b = (0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    got = ops.rmsnorm(x, w, 1e-5).float()
    xf = x.float()
    ref = rb(rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)) * w.float())
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)
    got = ops.layernorm(x, w, b, 1e-5).float()
    ref = rb(torch.nn.functional.layer_norm(xf, (H,), w.float(), b.float(), 1e-5))
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("M", [1, 6, 8])
@pytest.mark.parametrize("H,N", [(768, 3072), (2048, 512), (4096, 1024), (8192, 256)])
def test_fused_norm_linear_matches_unfused(M, H, N):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(H + M)
    x = (torch.randn(M, H, generator=gen, device="cuda") * 2).to(torch.bfloat16)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    b = (0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    W = (torch.randn(2 * N, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    from vllm_omni_draft_amd._lib import SdError
    fits = lambda ln: (M + 1) * (H + 8) * 2 <= 144 * 1024 and H <= (4096 if ln else 8192)
    for ln in (False, True):
        if not fits(ln):
            # the normalised rows (and the norm parameters) must fit in LDS; callers
            # fall back to the stand-alone norm
            with pytest.raises(SdError):
                ops.norm_linear(W[:N].contiguous(), x, g, b if ln else None, 1e-5, ln, None,
                                ops.EPI_F32)
            continue
        xn = ops.layernorm(x, g, b, 1e-5) if ln else ops.rmsnorm(x, g, 1e-5)
        for epi, Wm in ((ops.EPI_F32, W[:N].contiguous()), (ops.EPI_STORE, W[:N].contiguous()),
                        (ops.EPI_SILU_MUL, W), (ops.EPI_RELU, W[:N].contiguous())):
            ref = ops.linear(Wm, xn, None, None, epi).float()
            got = ops.norm_linear(Wm, x, g, b if ln else None, 1e-5, ln, None, epi).float()
            # the same arithmetic in the same order: a layer may take either path depending
            # on how many rows a pass carries, and a row's bits must not depend on that
            assert torch.equal(got, ref), (ln, epi)
    if not fits(False):
        return
    # rows are independent in the fused kernel too
    full = ops.norm_linear(W[:N].contiguous(), x, g, None, 1e-5, False, None, ops.EPI_F32)
    one = ops.norm_linear(W[:N].contiguous(), x[M - 1:].contiguous(), g, None, 1e-5, False, None,
                          ops.EPI_F32)
    assert torch.equal(one[0], full[M - 1])


@pytest.mark.parametrize("Hq,Hkv,D,H,rope", [(4, 2, 64, 256, True), (32, 8, 128, 4096, True),
                                              (12, 12, 64, 768, False), (14, 2, 64, 896, True)])
def test_fused_qkv_rope_append_matches_unfused(Hq, Hkv, D, H, rope):
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(Hq * D)
    B, q_len = 2, 3
    M = B * q_len
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda() if rope else None
    x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    W = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.tensor([[3, 1, 4], [0, 5, 2]], dtype=torch.int32, device="cuda")
    seq_lens = torch.tensor([17, 30], dtype=torch.int32, device="cuda")
    caches = []
    for fused in (False, True):
        kc = torch.zeros(6, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        if fused:
            q = ops.norm_qkv_rope(W, x, g, None, 1e-5, False, bias, cs, kc, vc, bt, seq_lens, -1, B,
                                  q_len, Hq, Hkv, D)
        else:
            qkv = ops.linear(W, ops.rmsnorm(x, g, 1e-5), bias, None, ops.EPI_STORE)
            ops.rope_append(qkv, cs, kc, vc, bt, seq_lens, -1, B, q_len, Hq, Hkv, D)
            q = qkv[:, :Hq * D].contiguous()
        caches.append((q.float(), kc.float(), vc.float()))
    for a, b in zip(*caches):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2)
    assert caches[1][1].abs().sum() > 0 and caches[1][2].abs().sum() > 0


# ------------------------------------------------------------------ fp8 weights
def _lib_error():
    from vllm_omni_draft_amd._lib import SdError
    return SdError


def _fp8_pow2(W):
    """(fp8 values, power-of-two f32 scale per row, the bf16 matrix they stand for exactly)."""
    Wf = W.float()
    e = torch.ceil(torch.log2(Wf.abs().amax(dim=1).clamp_min(1e-12) / 448.0))
    scale = torch.exp2(e).to(torch.float32)
    q = (Wf / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    exact = (q.float() * scale[:, None]).to(torch.bfloat16)
    assert torch.equal(exact.float(), q.float() * scale[:, None])
    return q, scale, exact


@pytest.mark.parametrize("M", [1, 6, 16])
@pytest.mark.parametrize("N,K", [(64, 512), (130, 768), (1024, 4096), (257, 896), (96, 14336)])
def test_fp8_linear_is_the_bf16_kernel_on_dequantised_weights(M, N, K):
    """With power-of-two scales the dequantised matrix is exact in bf16 and scaling commutes
    with every rounding, so the fp8 kernel must reproduce the bf16 kernel bit for bit, in
    every epilogue: this pins the byte layout, the in-register conversion and the scale
    indexing (plain, gated) at once."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(M * 11 + N)
    x = torch.randn(M, K, generator=gen, device="cuda").to(torch.bfloat16)
    bias = (torch.randn(2 * N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=gen, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W2[3] *= 37.0          # rows of very different magnitude: per-row scales matter
    W2[N + 1] *= 0.01
    q2, s2, exact2 = _fp8_pow2(W2)
    q, sc, exact = q2[:N].contiguous(), s2[:N].contiguous(), exact2[:N].contiguous()
    Wt = ops.tile_weights(q, ops.TILE_PLAIN, scale=sc)
    assert Wt.data.dtype == torch.uint8 and Wt.nbytes() < 0.6 * ops.tile_weights(exact).nbytes()
    for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_RELU, ops.EPI_F32, ops.EPI_GELU_TANH):
        for b in (None, bias[:N].contiguous()):
            r = res if epi == ops.EPI_RESIDUAL else None
            assert torch.equal(ops.linear(Wt, x, b, r, epi), ops.linear(exact, x, b, r, epi)), epi
    Wg = ops.tile_weights(q2, ops.TILE_GATED, scale=s2)
    for b in (None, bias):
        assert torch.equal(ops.linear(Wg, x, b, None, ops.EPI_SILU_MUL),
                           ops.linear(exact2, x, b, None, ops.EPI_SILU_MUL))


def test_fp8_linear_general_scales_and_quantisation_error():
    """Arbitrary per-row scales: the f32 output is exactly (sum over the fp8 values) * scale,
    and the quantised layer stays within e4m3's relative step of the bf16 layer."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 6, 512, 4096
    x = torch.randn(M, K, generator=gen, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, generator=gen, device="cuda") * 0.02).to(torch.bfloat16)
    q, scale = ops.quantize_fp8(W)
    assert q.dtype == torch.float8_e4m3fn and scale.dtype == torch.float32
    got = ops.linear(ops.tile_weights(q, scale=scale), x, None, None, ops.EPI_F32)
    unscaled = ops.linear(q.to(torch.bfloat16), x, None, None, ops.EPI_F32)
    assert torch.equal(got, unscaled * scale[None, :])
    ref = x.double() @ W.double().t()
    # 3 mantissa bits: relative steps of 6-12%, i.e. ~2-3% rms error per weight, which carries
    # over to a sum of K independent terms
    err = (got.double() - ref).abs()
    assert err.max().item() < 0.05 * ref.abs().max().item()
    assert err.mean().item() < 0.05 * ref.abs().mean().item()


@pytest.mark.parametrize("ln", [False, True])
def test_fp8_fused_norm_and_qkv_rope_match_bf16_kernels(ln):
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(17)
    Hq, Hkv, D, H, B, q_len = 4, 2, 64, 256, 2, 3
    M = B * q_len
    x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    nb = (0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16) if ln else None
    W2 = (torch.randn(2 * 192, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    q2, s2, exact2 = _fp8_pow2(W2)
    for epi, tm in ((ops.EPI_F32, ops.TILE_PLAIN), (ops.EPI_STORE, ops.TILE_PLAIN),
                    (ops.EPI_SILU_MUL, ops.TILE_GATED), (ops.EPI_RELU, ops.TILE_PLAIN)):
        got = ops.norm_linear(ops.tile_weights(q2, tm, scale=s2), x, g, nb, 1e-5, ln, None, epi)
        ref = ops.norm_linear(ops.tile_weights(exact2, tm), x, g, nb, 1e-5, ln, None, epi)
        assert torch.equal(got, ref), epi
    # QKV + rotary + cache append: the scale is indexed by source row through the rotary tiling
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda()
    Wq = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    Wq[5] *= 20.0
    qq, sq, exq = _fp8_pow2(Wq)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.tensor([[3, 1, 4], [0, 5, 2]], dtype=torch.int32, device="cuda")
    seq_lens = torch.tensor([17, 30], dtype=torch.int32, device="cuda")
    outs = []
    for Wt in (ops.tile_weights(qq, ops.TILE_ROTARY, D, scale=sq), ops.tile_weights(exq, ops.TILE_ROTARY, D)):
        kc = torch.zeros(6, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        qo = ops.norm_qkv_rope(Wt, x, g, nb, 1e-5, ln, bias, cs, kc, vc, bt, seq_lens, -1, B, q_len,
                               Hq, Hkv, D)
        outs.append((qo, kc, vc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][1].abs().sum() > 0


def test_fp8_rejects_unsupported_uses():
    ops = _ops()
    from vllm_omni_draft_amd._lib import SdError
    W = torch.randn(64, 96, device="cuda").to(torch.bfloat16)        # K % 64 != 0
    q, s = ops.quantize_fp8(W)
    with pytest.raises(SdError):
        ops.tile_weights(q, scale=s)
    q, s = ops.quantize_fp8(torch.randn(64, 128, device="cuda").to(torch.bfloat16))
    with pytest.raises(SdError):
        ops.tile_weights(q)                                           # no scale


@pytest.mark.parametrize("M", [17, 50, 192])
@pytest.mark.parametrize("N,K", [(64, 512), (130, 768), (1024, 4096), (96, 14336), (20480, 256)])
def test_fp8_wide_linear_is_the_bf16_wide_kernel_on_dequantised_weights(M, N, K):
    """The 17..256-row family with fp8 tiles: bit-identical to its bf16 form on the dequantised
    matrix (power-of-two scales), through split-K partial sums and the direct epilogue alike."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(M * 13 + N)
    x = torch.randn(M, K, generator=gen, device="cuda").to(torch.bfloat16)
    bias = (torch.randn(2 * N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=gen, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W2[3] *= 37.0
    W2[N + 1] *= 0.01
    q2, s2, exact2 = _fp8_pow2(W2)
    q, sc, exact = q2[:N].contiguous(), s2[:N].contiguous(), exact2[:N].contiguous()
    Wt, Wb = ops.tile_weights(q, ops.TILE_PLAIN, scale=sc), ops.tile_weights(exact)
    for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_RELU, ops.EPI_F32, ops.EPI_GELU_TANH):
        for b in (None, bias[:N].contiguous()):
            r = res if epi == ops.EPI_RESIDUAL else None
            assert torch.equal(ops.linear_wide(Wt, x, b, r, epi), ops.linear_wide(Wb, x, b, r, epi)), epi
    Wg, Wgb = ops.tile_weights(q2, ops.TILE_GATED, scale=s2), ops.tile_weights(exact2, ops.TILE_GATED)
    for b in (None, bias):
        assert torch.equal(ops.linear_wide(Wg, x, b, None, ops.EPI_SILU_MUL),
                           ops.linear_wide(Wgb, x, b, None, ops.EPI_SILU_MUL))


def test_fp8_wide_qkv_rope_matches_bf16_wide():
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(23)
    Hq, Hkv, D, H, B, q_len = 4, 2, 64, 256, 6, 5
    M = B * q_len
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda()
    x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    Wq = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    Wq[5] *= 20.0
    qq, sq, exq = _fp8_pow2(Wq)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.arange(B * 3, dtype=torch.int32, device="cuda").view(B, 3)
    seq_lens = torch.tensor([17, 30, 3, 9, 21, 12], dtype=torch.int32, device="cuda")
    outs = []
    for Wt in (ops.tile_weights(qq, ops.TILE_ROTARY, D, scale=sq), ops.tile_weights(exq, ops.TILE_ROTARY, D)):
        kc = torch.zeros(B * 3, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        qo = ops.qkv_rope_wide(Wt, x, bias, cs, kc, vc, bt, seq_lens, -1, B, q_len, Hq, Hkv, D)
        outs.append((qo, kc, vc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][1].abs().sum() > 0


# ------------------------------------------------------------------ fp8 activations (W8A8)
def _quantise_rows(x, scales=None):
    """The oracle's activation quantisation: per row, q = e4m3(clamp(x * (448 / amax), +-448))
    (round to nearest even), scale = amax / 448.  scales ([2, M], the library's): quantise with
    those factors (the two divisions may differ from torch's in the last bit)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    ok = amax > 0
    inv = torch.where(ok, 448.0 / amax, torch.ones_like(amax))
    scale = torch.where(ok, amax / 448.0, torch.ones_like(amax))
    if scales is not None:
        scale, inv = scales[0], scales[1]
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale, inv


def test_row_scales_fp8_match_the_oracle():
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(70, 1024, generator=gen, device="cuda") * 3.0).to(torch.bfloat16)
    x[5] = 0
    x[9] *= 1e-3
    got = ops.row_scales_fp8(x)
    _, scale, inv = _quantise_rows(x)
    assert torch.allclose(got[0], scale, rtol=3e-7, atol=0) and torch.allclose(got[1], inv, rtol=3e-7, atol=0)
    assert got[0][5].item() == 1.0 and got[1][5].item() == 1.0


@pytest.mark.parametrize("M", [17, 50, 128, 192, 256])
@pytest.mark.parametrize("N,K", [(64, 512), (130, 768), (1024, 4096), (96, 14336), (20480, 256), (2048, 896)])
def test_w8a8_wide_linear_against_the_quantising_oracle(M, N, K):
    """fp8 weights x fp8 activations on the fp8 matrix instruction, against the f64 sum of the same
    quantised operands.  Stated tolerance: the instruction aligns the 128 products of a step to
    the largest one and drops what falls below about 2^-21 of it (measured: exact on small
    integers, up to 27 on sums whose largest product is 1.2e5 over K = 512), so an output may be
    off by K * 2^-20 * (largest |x_q| of the row) * (largest |w_q| of the weight row) in units of
    the two scales; the quantisation itself (3 mantissa bits) is two orders above that.  Through
    split-K partial sums and the direct epilogue, the epilogues the path has, biases, ragged N."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=gen, device="cuda") * 2.0).to(torch.bfloat16)
    x[1] *= 30.0
    x[2] = 0
    bias = (torch.randn(2 * N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=gen, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W2[3] *= 37.0
    q2, s2 = ops.quantize_fp8(W2)
    xq, xs, _ = _quantise_rows(x, ops.row_scales_fp8(x))
    ref2 = (xq.double() @ q2.double().T) * xs.double()[:, None] * s2.double()[None, :]
    tol = (K * 2.0 ** -20 * xq.double().abs().amax(dim=1)[:, None] * q2.double().abs().amax(dim=1)[None, :]
           * xs.double()[:, None] * s2.double()[None, :]) + 1e-30
    Wt = ops.tile_weights(q2[:N].contiguous(), ops.TILE_PLAIN, scale=s2[:N].contiguous())
    ref, tol2, tol = ref2[:, :N], tol, tol[:, :N]
    y = ops.linear_wide(Wt, x, None, None, ops.EPI_F32, fp8_activations=True)
    assert ((y.double() - ref).abs() <= tol).all(), ((y.double() - ref).abs() / tol).max().item()
    yb = ops.linear_wide(Wt, x, bias[:N].contiguous(), None, ops.EPI_F32, fp8_activations=True)
    assert ((yb.double() - ref - bias[:N].double()).abs() <= tol + 1e-6 * (1 + ref.abs())).all()
    # bf16 epilogues: the f32 result rounded as the bf16-activation kernels round theirs
    st = ops.linear_wide(Wt, x, None, None, ops.EPI_STORE, fp8_activations=True)
    assert torch.equal(st, y.to(torch.bfloat16))
    rs = ops.linear_wide(Wt, x, None, res, ops.EPI_RESIDUAL, fp8_activations=True)
    assert torch.equal(rs, (y.to(torch.bfloat16).float() + res.float()).to(torch.bfloat16))
    Wg = ops.tile_weights(q2, ops.TILE_GATED, scale=s2)
    g = ops.linear_wide(Wg, x, None, None, ops.EPI_SILU_MUL, fp8_activations=True).float()
    gr, ur = ref2[:, :N].float(), ref2[:, N:].float()
    want = torch.nn.functional.silu(gr) * ur
    err = (g - want).abs()      # bf16 roundings of gate, silu, up and the product, plus the sums' tolerance
    bound = 0.02 * want.abs() + 2.0 * (tol2[:, :N] * ur.abs() + tol2[:, N:] * gr.abs()).float() + 1e-6
    assert (err <= bound).all(), (err / bound).max().item()
    # a row's result does not depend on what shares the pass
    if M >= 50:
        part = ops.linear_wide(Wt, x[:33].contiguous(), None, None, ops.EPI_F32, fp8_activations=True)
        assert torch.equal(part, y[:33])
    with pytest.raises(_lib_error()):
        ops.linear_wide(Wt, x, None, None, ops.EPI_RELU, fp8_activations=True)


def test_w8a8_wide_needs_fp8_weights_and_whole_chunks():
    ops = _ops()
    x = torch.randn(32, 192, device="cuda").to(torch.bfloat16)
    W = torch.randn(64, 192, device="cuda").to(torch.bfloat16)
    with pytest.raises(_lib_error()):
        ops.linear_wide(ops.tile_weights(W), x, fp8_activations=True)          # bf16 weights
    q, s = ops.quantize_fp8(W)
    with pytest.raises(_lib_error()):
        ops.linear_wide(ops.tile_weights(q, scale=s), x, fp8_activations=True)   # K % 128 != 0


def test_w8a8_wide_qkv_rope_against_the_bf16_activation_form():
    """The rotary QKV launch with fp8 activations: q and the cache entries stay within the
    quantisation error of the bf16-activation launch on the same fp8 weights (one part in 16
    of a row's largest value, e4m3's 3-bit mantissa), and every cache slot that one writes the
    other writes too."""
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(29)
    Hq, Hkv, D, H, B, q_len = 4, 2, 64, 256, 6, 5
    M = B * q_len
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda()
    x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    Wq = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    qq, sq = ops.quantize_fp8(Wq)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.arange(B * 3, dtype=torch.int32, device="cuda").view(B, 3)
    seq_lens = torch.tensor([17, 30, 3, 9, 21, 12], dtype=torch.int32, device="cuda")
    Wt = ops.tile_weights(qq, ops.TILE_ROTARY, D, scale=sq)
    outs = []
    for a8 in (True, False):
        kc = torch.zeros(B * 3, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        qo = ops.qkv_rope_wide(Wt, x, bias, cs, kc, vc, bt, seq_lens, -1, B, q_len, Hq, Hkv, D,
                               fp8_activations=a8)
        outs.append((qo, kc, vc))
    for a, b in zip(*outs):
        assert ((a != 0) == (b != 0)).float().mean().item() > 0.999
        assert (a.float() - b.float()).abs().max().item() < b.float().abs().max().item() / 16


# ------------------------------------------------------------------ LM head with the argmax inside
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("M,N,K", [(1, 128256, 2048), (6, 128256, 4096), (2, 300, 256), (16, 1000, 512),
                                   (3, 8192, 896)])
def test_lm_head_argmax_epilogue_equals_argmax_of_its_logits(M, N, K, ln):
    """sd_norm_linear_argmax: every row's first argmax from the slice maxima the head's workgroups
    leave equals the first argmax of the f32 logits the F32 epilogue writes (same kernel, same
    bits), also with exact ties (duplicated weight rows: the lowest column wins) and fp8 weights."""
    ops = _ops()
    if ln and K > 4096:
        pytest.skip("LayerNorm rows beyond 4096 columns are not fused")
    gen = torch.Generator(device="cuda").manual_seed(M * 31 + N)
    x = torch.randn(M, K, generator=gen, device="cuda").to(torch.bfloat16)
    g = (1 + 0.1 * torch.randn(K, generator=gen, device="cuda")).to(torch.bfloat16)
    b = (0.1 * torch.randn(K, generator=gen, device="cuda")).to(torch.bfloat16) if ln else None
    W = (torch.randn(N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for variant in ("bf16", "ties", "fp8"):
        if variant == "ties":      # the row maxima appear two and three times
            logits = ops.norm_linear(ops.tile_weights(W), x, g, b, 1e-5, ln, None, ops.EPI_F32)
            top = logits.argmax(-1)
            for m in range(M):
                t = int(top[m])
                for dst in ((t + 97) % N, (t + N // 2) % N):
                    W[dst] = W[t]
        if variant == "fp8":
            q, sc = ops.quantize_fp8(W)
            Wt = ops.tile_weights(q, ops.TILE_PLAIN, scale=sc)
        else:
            Wt = ops.tile_weights(W)
        logits = ops.norm_linear(Wt, x, g, b, 1e-5, ln, None, ops.EPI_F32)
        pv, pi = ops.norm_linear_argmax(Wt, x, g, b, 1e-5, ln)
        got = ops.argmax_from_partials(pv, pi, M)
        want = ops.argmax_rows(logits)
        assert got.tolist() == want.tolist(), variant
        assert torch.equal(pv.max(dim=1).values, logits.max(dim=1).values)
        if variant == "ties":
            mx = logits.max(dim=1, keepdim=True).values
            first = (logits == mx).float().argmax(dim=1)
            assert got.tolist() == first.tolist()
            assert ((logits == mx).sum(1) >= 2).float().mean().item() >= 0.5   # (rows may overwrite each other's copies)


# ------------------------------------------------------------------ 17..256 token rows
@pytest.mark.parametrize("M", [17, 32, 50, 192, 256])
@pytest.mark.parametrize("N,K", [(64, 512), (130, 768), (1024, 4096), (257, 896), (96, 14336)])
def test_wide_linear_epilogues(M, N, K):
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=gen, device="cuda")).to(torch.bfloat16)
    bias = (torch.randn(2 * N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, generator=gen, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(2 * N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W = W2[:N].contiguous()
    for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_RELU, ops.EPI_F32, ops.EPI_GELU_TANH):
        for b in (None, bias[:N].contiguous()):
            got = ops.linear_wide(W, x, b, res if epi == ops.EPI_RESIDUAL else None, epi).float()
            ref = _ref_linear(W, x, b, res, epi, ops)
            torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    for b in (None, bias):
        got = ops.linear_wide(W2, x, b, None, ops.EPI_SILU_MUL).float()
        ref = _ref_linear(W2, x, b, None, ops.EPI_SILU_MUL, ops)
        torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)


def test_wide_linear_f32_accuracy_and_row_independence():
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(0)
    W = (torch.randn(512, 4096, generator=gen, device="cuda") * 0.02).to(torch.bfloat16)
    x = torch.randn(192, 4096, generator=gen, device="cuda").to(torch.bfloat16)
    Wt = ops.tile_weights(W)
    full = ops.linear_wide(Wt, x, None, None, ops.EPI_F32)
    ref = x.double() @ W.double().t()
    assert (full.double() - ref).abs().max().item() < 2e-5 * 4096 ** 0.5
    # a row's bits do not depend on how many rows share the pass (the K split is a
    # function of the matrix alone)
    for M, m0 in ((1, 0), (6, 100), (32, 160), (100, 92)):
        part = ops.linear_wide(Wt, x[m0:m0 + M].contiguous(), None, None, ops.EPI_F32)
        assert torch.equal(part, full[m0:m0 + M]), (M, m0)


@pytest.mark.parametrize("K", [512, 1792, 640])
def test_many_row_passes_on_four_waves_keep_every_bit(K):
    """Matrices of 1024..2047 weight tiles run their passes of 65..256 rows on the four-wave
    kernel (loads spread through the MFMA stream) when K is a multiple of 256 (K = 640: the
    fallback to the eight-wave kernel).  Its rows equal, bit for bit, the same rows computed in
    passes of at most 64 rows, which the eight-wave kernel serves: every epilogue it has, bf16
    and fp8 weights, row counts that are no multiple of 16, a last tile that is partly outside
    N; and the reference within the tolerance of the other wide tests."""
    ops = _ops()
    gen = torch.Generator(device="cuda").manual_seed(K)
    N, Ng = 16424, 8204                 # 1027 plain tiles; 1026 gated tiles (the last one half used)
    W = (torch.randn(N, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    W2 = (torch.randn(2 * Ng, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias_all = (torch.randn(N, generator=gen, device="cuda") * 0.1).to(torch.bfloat16)
    bias = bias_all[:2 * Ng].contiguous()
    x_all = torch.randn(256, K, generator=gen, device="cuda").to(torch.bfloat16)
    res_all = torch.randn(256, N, generator=gen, device="cuda").to(torch.bfloat16)
    q, sc = ops.quantize_fp8(W)
    q2, sc2 = ops.quantize_fp8(W2)
    forms = [(ops.tile_weights(W), ops.tile_weights(W2, ops.TILE_GATED), True),
             (ops.tile_weights(q, ops.TILE_PLAIN, scale=sc), ops.tile_weights(q2, ops.TILE_GATED, scale=sc2), False)]
    for M in (65, 100, 192, 250, 256):
        x, res = x_all[:M].contiguous(), res_all[:M].contiguous()
        for Wt, Wg, check_ref in forms:
            for epi in (ops.EPI_STORE, ops.EPI_RESIDUAL, ops.EPI_F32):
                b = bias_all if epi != ops.EPI_F32 else None
                r = res if epi == ops.EPI_RESIDUAL else None
                got = ops.linear_wide(Wt, x, b, r, epi)
                few = torch.cat([ops.linear_wide(Wt, x[i:i + 64].contiguous(), b,
                                                 None if r is None else r[i:i + 64].contiguous(), epi)
                                 for i in range(0, M, 64)])
                assert torch.equal(got, few), (M, epi)
                if check_ref and M in (65, 256):
                    torch.testing.assert_close(got.float(), _ref_linear(W, x, b, res, epi, ops), rtol=2e-2, atol=2e-2)
            got = ops.linear_wide(Wg, x, bias, None, ops.EPI_SILU_MUL)
            few = torch.cat([ops.linear_wide(Wg, x[i:i + 64].contiguous(), bias, None, ops.EPI_SILU_MUL)
                             for i in range(0, M, 64)])
            assert torch.equal(got, few), M
            if check_ref and M in (65, 256):
                torch.testing.assert_close(got.float(), _ref_linear(W2, x, bias, None, ops.EPI_SILU_MUL, ops),
                                           rtol=3e-2, atol=3e-2)


def test_many_row_qkv_rope_on_four_waves_keeps_every_bit():
    """The fused QKV + rotary + cache-append epilogue on the four-wave kernel (a projection of
    1024 tiles, 96 rows): queries and both caches equal the same tokens run as two passes of 48
    rows, which the eight-wave kernel serves."""
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    Hq, Hkv, D, H = 96, 16, 128, 512
    gen = torch.Generator(device="cuda").manual_seed(5)
    B, q_len = 16, 6
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda()
    x = torch.randn(B * q_len, H, generator=gen, device="cuda").to(torch.bfloat16)
    W = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    Wt = ops.tile_weights(W, ops.TILE_ROTARY, D)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.randperm(B * 3, generator=gen, device="cuda").to(torch.int32).view(B, 3).contiguous()
    seq_lens = torch.randint(1, 30, (B,), generator=gen, device="cuda").to(torch.int32)
    outs = []
    for parts in (1, 2):
        kc = torch.zeros(B * 3, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        nb = B // parts
        q = torch.cat([ops.qkv_rope_wide(Wt, x[b0 * q_len:(b0 + nb) * q_len].contiguous(), bias, cs, kc, vc,
                                         bt[b0:b0 + nb].contiguous(), seq_lens[b0:b0 + nb].contiguous(), -1, nb,
                                         q_len, Hq, Hkv, D) for b0 in range(0, B, nb)])
        outs.append((q, kc, vc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][1].abs().sum() > 0 and outs[0][2].abs().sum() > 0


@pytest.mark.parametrize("Hq,Hkv,D,H,rope", [(32, 8, 128, 4096, True), (12, 12, 64, 768, False),
                                              (14, 2, 64, 896, True)])
def test_wide_qkv_rope_append_matches_narrow_ops(Hq, Hkv, D, H, rope):
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(Hq * D)
    B, q_len = 8, 6
    M = B * q_len
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda() if rope else None
    x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    W = (torch.randn((Hq + 2 * Hkv) * D, H, generator=gen, device="cuda") * H ** -0.5).to(torch.bfloat16)
    bias = (0.1 * torch.randn((Hq + 2 * Hkv) * D, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.randperm(B * 3, generator=gen, device="cuda").to(torch.int32).view(B, 3).contiguous()
    seq_lens = torch.randint(1, 30, (B,), generator=gen, device="cuda").to(torch.int32)
    caches = []
    for wide in (False, True):
        kc = torch.zeros(B * 3, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        if wide:
            q = ops.qkv_rope_wide(ops.tile_weights(W, ops.TILE_ROTARY, D), x, bias, cs, kc, vc, bt,
                                  seq_lens, -1, B, q_len, Hq, Hkv, D)
        else:
            qkv = torch.cat([ops.linear(W, x[i:i + 16].contiguous(), bias, None, ops.EPI_STORE)
                             for i in range(0, M, 16)])
            ops.rope_append(qkv, cs, kc, vc, bt, seq_lens, -1, B, q_len, Hq, Hkv, D)
            q = qkv[:, :Hq * D].contiguous()
        caches.append((q.float(), kc.float(), vc.float()))
    for a, b in zip(*caches):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2)
    assert caches[1][1].abs().sum() > 0 and caches[1][2].abs().sum() > 0
