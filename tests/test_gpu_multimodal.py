"""Multimodal prompts on the GPU: the multimodal rotary embedding in the fused QKV kernels of
both families, input rows given by the caller, and whole engines on a prompt with images
(logits against the oracle; greedy speculative decoding == greedy autoregressive decoding)."""
import pytest
import torch

from oracle.models import OracleConfig, OracleLM, random_state

pytestmark = pytest.mark.gpu
IMG = 90

VL = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=128,
                  ffn=320, vocab=320, norm_eps=1e-6, rope_theta=1e6, max_pos=256,
                  mrope_section=(16, 24, 24))
VL64 = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64,
                    ffn=256, vocab=320, norm_eps=1e-6, rope_theta=1e4, max_pos=256,
                    mrope_section=(6, 13, 13))
DRAFT = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                     ffn=256, vocab=320, max_pos=256)


def _bf(state):
    return {k: v.to(torch.bfloat16).float() for k, v in state.items()}


def _product(oc, state, qkv_bias=False):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        norm_eps=oc.norm_eps, rope_theta=oc.rope_theta, max_pos=oc.max_pos,
        tie_embeddings=oc.tie_embeddings, qkv_bias=qkv_bias, mrope_section=oc.mrope_section)
    return weights.from_state_dict(cfg, state, "cuda")


def _prompt(segments):
    from vllm_omni_draft_amd import multimodal
    ids, grids, nxt = [], [], 1
    for s in segments:
        if isinstance(s, tuple):
            ids += [IMG] * multimodal.image_token_count(s)
            grids.append(s)
        else:
            ids += [(nxt + j) % 80 for j in range(s)]
            nxt += s
    return ids, grids


def _image_rows(n, hidden, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, hidden, generator=g) * 0.05).to(torch.bfloat16)


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("D,section", [(128, (16, 24, 24)), (64, (6, 13, 13))])
def test_fused_qkv_rotary_positions(wide, D, section):
    """Explicit positions equal to the cache positions, or a zero delta, change nothing; a
    delta d gives the q and k of a sequence whose cache position is d further on; per-component
    positions rotate every frequency by the position of its section."""
    from vllm_omni_draft_amd import config, ops
    Hq, Hkv, K, B, q_len = 4, 2, 256, 2, 3
    gen = torch.Generator().manual_seed(3)
    W = (torch.randn((Hq + 2 * Hkv) * D, K, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    x = torch.randn(B * q_len, K, generator=gen).to(torch.bfloat16).cuda()
    cs = config.rope_table(config.create_model_config(None, head_dim=D, max_pos=128), "cuda")
    Wt = ops.tile_weights(W, ops.TILE_ROTARY, D)
    n_pages = 24
    bt = torch.arange(B * 8, dtype=torch.int32, device="cuda").view(B, 8)

    def run(lens, **kw):
        kc = torch.zeros(n_pages, Hkv, 16 * D, dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
        if wide:
            q = ops.qkv_rope_wide(Wt, x, None, cs, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, **kw)
        else:
            q = ops.norm_qkv_rope(Wt, x, None, None, 0.0, False, None, cs, kc, vc, bt, sl, 0, B,
                                  q_len, Hq, Hkv, D, **kw)
        torch.cuda.synchronize()
        return q, kc, vc

    lens = [5, 20]
    base = run(lens)
    cache_pos = torch.tensor([l + t for l in lens for t in range(q_len)], dtype=torch.int32)
    same = run(lens, rope_pos=cache_pos.repeat(3, 1).contiguous().cuda(), mrope_section=section)
    zero = run(lens, rope_delta=torch.zeros(B, dtype=torch.int32, device="cuda"), mrope_section=section)
    for got in (same, zero):
        assert all(torch.equal(a, b) for a, b in zip(got, base))

    delta = [7, -4]
    moved = run(lens, rope_delta=torch.tensor(delta, dtype=torch.int32, device="cuda"),
                mrope_section=section)
    want = run([l + d for l, d in zip(lens, delta)])
    assert torch.equal(moved[0], want[0])          # q rotated as at the shifted position
    assert torch.equal(moved[2], base[2])          # v and the cache slots do not move
    # k: same values as the shifted run, found in the slots of the unshifted positions
    def k_rows(kc, positions):
        rows = []
        for b in range(B):
            for t in range(q_len):
                p = positions[b] + t
                page = kc[bt[b, p // 16].item()]                   # [Hkv, 16*D] as [D/8][16][8]
                rows.append(page.view(Hkv, D // 8, 16, 8)[:, :, p % 16, :].reshape(Hkv, D))
        return torch.stack(rows)
    assert torch.equal(k_rows(moved[1], lens), k_rows(want[1], [l + d for l, d in zip(lens, delta)]))

    # per-component positions: compare with a float reference of the rotation
    pos3 = torch.stack([cache_pos + 1, cache_pos * 2, cache_pos + 9]).to(torch.int32)
    got_q = run(lens, rope_pos=pos3.contiguous().cuda(), mrope_section=section)[0].float().cpu()
    raw = (x.float().cpu() @ W.float().cpu().T).to(torch.bfloat16).float()[:, :Hq * D].view(-1, Hq, D)
    comp = torch.repeat_interleave(torch.arange(3), torch.tensor(section))
    per_freq = pos3.long()[comp].T                                        # [rows, D/2]
    table = cs.cpu()
    idx = torch.arange(D // 2)
    c, s = table[per_freq, idx][:, None, :], table[per_freq, idx + D // 2][:, None, :]
    lo, hi = raw[..., :D // 2], raw[..., D // 2:]
    ref = torch.cat([lo * c - hi * s, hi * c + lo * s], dim=-1).reshape(-1, Hq * D)
    torch.testing.assert_close(got_q, ref.to(torch.bfloat16).float(), rtol=0.02, atol=0.02)


def test_rotary_positions_are_validated():
    from vllm_omni_draft_amd import _lib, config, ops, weights
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    plain = OracleConfig(**{**vars(VL64), "mrope_section": None})
    w = _product(plain, _bf(random_state(plain, seed=1)))
    model = DecoderModel(w, 64, 128)
    kv = PagedKVCache(w.cfg, 8, 1, 64, "cuda")
    kv.ensure_capacity(0, 16)
    tok = torch.zeros(4, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.SdError):   # a model without sections takes no rotary positions
        model.forward(kv, tok, 0, 0, zero, 0, 1, 4, None, 64,
                      rope_pos=torch.zeros(3, 4, dtype=torch.int32, device="cuda"))
    w2 = _product(VL64, _bf(random_state(VL64, seed=1)))
    m2 = DecoderModel(w2, 64, 128)
    with pytest.raises(_lib.SdError):   # wrong shape
        m2.forward(kv, tok, 0, 0, zero, 0, 1, 4, None, 64,
                   rope_pos=torch.zeros(3, 5, dtype=torch.int32, device="cuda"))
    with pytest.raises(_lib.SdError):
        m2.forward(kv, tok, 0, 0, zero, 0, 1, 4, None, 64,
                   input_embeds=torch.zeros(4, 64, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(_lib.SdError):   # sections beyond D/2 at the kernel entry point
        x = torch.zeros(2, 128, dtype=torch.bfloat16, device="cuda")
        ops.norm_qkv_rope(w2.layers[0].qkv_w, x, None, None, 0.0, False, None, w2.rope_cs, kv.k[0], kv.v[0],
                          kv.block_tables, zero, 0, 1, 2, 4, 2, 64,
                          rope_delta=zero, mrope_section=(20, 20, 0))


@pytest.mark.parametrize("oc", [VL, VL64], ids=["d128", "d64"])
@pytest.mark.parametrize("wide", [False, True])
def test_forward_with_images_matches_oracle(oc, wide):
    """Prompt with two images through the native forward (chunks with explicit positions, then
    text rows positioned by the delta) against the oracle with bf16 activations."""
    from vllm_omni_draft_amd import multimodal
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    state = _bf(random_state(oc, seed=31, qkv_bias=True))
    w = _product(oc, state, qkv_bias=True)
    ids, grids = _prompt([3, (1, 4, 6), 4, (1, 2, 4), 5])
    n = len(ids)
    img = _image_rows(sum(t == IMG for t in ids), oc.hidden, 7)
    embeds, pos = multimodal.prepare_prompt(ids, w, img.cuda(), grids, image_token_id=IMG)
    _, delta = multimodal.mrope_positions(ids, IMG, grids)
    assert delta < 0
    tail = [11, 12, 13]
    model = DecoderModel(w, 128, 128, wide=wide)
    kv = PagedKVCache(w.cfg, 16, 1, 128, "cuda")
    kv.ensure_capacity(0, n + len(tail))
    tok = torch.tensor(ids + tail, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_dev = torch.tensor([delta], dtype=torch.int32, device="cuda")
    outs, c0 = [], 0
    for q in (8, 8, n - 16):
        lg = torch.empty(q, oc.vocab, dtype=torch.float32, device="cuda")
        model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128,
                      input_embeds=embeds[c0:c0 + q].contiguous(),
                      rope_pos=pos[:, c0:c0 + q].contiguous().cuda())
        outs.append(lg)
        c0 += q
    lg = torch.empty(len(tail), oc.vocab, dtype=torch.float32, device="cuda")
    model.forward(kv, tok, 0, n, zero, n, 1, len(tail), lg, 128, rope_delta=d_dev)
    outs.append(lg)
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu()

    lm = OracleLM(oc, state, bf16_acts=True)
    cache = lm.new_cache()
    tail_pos = torch.tensor([[n + delta + j for j in range(len(tail))]] * 3)
    ref = torch.cat([lm.forward(ids, 0, cache, rope_pos=pos, input_embeds=embeds.float().cpu()),
                     lm.forward(tail, n, cache, rope_pos=tail_pos)])
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    # the positions are in use: ordinary positions give other logits for the text after the images
    plain = lm.forward(ids, 0, lm.new_cache(), input_embeds=embeds.float().cpu())
    after = slice(8, n)   # rows behind the first image tokens
    assert (got[after] - plain[after]).abs().mean().item() > 2 * (got[after] - ref[after]).abs().mean().item()


def _engine(tw, dw, k, **kw):
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    return SpecEngine(tw, dw, EngineConfig(k=k, max_batch=kw.pop("max_batch", 2), max_seq_len=160,
                                           split_tokens=128, **kw))


def _generate(eng, requests, n_new):
    outs = [[eng.add_request(i, ids, input_embeds=e, rope_positions=p)]
            for i, (ids, e, p) in enumerate(requests)]
    while min(len(o) for o in outs) < n_new:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    return [o[:n_new] for o in outs]


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("rows", ["narrow", "wide"])
def test_engine_with_images_spec_equals_ar_and_oracle(use_graph, rows):
    """A batch of one prompt with images and one text prompt: the speculative stream equals
    the autoregressive one, with a text-only draft and with a draft that has rotary sections
    itself; the first tokens equal the oracle's greedy continuation."""
    from vllm_omni_draft_amd import multimodal
    state = _bf(random_state(VL, seed=41, qkv_bias=True))
    tw = _product(VL, state, qkv_bias=True)
    dw_text = _product(DRAFT, _bf(random_state(DRAFT, seed=42)))
    dw_vl = _product(VL64, _bf(random_state(VL64, seed=43, qkv_bias=True)), qkv_bias=True)
    ids, grids = _prompt([4, (1, 4, 4), 3, (1, 2, 6), 2])
    img = _image_rows(sum(t == IMG for t in ids), VL.hidden, 9)
    embeds, pos = multimodal.prepare_prompt(ids, tw, img.cuda(), grids, image_token_id=IMG)
    text = list(range(30, 47))
    requests = [(ids, embeds, pos), (text, None, None)]
    kw = dict(use_graph=use_graph)
    if rows == "wide":
        kw.update(rows="wide")
    ar = _generate(_engine(tw, None, 0, **kw), requests, 40)
    for dw in (dw_text, dw_vl):
        eng = _engine(tw, dw, 3, **kw)
        assert _generate(eng, requests, 40) == ar
        assert eng.metrics()["draft"] > 0
    # a text prompt in slot 1 is not disturbed by slot 0's delta, and the other way round
    solo = _generate(_engine(tw, None, 0, max_batch=1, **kw), [requests[0]], 40)
    assert solo[0] == ar[0]
    if rows == "narrow":
        lm = OracleLM(VL, state, bf16_acts=True)
        _, delta = multimodal.mrope_positions(ids, IMG, grids)
        cache = lm.new_cache()
        logits = lm.forward(ids, 0, cache, rope_pos=pos, input_embeds=embeds.float().cpu())
        want = []
        for j in range(6):
            nxt = int(logits[-1].argmax())
            want.append(nxt)
            p = len(ids) + j
            logits = lm.forward([nxt], p, cache, rope_pos=torch.full((3, 1), p + delta))
        got = solo[0][:6]
        # bf16 near-ties may flip a token: demand the common prefix to be long, not identity
        agree = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), 6)
        assert agree >= 2, (got, want)


def test_removing_a_request_moves_its_delta():
    from vllm_omni_draft_amd import multimodal
    state = _bf(random_state(VL, seed=41, qkv_bias=True))
    tw = _product(VL, state, qkv_bias=True)
    ids, grids = _prompt([4, (1, 4, 4), 3])
    img = _image_rows(sum(t == IMG for t in ids), VL.hidden, 9)
    embeds, pos = multimodal.prepare_prompt(ids, tw, img.cuda(), grids, image_token_id=IMG)
    text = list(range(30, 47))
    alone = _generate(_engine(tw, None, 0, max_batch=1), [(ids, embeds, pos)], 24)[0]
    eng = _engine(tw, None, 0, max_batch=2)
    first = [eng.add_request(0, text), eng.add_request(1, ids, input_embeds=embeds, rope_positions=pos)]
    out = [first[1]]
    out.extend(eng.step()[1])
    eng.remove_request(0)            # the image prompt moves from slot 1 to slot 0
    while len(out) < 24:
        out.extend(eng.step()[0])
    assert out[:24] == alone


def test_worker_runs_pixel_patches_through_the_vision_tower():
    """execute_model with pixel patches (vision tower on the GPU, bf16) gives the stream of the
    same prompt fed with the tower's rows, and speculative == autoregressive."""
    from vllm_omni_draft_amd import multimodal
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                         SpecDecodeConfig, SpecDecodeWorker)
    from vllm_omni_draft_amd.vision import VisionConfig, VisionTower, patchify
    vcfg = VisionConfig(depth=2, embed_dim=64, num_heads=4, mlp_ratio=2, hidden_size=VL.hidden,
                        patch_size=4)
    gen = torch.Generator().manual_seed(5)
    E, M = vcfg.embed_dim, vcfg.embed_dim * 4
    shapes = {"patch_embed.proj.weight": (E, 3, 2, 4, 4), "merger.ln_q.weight": (E,),
              "merger.ln_q.bias": (E,), "merger.mlp.0.weight": (M, M), "merger.mlp.0.bias": (M,),
              "merger.mlp.2.weight": (VL.hidden, M), "merger.mlp.2.bias": (VL.hidden,)}
    for i in range(vcfg.depth):
        b = f"blocks.{i}."
        shapes.update({b + "norm1.weight": (E,), b + "norm1.bias": (E,), b + "norm2.weight": (E,),
                       b + "norm2.bias": (E,), b + "attn.qkv.weight": (3 * E, E),
                       b + "attn.qkv.bias": (3 * E,), b + "attn.proj.weight": (E, E),
                       b + "attn.proj.bias": (E,), b + "mlp.fc1.weight": (2 * E, E),
                       b + "mlp.fc1.bias": (2 * E,), b + "mlp.fc2.weight": (E, 2 * E),
                       b + "mlp.fc2.bias": (E,)})
    vstate = {"visual." + k: torch.randn(*shp, generator=gen) * 0.08 + (1.0 if k.endswith("norm1.weight")
              or k.endswith("norm2.weight") or k.endswith("ln_q.weight") else 0.0) for k, shp in shapes.items()}
    tower = VisionTower(vcfg, vstate, device="cuda")
    image = torch.randn(3, 16, 16, generator=gen)
    patches, grid = patchify(image, vcfg)
    rows = tower.forward(patches, [grid])
    assert rows.shape == (multimodal.image_token_count(grid), VL.hidden) and rows.dtype == torch.bfloat16
    assert torch.isfinite(rows.float()).all() and rows.float().abs().max() > 0

    state = _bf(random_state(VL, seed=41, qkv_bias=True))
    tw = _product(VL, state, qkv_bias=True)
    dw = _product(DRAFT, _bf(random_state(DRAFT, seed=42)))
    ids = [3, 4] + [IMG] * multimodal.image_token_count(grid) + [5, 6, 7]

    def run(k, mm):
        w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=dw if k else None, num_speculative_tokens=k,
                                              max_num_seqs=1, max_model_len=120, split_tokens=128,
                                              vision=tower, image_token_id=IMG))
        w.init_device()
        w.load_model()
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(
            "r", True, ids, multi_modal_data=mm)])).sampled_token_ids[0]
        while len(out) < 30:
            out += w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", False)])).sampled_token_ids[0]
        return out[:30]

    via_pixels = run(3, {"pixel_values": patches, "image_grid_thw": [grid]})
    via_rows = run(3, {"image_embeds": rows, "image_grid_thw": [grid]})
    ar = run(0, {"pixel_values": patches, "image_grid_thw": [grid]})
    assert via_pixels == via_rows == ar
    text_only = run(0, None)
    assert text_only != ar          # the image rows and positions reach the model


# ---------------------------------------------------------------------------- the named multimodal configuration
# Qwen2-VL-7B's language model (rotary sections, QKV biases, 7 query heads per KV head) as the
# target and Qwen2.5-0.5B's shape as the text draft, both with fp8 weights, k = 5; the small pair
# keeps the oracle fast, the full-width pair (two layers each) has the real layer shapes.
CONFIG5 = {
    "small": (VL, OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                               ffn=256, vocab=320, norm_eps=1e-6, rope_theta=1e6, max_pos=256,
                               tie_embeddings=True)),
    "qwen2-vl-7b+0.5b": (
        OracleConfig(arch="llama", n_layers=2, hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128,
                     ffn=18944, vocab=152064, norm_eps=1e-6, rope_theta=1e6, max_pos=512,
                     mrope_section=(16, 24, 24)),
        OracleConfig(arch="llama", n_layers=2, hidden=896, n_heads=14, n_kv_heads=2, head_dim=64,
                     ffn=4864, vocab=152064, norm_eps=1e-6, rope_theta=1e6, max_pos=512,
                     tie_embeddings=True)),
}


def _product_fp8(oc, state):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        norm_eps=oc.norm_eps, rope_theta=oc.rope_theta, max_pos=oc.max_pos,
        tie_embeddings=oc.tie_embeddings, qkv_bias=True, mrope_section=oc.mrope_section)
    return weights.from_state_dict(cfg, state, "cuda", fp8=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("rows", ["narrow", "wide"])
@pytest.mark.parametrize("pair", list(CONFIG5))
def test_image_prompts_with_fp8_weights_and_five_proposals(pair, rows):
    """Everything the image-conditioned configuration combines, in one engine: a target with
    rotary sections and QKV biases whose prompt carries image rows, fp8 weights on target and
    draft, a text-only draft of Qwen2.5-0.5B's shape, k = 5, both kernel families.  (1) the fp8
    target's logits over the image prompt (explicit three-component positions) and the text
    behind it (cache position + delta) against the oracle run on the dequantised matrices;
    (2) greedy speculative == greedy autoregressive for a batch of an image prompt and a text
    prompt, and proposals are accepted and rejected along the way."""
    from tests.test_gpu_model import _fp8_dequantised
    from vllm_omni_draft_amd import multimodal
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    tcfg, dcfg = CONFIG5[pair]
    wide = rows == "wide"
    tstate = _bf(random_state(tcfg, seed=51, qkv_bias=True))
    tw = _product_fp8(tcfg, tstate)
    assert tw.fp8 and tw.cfg.mrope_section
    ids, grids = _prompt([3, (1, 4, 6), 4, (1, 2, 4), 5])
    n = len(ids)
    img = _image_rows(sum(t == IMG for t in ids), tcfg.hidden, 7)
    embeds, pos = multimodal.prepare_prompt(ids, tw, img.cuda(), grids, image_token_id=IMG)
    _, delta = multimodal.mrope_positions(ids, IMG, grids)
    tail = [11, 12, 13, 14, 15, 16]                      # one verify pass of k + 1 rows
    model = DecoderModel(tw, 128, 128, wide=wide)
    kv = PagedKVCache(tw.cfg, 16, 1, 128, "cuda")
    kv.ensure_capacity(0, n + len(tail))
    tok = torch.tensor(ids + tail, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_dev = torch.tensor([delta], dtype=torch.int32, device="cuda")
    G = tcfg.n_heads // tcfg.n_kv_heads
    step = min(8, 64 // G)                               # rows of one pass: q_len * G <= 64
    outs, c0 = [], 0
    while c0 < n:
        q = min(step, n - c0)
        lg = torch.empty(q, tcfg.vocab, dtype=torch.float32, device="cuda")
        model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128,
                      input_embeds=embeds[c0:c0 + q].contiguous(),
                      rope_pos=pos[:, c0:c0 + q].contiguous().cuda())
        outs.append(lg)
        c0 += q
    lg = torch.empty(len(tail), tcfg.vocab, dtype=torch.float32, device="cuda")
    model.forward(kv, tok, 0, n, zero, n, 1, len(tail), lg, 128, rope_delta=d_dev)
    outs.append(lg)
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu()
    lm = OracleLM(tcfg, _fp8_dequantised(tcfg, tstate), bf16_acts=True)
    cache = lm.new_cache()
    tail_pos = torch.tensor([[n + delta + j for j in range(len(tail))]] * 3)
    ref = torch.cat([lm.forward(ids, 0, cache, rope_pos=pos, input_embeds=embeds.float().cpu()),
                     lm.forward(tail, n, cache, rope_pos=tail_pos)])
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.8
    del lm, ref, cache, model, kv

    dw = _product_fp8(dcfg, _bf(random_state(dcfg, seed=52, qkv_bias=True)))
    assert dw.fp8 and not dw.cfg.mrope_section
    text = list(range(30, 47))
    requests = [(ids, embeds, pos), (text, None, None)]

    def engine(draft, k):
        return SpecEngine(tw, draft, EngineConfig(k=k, max_batch=2, max_seq_len=160, split_tokens=128,
                                                  rows=rows))
    ar = _generate(engine(None, 0), requests, 40)
    eng = engine(dw, 5)
    assert eng.wide == wide and eng.k == 5
    assert _generate(eng, requests, 40) == ar
    m = eng.metrics()
    assert m["draft"] > 0 and m["draft"] % 5 == 0 and m["emitted"] >= 2 * 39
