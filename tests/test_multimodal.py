"""Multimodal prompts on the CPU: rotary positions against the `transformers` implementation in
this image (Qwen2VLModel.get_rope_index), the oracle's multimodal rotary embedding against
Qwen2VLTextModel on seeded random weights, and the host-side merging of image embeddings."""
import types

import pytest
import torch

from oracle.models import OracleConfig, OracleLM, random_state
from vllm_omni_draft_amd import multimodal

IMG = 90  # placeholder token id of the tiny vocabularies used here


def _prompt(segments):
    """segments: ints (that many text tokens) or (t, h, w) grids -> (token ids, grids)."""
    ids, grids, nxt = [], [], 1
    for s in segments:
        if isinstance(s, tuple):
            ids += [IMG] * multimodal.image_token_count(s)
            grids.append(s)
        else:
            ids += [(nxt + j) % 80 for j in range(s)]
            nxt += s
    return ids, grids


@pytest.mark.parametrize("segments", [
    [5], [3, (1, 4, 6), 4], [(1, 2, 2), 1], [2, (1, 8, 4), 3, (1, 4, 4), 6], [1, (2, 4, 4), 2]])
def test_mrope_positions_match_transformers(segments):
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel
    ids, grids = _prompt(segments)
    pos, delta = multimodal.mrope_positions(ids, IMG, grids)
    fake = types.SimpleNamespace(
        config=types.SimpleNamespace(vision_config=types.SimpleNamespace(spatial_merge_size=2)))
    fake.get_vision_position_ids = lambda *a, **k: Qwen2VLModel.get_vision_position_ids(fake, *a, **k)
    input_ids = torch.tensor([ids])
    # an image with t frames is fed as image type here as well (one grid per run of placeholders)
    types_ = torch.tensor([[1 if t == IMG else 0 for t in ids]], dtype=torch.int32)
    ref_pos, ref_delta = Qwen2VLModel.get_rope_index(
        fake, input_ids, types_, image_grid_thw=torch.tensor(grids) if grids else None)
    assert torch.equal(pos.long(), ref_pos[:, 0])
    assert delta == int(ref_delta[0, 0])
    if not grids:
        assert delta == 0 and torch.equal(pos[0], torch.arange(len(ids), dtype=torch.int32))


def test_mrope_positions_reject_inconsistent_prompts():
    with pytest.raises(ValueError):
        multimodal.mrope_positions([1, IMG, IMG, 2], IMG, [(1, 4, 4)])      # too few placeholders
    with pytest.raises(ValueError):
        multimodal.mrope_positions([1, 2], IMG, [(1, 2, 2)])                # grid without a run
    with pytest.raises(ValueError):
        multimodal.mrope_positions([IMG, 3, IMG], IMG, [(1, 2, 2)])         # run without a grid
    with pytest.raises(ValueError):
        ids, grids = _prompt([2, (5, 4, 4), 3])                              # a clip: more frames than
        multimodal.mrope_positions(ids, IMG, grids)                          # merged rows / columns
    with pytest.raises(ValueError):
        multimodal.image_token_count((1, 3, 4))


def test_merge_image_embeds():
    ids = [1, IMG, IMG, 7, IMG]
    tok = torch.arange(5 * 4, dtype=torch.float32).view(5, 4)
    img = -torch.ones(3, 4) * torch.tensor([[1.], [2.], [3.]])
    out = multimodal.merge_image_embeds(tok, ids, IMG, img)
    assert torch.equal(out[[0, 3]], tok[[0, 3]])
    assert torch.equal(out[[1, 2, 4]], img)
    with pytest.raises(ValueError):
        multimodal.merge_image_embeds(tok, ids, IMG, img[:2])


def test_oracle_mrope_matches_transformers_text_model():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextModel
    cfg = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=32,
                       ffn=192, vocab=96, norm_eps=1e-6, rope_theta=10000.0, max_pos=128,
                       mrope_section=(4, 6, 6))
    state = random_state(cfg, seed=5, qkv_bias=True)
    hc = Qwen2VLTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn,
                           num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                           num_key_value_heads=cfg.n_kv_heads, max_position_embeddings=cfg.max_pos,
                           rms_norm_eps=cfg.norm_eps,
                           rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta,
                                            "mrope_section": [4, 6, 6]},
                           attn_implementation="eager")
    hf = Qwen2VLTextModel(hc)
    body = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    missing, unexpected = hf.load_state_dict(body, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    hf.eval()
    ids, grids = _prompt([3, (1, 4, 6), 4, (1, 2, 4), 5])
    pos, delta = multimodal.mrope_positions(ids, IMG, grids)
    n = len(ids)
    gen = torch.Generator().manual_seed(2)
    embeds = state["model.embed_tokens.weight"][torch.tensor(ids)].clone()
    mask = torch.tensor(ids) == IMG
    embeds[mask] = torch.randn(int(mask.sum()), cfg.hidden, generator=gen) * 0.05
    # the prompt, then two text tokens whose rotary positions are cache position + delta
    tail = [11, 12]
    tail_pos = torch.tensor([[n + delta, n + 1 + delta]] * 3)
    all_embeds = torch.cat([embeds, state["model.embed_tokens.weight"][torch.tensor(tail)]])
    all_pos = torch.cat([pos.long(), tail_pos], dim=1)
    with torch.no_grad():
        hidden = hf(inputs_embeds=all_embeds[None], position_ids=all_pos[:, None, :]).last_hidden_state[0]
    ref = hidden @ state["lm_head.weight"].T
    lm = OracleLM(cfg, state)
    cache = lm.new_cache()
    got = torch.cat([
        lm.forward(ids[:9], 0, cache, rope_pos=pos[:, :9], input_embeds=embeds[:9]),
        lm.forward(ids[9:], 9, cache, rope_pos=pos[:, 9:], input_embeds=embeds[9:]),
        lm.forward(tail, n, cache, rope_pos=tail_pos)])
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
    # the positions matter: ordinary rotary positions give other logits
    plain = OracleLM(cfg, state)
    other = plain.forward(ids, 0, plain.new_cache(), input_embeds=embeds)
    assert (other - got[:n]).abs().max() > 1e-3


def test_config_reads_mrope_sections_and_checkpoint_names():
    from vllm_omni_draft_amd import config, weights
    hf = {"model_type": "qwen2_vl", "text_config": {
        "num_hidden_layers": 2, "hidden_size": 256, "num_attention_heads": 2, "num_key_value_heads": 2,
        "intermediate_size": 512, "vocab_size": 512,
        "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}}}
    c = config.from_hf_config(hf)
    assert c.mrope_section == (16, 24, 24) and c.qkv_bias and c.head_dim == 128
    config.rope_table(c)   # "mrope" scaling keeps the plain frequencies
    with pytest.raises(ValueError):
        config.create_model_config(c, mrope_section=(16, 24, 20))
    with pytest.raises(ValueError):
        config.create_model_config(None, arch="opt", mrope_section=(8, 12, 12))
    assert config.preset("qwen2-vl-7b").mrope_section == (16, 24, 24)
    st = {"model.language_model.layers.0.mlp.up_proj.weight": torch.zeros(1),
          "model.language_model.norm.weight": torch.zeros(1), "model.visual.blocks.0.x": torch.zeros(1),
          "visual.merger.y": torch.zeros(1), "lm_head.weight": torch.zeros(1)}
    assert sorted(weights.language_model_state(st)) == [
        "lm_head.weight", "model.layers.0.mlp.up_proj.weight", "model.norm.weight"]


def _tiny_vision():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import (Qwen2VisionTransformerPretrainedModel,
                                                                  Qwen2VLVisionConfig)
    from vllm_omni_draft_amd.vision import VisionConfig
    hc = Qwen2VLVisionConfig(depth=2, embed_dim=64, hidden_size=96, num_heads=4, mlp_ratio=2,
                             in_channels=3, patch_size=4, spatial_merge_size=2, temporal_patch_size=2,
                             hidden_act="quick_gelu", attn_implementation="eager")
    torch.manual_seed(3)
    hf = Qwen2VisionTransformerPretrainedModel(hc).eval()
    for p in hf.parameters():          # default init leaves LayerNorms trivial: perturb everything
        p.data.add_(torch.randn_like(p) * 0.05)
    cfg = VisionConfig(depth=2, embed_dim=64, num_heads=4, mlp_ratio=2, hidden_size=96, patch_size=4)
    return hf, cfg


def test_vision_tower_matches_transformers():
    from vllm_omni_draft_amd.vision import VisionTower, patchify
    hf, cfg = _tiny_vision()
    state = {"visual." + k: v for k, v in hf.state_dict().items()}
    tower = VisionTower(cfg, state, device="cpu", dtype=torch.float32)
    gen = torch.Generator().manual_seed(1)
    images = [torch.randn(3, 16, 24, generator=gen), torch.randn(3, 8, 8, generator=gen)]
    parts = [patchify(im, cfg) for im in images]
    patches = torch.cat([p for p, _ in parts])
    grids = [g for _, g in parts]
    assert grids == [(1, 4, 6), (1, 2, 2)]
    got = tower.forward(patches, grids)
    with torch.no_grad():
        ref = hf(patches, grid_thw=torch.tensor(grids)).pooler_output
    assert got.shape == (sum(multimodal.image_token_count(g) for g in grids), 96)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    # a two-frame grid (video): attention stays inside a frame
    vid = torch.randn(2 * 16, patches.shape[1], generator=gen)
    got = tower.forward(vid, [(2, 4, 4)])
    with torch.no_grad():
        ref = hf(vid, grid_thw=torch.tensor([[2, 4, 4]])).pooler_output
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        tower.forward(patches[:-1], grids)
    with pytest.raises(ValueError):
        VisionTower(cfg, {"visual.patch_embed.proj.weight": torch.zeros(1)}, device="cpu")


def test_patchify_matches_the_image_processor_layout():
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from vllm_omni_draft_amd.vision import VisionConfig, patchify
    cfg = VisionConfig(patch_size=4)
    img = torch.arange(3 * 16 * 24, dtype=torch.float32).view(3, 16, 24)
    ref, gh, gw = Qwen2VLImageProcessorPil.patchify(None, img.numpy(), patch_size=4, merge_size=2,
                                                    temporal_patch_size=2)
    patches, grid = patchify(img, cfg)
    assert tuple(grid) == (1, gh, gw)
    assert torch.equal(patches, torch.from_numpy(ref.copy()))
    with pytest.raises(ValueError):
        patchify(torch.zeros(3, 12, 16), cfg)
