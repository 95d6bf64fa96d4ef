"""Loading local Hugging Face checkpoints (config.json + *.safetensors shards) into the tiled
layouts: the loaded model must be, tensor for tensor, the one built from the same state dict in
memory (bf16 and fp8), for Llama-style, OPT ("decoder." key prefix) and Qwen2-VL checkpoints
(language model under "model.language_model.", vision tower beside it)."""
import json
import os

import pytest
import torch

from oracle.models import OracleConfig, random_state

pytestmark = pytest.mark.gpu


def _write(dirname, hf_cfg, state, n_files=2):
    from safetensors.torch import save_file
    os.makedirs(dirname, exist_ok=True)
    with open(os.path.join(dirname, "config.json"), "w") as f:
        json.dump(hf_cfg, f)
    keys = sorted(state)
    for i in range(n_files):
        part = {k: state[k].contiguous() for k in keys[i::n_files]}
        save_file(part, os.path.join(dirname, f"model-{i + 1:05d}-of-{n_files:05d}.safetensors"))


def _same(a, b):
    ta, tb = a.data if hasattr(a, "mode") else a, b.data if hasattr(b, "mode") else b
    if ta is None or tb is None:
        return ta is None and tb is None
    ok = torch.equal(ta, tb)
    sa, sb = getattr(a, "scale", None), getattr(b, "scale", None)
    return ok and ((sa is None and sb is None) or torch.equal(sa, sb))


def _assert_same_weights(w1, w2):
    for name in ("embed", "final_norm_w", "final_norm_b", "lm_head", "pos_table", "rope_cs"):
        assert _same(getattr(w1, name), getattr(w2, name)), name
    assert w1.embed_tiled == w2.embed_tiled and len(w1.layers) == len(w2.layers)
    for l1, l2 in zip(w1.layers, w2.layers):
        for name in vars(l1):
            assert _same(getattr(l1, name), getattr(l2, name)), name


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("kind", ["llama", "qwen2_tied", "opt", "qwen2_vl"])
def test_safetensors_dir_loads_the_same_model(tmp_path, kind, fp8):
    from vllm_omni_draft_amd import config, weights
    if kind == "opt":
        oc = OracleConfig(arch="opt", n_layers=2, hidden=192, n_heads=3, n_kv_heads=3, head_dim=64,
                          ffn=384, vocab=272, max_pos=128, tie_embeddings=True)
        hf = {"model_type": "opt", "num_hidden_layers": 2, "hidden_size": 192, "num_attention_heads": 3,
              "ffn_dim": 384, "vocab_size": 272, "max_position_embeddings": 128,
              "tie_word_embeddings": True}
        state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=3).items()}
        on_disk = {k[len("model."):]: v for k, v in state.items()}        # "decoder.…" as OPT checkpoints
    else:
        tied = kind == "qwen2_tied"
        oc = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                          ffn=512, vocab=320, max_pos=128, norm_eps=1e-6, rope_theta=1e6,
                          tie_embeddings=tied)
        bias = kind != "llama"
        hf = {"model_type": {"llama": "llama", "qwen2_tied": "qwen2", "qwen2_vl": "qwen2_vl"}[kind],
              "num_hidden_layers": 2, "hidden_size": 256, "num_attention_heads": 4,
              "num_key_value_heads": 2, "intermediate_size": 512, "vocab_size": 320,
              "max_position_embeddings": 128, "rms_norm_eps": 1e-6, "rope_theta": 1e6,
              "tie_word_embeddings": tied}
        state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=3, qkv_bias=bias).items()}
        on_disk = dict(state)
        if kind == "qwen2_vl":
            hf["rope_scaling"] = {"type": "mrope", "mrope_section": [8, 12, 12]}
            on_disk = {("model.language_model." + k[len("model."):] if k.startswith("model.") else k): v
                       for k, v in state.items()}
            on_disk["model.visual.patch_embed.proj.weight"] = torch.zeros(4, 3, 2, 2, 2, dtype=torch.bfloat16)
    d = str(tmp_path / kind)
    _write(d, hf, on_disk)
    cfg = config.load_hf_config(d)
    assert cfg.arch == oc.arch and cfg.n_layers == 2 and cfg.hidden == oc.hidden
    if kind == "qwen2_vl":
        assert cfg.mrope_section == (8, 12, 12) and cfg.qkv_bias
    loaded = weights.from_safetensors_dir(cfg, d, "cuda", fp8=fp8)
    direct = weights.from_state_dict(cfg, state, "cuda", fp8=fp8)
    assert loaded.fp8 == fp8
    _assert_same_weights(loaded, direct)


def test_safetensors_dir_errors(tmp_path):
    from vllm_omni_draft_amd import config, weights
    with pytest.raises(FileNotFoundError):
        weights.from_safetensors_dir(config.create_model_config(None), str(tmp_path), "cuda")
    with pytest.raises(ValueError):
        config.from_hf_config({"model_type": "mamba"})
