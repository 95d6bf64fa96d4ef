"""The oracle's model arithmetic against the `transformers` implementations in
this image (Llama, Llama-3 rope scaling, Qwen2 with q/k/v bias, OPT), on seeded
random weights built from locally constructed configs (no hub access)."""
import numpy as np
import pytest
import torch

from oracle.models import OracleConfig, OracleLM, random_state, rope_inv_freq


def _hf_llama(cfg: OracleConfig, qwen=False):
    if qwen:
        from transformers import Qwen2Config, Qwen2ForCausalLM
        c = Qwen2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn,
                        num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                        num_key_value_heads=cfg.n_kv_heads, max_position_embeddings=cfg.max_pos,
                        rms_norm_eps=cfg.norm_eps, rope_theta=cfg.rope_theta,
                        tie_word_embeddings=cfg.tie_embeddings, attn_implementation="eager")
        return Qwen2ForCausalLM(c)
    from transformers import LlamaConfig, LlamaForCausalLM
    kw = {}
    if cfg.rope_scaling:
        kw["rope_scaling"] = dict(cfg.rope_scaling)
    c = LlamaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn,
                    num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                    num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
                    max_position_embeddings=cfg.max_pos, rms_norm_eps=cfg.norm_eps,
                    rope_theta=cfg.rope_theta, tie_word_embeddings=cfg.tie_embeddings,
                    attention_bias=False, attn_implementation="eager", **kw)
    return LlamaForCausalLM(c)


def _load(model, state):
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "lm_head" in m for m in missing), missing
    return model.eval()


@pytest.mark.parametrize("variant", ["llama", "llama3_scaling", "qwen2", "tied"])
def test_llama_family_matches_transformers(variant):
    cfg = OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=32,
                       ffn=192, vocab=96, norm_eps=1e-5, rope_theta=10000.0, max_pos=128)
    if variant == "llama3_scaling":
        cfg.rope_scaling = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                            "high_freq_factor": 4.0, "original_max_position_embeddings": 32}
    if variant == "tied":
        cfg.tie_embeddings = True
    state = random_state(cfg, seed=3, qkv_bias=(variant == "qwen2"))
    hf = _load(_hf_llama(cfg, qwen=(variant == "qwen2")), state)
    toks = torch.randint(0, cfg.vocab, (1, 40), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(toks).logits[0]
    lm = OracleLM(cfg, state)
    cache = lm.new_cache()
    # feed in three uneven chunks to exercise the incremental cache
    got = torch.cat([lm.forward(toks[0, :7].tolist(), 0, cache),
                     lm.forward(toks[0, 7:8].tolist(), 7, cache),
                     lm.forward(toks[0, 8:].tolist(), 8, cache)])
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)


def test_llama3_inv_freq_matches_transformers():
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    sc = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 8192}
    cfg = OracleConfig(arch="llama", n_layers=1, hidden=2048, n_heads=32, n_kv_heads=8, head_dim=64,
                       ffn=64, vocab=8, rope_theta=500000.0, rope_scaling=sc, max_pos=16)
    hc = LlamaConfig(hidden_size=2048, num_attention_heads=32, head_dim=64, rope_theta=500000.0,
                     rope_scaling=dict(sc), max_position_embeddings=131072)
    ref, _ = ROPE_INIT_FUNCTIONS["llama3"](hc, "cpu")
    torch.testing.assert_close(rope_inv_freq(cfg), ref.float(), rtol=1e-6, atol=0)


def test_opt_matches_transformers():
    from transformers import OPTConfig, OPTForCausalLM
    cfg = OracleConfig(arch="opt", n_layers=2, hidden=96, n_heads=3, n_kv_heads=3, head_dim=32,
                       ffn=160, vocab=80, norm_eps=1e-5, max_pos=64, tie_embeddings=True)
    c = OPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, ffn_dim=cfg.ffn,
                  num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                  max_position_embeddings=cfg.max_pos, word_embed_proj_dim=cfg.hidden,
                  do_layer_norm_before=True, dropout=0.0, attn_implementation="eager")
    state = random_state(cfg, seed=9)
    hf = OPTForCausalLM(c)
    missing, unexpected = hf.load_state_dict(state, strict=False)
    assert not unexpected and all("lm_head" in m for m in missing), (missing, unexpected)
    hf.eval()
    toks = torch.randint(0, cfg.vocab, (1, 30), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = hf(toks).logits[0]
    lm = OracleLM(cfg, state)
    cache = lm.new_cache()
    got = torch.cat([lm.forward(toks[0, :11].tolist(), 0, cache),
                     lm.forward(toks[0, 11:].tolist(), 11, cache)])
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)


def test_product_rope_table_matches_oracle():
    from oracle.models import rope_table as oracle_table
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    sc = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 8192}
    for scaling in (None, sc):
        oc = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=2, head_dim=64,
                          ffn=64, vocab=8, rope_theta=500000.0, rope_scaling=scaling, max_pos=300)
        pc = create_model_config(None, head_dim=64, rope_theta=500000.0, rope_scaling=scaling,
                                 max_pos=300)
        torch.testing.assert_close(rope_table(pc), oracle_table(oc), rtol=1e-6, atol=1e-6)
