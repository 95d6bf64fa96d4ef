"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/sd_amd.h declares (no compute calls: there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sd_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = _declared()
    for need in ("sd_rejection_sample", "sd_greedy_verify", "sd_paged_attention", "sd_rope_append",
                 "sd_linear", "sd_model_forward", "sd_graph_launch", "sd_last_error"):
        assert need in names


def test_library_exports_every_declared_symbol():
    from vllm_omni_draft_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        subprocess.run(["make", "-C", ROOT, "-j8"], check=True)
    lib = _lib.load()
    declared = _declared()
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sd_version() >= 100


def test_library_contains_gfx950_code_object():
    from vllm_omni_draft_amd import _lib
    out = subprocess.run(["strings", "-a", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vllm_omni_draft_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "lib_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SdError):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch
    from vllm_omni_draft_amd import _lib, ops
    with pytest.raises(_lib.SdError):
        ops.greedy_verify(torch.zeros(1, 2, dtype=torch.int64), torch.zeros(1, 1, dtype=torch.int64))
