"""Per-kernel time of the target's linears (Llama-3-8B shapes) against the row count M = 6..16."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_linear import case  # noqa: E402

for M in (6, 8, 9, 12, 16):
    print(f"--- target shapes (Llama-3-8B), M={M}")
    case("qkv (norm)", 6144, 4096, M, "norm")
    case("o (+res)", 4096, 4096, M, "res")
    case("gate/up (norm, silu)", 14336, 4096, M, "gated", L=24)
    case("down (+res)", 4096, 14336, M, "res", L=24)
