"""Verify-attention time against the number of query tokens (same KV bytes): is the slowdown from
1 to 6 query tokens gradual (data-dependent matrix-pipe power / clocks) or a step?
    python experiments/bench_attention_qlen.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attention import run  # noqa: E402

if __name__ == "__main__":
    for q_len in (1, 2, 3, 4, 6, 8):
        run(32, 4096, 32, 8, 128, q_len, 4096)
