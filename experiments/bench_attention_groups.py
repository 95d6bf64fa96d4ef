"""Verify-attention rate for head groups that need the two-tile kernel (more than 32 query rows
per KV head): Qwen2.5-7B heads (7 per KV head) and one rank of Llama-3-70B/TP8 (8 per KV head).
    python experiments/bench_attention_groups.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attention import run  # noqa: E402

if __name__ == "__main__":
    for Hq, Hkv, q_len in ((28, 4, 6), (8, 1, 5), (32, 8, 6)):
        for split in (1024, 4096):
            run(32, 4096, Hq, Hkv, 128, q_len, split)
