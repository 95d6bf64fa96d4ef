"""MI355X-native speculative-decoding draft/verify engine.

Python host code over a C-ABI HIP library (libsd_amd.so, include/sd_amd.h).
There is no CPU or PyTorch fallback: importing is cheap, but every operation
needs the built library and a GPU and raises otherwise.
"""
__version__ = "0.1.0"
