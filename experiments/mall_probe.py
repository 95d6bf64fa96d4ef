"""How fast a weight-streaming linear launch runs when its matrix was read shortly before:
cold (after a 1 GiB flush), right after itself, after a dense read by another kernel, after a
sparse touch (one dword per 128-byte line), and after a touch followed by an unrelated 48 MB
stream that turns over the XCDs' L2 (what is left then sits in the memory-side cache)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from vllm_omni_draft_amd import ops
lib = ctypes.CDLL('experiments/libmall.so')
for f in (lib.sd_prefetch, lib.sd_touch):
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = 'cuda'
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
other = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
def t_us(fn, pre):
    ts = []
    for _ in range(9):
        pre(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts)//2]
st = torch.cuda.current_stream().cuda_stream
shapes = [(2048, 2048, 1), (16384, 2048, 1), (2048, 8192, 1), (3072, 2048, 1),
          (4096, 4096, 6), (6144, 4096, 6), (4096, 14336, 6), (28672, 4096, 6)]
for (N, K, M) in shapes:
    W = ops.tile_weights((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.linear(W, x, None, None, ops.EPI_STORE, out=out)
    p, nbytes = W.data.data_ptr(), W.nbytes()
    cold = t_us(run, lambda: flush.zero_())
    warm = t_us(run, lambda: run())
    pf = t_us(run, lambda: (flush.zero_(), lib.sd_prefetch(p, nbytes, 2048, st)))
    tc = t_us(run, lambda: (flush.zero_(), lib.sd_touch(p, nbytes, 1024, st)))
    mall = t_us(run, lambda: (flush.zero_(), lib.sd_touch(p, nbytes, 1024, st),
                              lib.sd_prefetch(other.data_ptr(), other.numel(), 2048, st)))
    pfk = t_us(lambda: lib.sd_prefetch(p, nbytes, 2048, st), lambda: flush.zero_())
    tk = t_us(lambda: lib.sd_touch(p, nbytes, 1024, st), lambda: flush.zero_())
    f = lambda t: f"{t:.1f}us ({nbytes / t * 1e-6:.2f} TB/s)"
    print(f"N={N} K={K} M={M} {nbytes / 1e6:.1f} MB: cold {f(cold)} | after itself {f(warm)} | "
          f"after dense read {f(pf)} | after touch {f(tc)} | after touch + 48 MB of other reads {f(mall)} | "
          f"dense read cold {f(pfk)} | touch cold {f(tk)}", flush=True)
