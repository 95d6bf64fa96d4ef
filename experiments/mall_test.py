import ctypes, sys, torch
sys.path.insert(0, '.')
from vllm_omni_draft_amd import ops
lib = ctypes.CDLL('experiments/libmall.so')
lib.sd_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = 'cuda'
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def t_us(fn, pre):
    ts = []
    for _ in range(7):
        pre(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts)//2]
st = torch.cuda.current_stream().cuda_stream
for (N, K, M) in [(2048, 2048, 1), (4096, 4096, 6), (8192, 4096, 1), (4096, 14336, 6), (14336, 4096, 6), (28672, 4096, 6)]:
    W = ops.tile_weights((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.linear(W, x, None, None, ops.EPI_STORE, out=out)
    nbytes = W.nbytes()
    cold = t_us(run, lambda: flush.zero_())
    warm = t_us(run, lambda: run())
    pf = t_us(run, lambda: (flush.zero_(), lib.sd_prefetch(W.data.data_ptr(), nbytes, 2048, st)))
    pfk = t_us(lambda: lib.sd_prefetch(W.data.data_ptr(), nbytes, 2048, st), lambda: flush.zero_())
    print(f"N={N} K={K} M={M} {nbytes/1e6:.1f}MB cold={cold:.1f}us ({nbytes/cold*1e-6:.2f}TB/s) warm-after-self={warm:.1f}us ({nbytes/warm*1e-6:.2f}) after-prefetch={pf:.1f}us ({nbytes/pf*1e-6:.2f}) prefetch-kernel-cold={pfk:.1f}us ({nbytes/pfk*1e-6:.2f})")
