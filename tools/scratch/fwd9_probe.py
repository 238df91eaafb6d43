"""The K = 128 -> 128 forward launch that writes only the activated copy (linear_fwd[9] of the bench line): what do the
statistics epilogue and the tile sums cost?  322 624 rows, timing slots."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from surfacenetworks_amd import kernels, _lib

def timed(fn, n=30):
    lib = _lib.load()
    for _ in range(3):
        fn()
    lib.sn_timing_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    cnt = int(lib.sn_timing_count())
    meta = np.zeros((max(cnt, 1), 5), np.int64); ms = np.zeros(max(cnt, 1), np.float64)
    written = ctypes.c_int64(0)
    _lib.call("sn_timing_drain", ms.ctypes.data, meta.ctypes.data, cnt, ctypes.addressof(written))
    lib.sn_timing_enable(0)
    big = ms[: written.value]
    big = big[meta[: written.value, 0] >= 0x100]
    return float(np.median(big)) * 1e3

dev = "cuda"
rows, K = 322624, 128
x = torch.randn(rows, K, device=dev); W = torch.randn(128, K, device=dev) * 0.1; b = torch.randn(128, device=dev)
cat = torch.empty(rows, 256, device=dev)
part = kernels.new_elu_stats_part(rows, dev)
tiles = kernels.new_tile_sums(rows, dev) if kernels.tile_sums_supported() else None
y = None
print("elu copy only, no statistics      %.1f us" % timed(lambda: kernels.linear_fwd(x, W, b, None, cat[:, :128], False, None)))
print("elu copy + statistics             %.1f us" % timed(lambda: kernels.linear_fwd(x, W, b, None, cat[:, :128], False, part)))
if tiles is not None:
    print("elu copy + statistics + tile sums %.1f us" % timed(lambda: kernels.linear_fwd(x, W, b, None, cat[:, :128], False, part, tiles)))
print("y only (no activation)            %.1f us" % timed(lambda: kernels.linear_fwd(x, W, b, None, None, True, None)))
out = torch.empty(rows, 128, device=dev)
print("copy 165 MB -> 165 MB (torch)      %.1f us" % (lambda: (lambda e0, e1: (e0.record(), [out.copy_(x) for _ in range(20)], e1.record(), torch.cuda.synchronize(), e0.elapsed_time(e1) * 50)[-1])(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))())
