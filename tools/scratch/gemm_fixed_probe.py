"""Fixed cost of the Linear forward / input-gradient launches: time at 1, 2, 4, 8 tiles per workgroup (256 workgroups)."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from surfacenetworks_amd import kernels, _lib

def timed(fn, n=20):
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
for name in ("fwd K=256 elu-only", "fwd K=128 elu-only", "fwd K=256 res+elu+y", "dgrad+elu C=256", "dgrad+elu C=128"):
    out = []
    for tiles in (1, 2, 4, 8, 16, 5000):       # 5000: above kSmallRows -> the two-pass prologue of the large launches
        rows = 32 * 256 * tiles if tiles < 5000 else 140000
        if name.startswith("fwd"):
            K = 256 if "256" in name else 128
            x = torch.randn(rows, K, device=dev); W = torch.randn(128, K, device=dev) * 0.1; b = torch.randn(128, device=dev)
            cat = torch.empty(rows, 256, device=dev)
            part = kernels.new_elu_stats_part(rows, dev)
            r = torch.randn(rows, 128, device=dev) if "res" in name else None
            wy = "res" in name
            t = timed(lambda: kernels.linear_fwd(x, W, b, r, cat[:, :128], wy, part))
        else:
            C = 256 if "256" in name else 128
            dy = torch.randn(rows, 128, device=dev); W = torch.randn(128, C, device=dev) * 0.1; x = torch.randn(rows, C, device=dev)
            mu = torch.randn(C, device=dev); B = torch.randn(C, device=dev); Cc = torch.randn(C, device=dev)
            t = timed(lambda: kernels.linear_dgrad_elu(dy, W, x, mu, B, Cc, None))
        out.append(f"{rows}: {t:.1f}")
    print(name, " | ".join(out), flush=True)
