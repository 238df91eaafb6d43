"""The models' last layer (128 -> 120 outputs): what would a 128-float row pitch of its output / of the loss gradient buy?
Forward, weight gradient and input gradient at 322 624 rows with ld = 120 and ld = 128 (timing slots)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from surfacenetworks_amd import kernels, _lib
from surfacenetworks_amd.kernels import _p, _ld, _stream

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
rows, K, J = 322624, 128, 120
x = torch.randn(rows, K, device=dev); W = torch.randn(J, K, device=dev) * 0.1; b = torch.randn(J, device=dev)
mu = x.mean(0); B = torch.randn(K, device=dev); Cc = torch.randn(K, device=dev)
for ld in (120, 128):
    ybuf = torch.empty(rows, ld, device=dev); y = ybuf[:, :J]
    dybuf = torch.randn(rows, ld, device=dev); dy = dybuf[:, :J]
    t_f = timed(lambda: _lib.call("sn_linear_fwd_tiles_f32", _p(x), _ld(x), _p(W), _ld(W), _p(b), None, 0, _p(y), ld, None, 0, rows, K, J, None, None, _stream()))
    t_w = timed(lambda: kernels.wgrad(dy, x, mu, want_colsum=True))
    t_d = timed(lambda: kernels.linear_dgrad_eluseg(dy, W, x, mu, B, Cc, None, 0))
    print(f"ld {ld}: forward {t_f:.1f} us, weight gradient {t_w:.1f} us, input gradient through the activation {t_d:.1f} us")
