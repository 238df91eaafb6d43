"""Weight-gradient kernels against fp64 on awkward ranges (run with SN_WGRAD_VARIANT=1|2|3)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels  # noqa: E402
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(5)
worst = 0.0
def check(name, dy, x, cen=None):
    global worst
    G, s = kernels.wgrad(dev(dy), dev(x), None if cen is None else dev(cen), want_colsum=True)
    xc = x.astype(np.float64) - (0 if cen is None else cen.astype(np.float64))
    ref = dy.astype(np.float64).T @ xc
    scale = np.abs(dy.astype(np.float64)).T @ np.abs(xc) + 1e-300
    err = float((np.abs(G.cpu().numpy().astype(np.float64) - ref) / scale).max())
    es = float(np.abs(s.cpu().numpy() - dy.astype(np.float64).sum(0)).max() / (np.abs(dy).sum(0).max() + 1e-300))
    fin = bool(torch.isfinite(G).all())
    worst = max(worst, err)
    print(f"{name:44s} rows {dy.shape[0]:7d} J {dy.shape[1]:3d} C {x.shape[1]:3d}  err/sum|dy||x| {err:.2e}  colsum {es:.1e}  finite {fin}", flush=True)
for C in (256, 128):
    for rows in (1, 7, 33, 500, 4097, 70001):
        check("normal", rng.standard_normal((rows, 128)).astype(np.float32), rng.standard_normal((rows, C)).astype(np.float32))
    rows = 40000
    dy = rng.standard_normal((rows, 128)).astype(np.float32); x = rng.standard_normal((rows, C)).astype(np.float32)
    check("centre", dy, x + 50.0, (np.full(C, 50.0) + rng.standard_normal(C)).astype(np.float32))
    check("columns over 2^40", dy * np.exp2(rng.integers(-20, 20, 128)).astype(np.float32), x * np.exp2(rng.integers(-20, 20, C)).astype(np.float32))
    ramp = np.exp2(np.linspace(-30, 30, rows)).astype(np.float32)[:, None]
    check("magnitude grows along the rows (2^-30..2^30)", dy * ramp, x * ramp)
    check("magnitude falls along the rows", dy * ramp[::-1], x * ramp[::-1])
    z = dy.copy(); z[: rows // 2] = 0; zx = x.copy(); zx[: rows // 3] = 0
    check("zero rows first", z, zx)
    sp = dy * (rng.random((rows, 128)) < 0.01)
    check("sparse dy (1 %)", sp.astype(np.float32), x)
    big = dy.copy(); big[rows - 5, 7] = 1e30; big[3, 9] = 1e-30
    check("one huge / one tiny entry", big, x)
    check("log-normal (sigma 3)", (dy * np.exp(3 * rng.standard_normal((rows, 128)))).astype(np.float32), (x * np.exp(3 * rng.standard_normal((rows, C)))).astype(np.float32))
    check("J = 120", dy[:, :120].copy(), x)
    check("tiny 1e-20 x huge 1e15", (dy * 1e-20).astype(np.float32), (x * 1e15).astype(np.float32))
print("worst", worst)
