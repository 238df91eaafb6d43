"""Prototype measurement: LDS-tiled Laplacian SpMM against the library's RB4 kernel on the config-5 batch."""
import ctypes as C, os, subprocess, sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels, mesh_ops
from surfacenetworks_amd.operators import OperatorPool

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "liblaptile.so")
lib = C.CDLL(so)
dev = "cuda"
rng = np.random.default_rng(5)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
if which == "c5":
    vs = rng.integers(1000, 20001, size=128)
    grids = [(int(np.sqrt(v)), int(v) // int(np.sqrt(v))) for v in vs]
else:
    grids = [(71, 71)] * 64
Ls = []
for n, m in grids:
    V, F = mesh_ops.grid_cloth(n, m, rng)
    Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
pool = OperatorPool(Ls, dev)
op = pool.assemble(np.arange(len(Ls)))
M, K = op.shape
rowptr, colind, vals = op.rowptr, op.colind, op.vals
N = 128
x = torch.randn(K, N, device=dev)
y0 = torch.empty(M, N, device=dev)
r = op.rb4()

def timeit(f, iters=50, warm=10):
    for _ in range(warm): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

ms0 = timeit(lambda: kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, y0))
ab = op.nnz * 8 + (M + 1) * 4 + K * N * 4 + M * N * 4
print(f"{which}: M={M} nnz={op.nnz} rb4 {ms0:.4f} ms frac={ab / ms0 / 1e-3 / 8e12:.3f}")
rp = rowptr.cpu().numpy().astype(np.int64); ci = colind.cpu().numpy().astype(np.int64)
rows = np.repeat(np.arange(M), np.diff(rp))
for TR in (64, 128, 256):
    tid = rows // TR
    ntiles = (M + TR - 1) // TR
    key = tid * K + ci
    uk, inv = np.unique(key, return_inverse=True)
    ut = uk // K
    uptr = np.zeros(ntiles + 1, np.int64); np.add.at(uptr, ut + 1, 1); uptr = np.cumsum(uptr)
    lidx = (inv - uptr[tid]).astype(np.uint16)
    ucols = (uk % K).astype(np.int32)
    max_u = int(np.diff(uptr).max())
    d_uptr = torch.from_numpy(uptr.astype(np.int32)).to(dev); d_uc = torch.from_numpy(ucols).to(dev)
    d_li = torch.from_numpy(lidx.view(np.int16)).to(dev)
    for CH in (32, 64, 128):
        shm = max_u * (CH + 4) * 4
        if shm > 150 * 1024 or (CH == 128 and TR == 256): continue
        y = torch.zeros(M, N, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            rc = lib.lt_spmm(C.c_void_p(d_uptr.data_ptr()), C.c_void_p(d_uc.data_ptr()), C.c_void_p(rowptr.data_ptr()),
                             C.c_void_p(d_li.data_ptr()), C.c_void_p(vals.data_ptr()), C.c_void_p(x.data_ptr()),
                             C.c_void_p(y.data_ptr()), M, ntiles, N, CH, TR, max_u, C.c_void_p(st))
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        ok = torch.allclose(y, y0, rtol=1e-5, atol=1e-5)
        ms = timeit(run)
        print(f"  TR={TR} CH={CH} U/row={len(uk) / M:.2f} max_u={max_u} lds={shm // 1024} KB: {ms:.4f} ms frac={ab / ms / 1e-3 / 8e12:.3f} ok={ok} equal={torch.equal(y, y0)}")
