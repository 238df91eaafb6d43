# Host-side cost of a planned block call WITHOUT a GPU: CPU tensors, Plan.run replaced by a no-op (what remains is the Python around
# sn_plan_run: keys, lookups, arenas, views, autograd).  python tools/scratch/plan_host_profile.py [--profile]
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import blocks, functional as snF, kernels, mesh_ops, plans
from surfacenetworks_amd import utils_pt as U
from surfacenetworks_amd.operators import SparseOperator
plans._ALLOW_CPU = True
plans.Plan.run = lambda self, big, small, ext: None
plans.reset(); snF.set_dirac_format("csr")
rng = np.random.default_rng(0)
V, F = mesh_ops.grid_cloth(12, 9, rng); ops = mesh_ops.mesh_operators(V, F)
nV, nF, Cc = V.shape[0], F.shape[0], 128
def op_of(A):
    o = SparseOperator.from_scipy(A, "cpu"); At = A.T.tocsr(); At.sort_indices(); o._t = SparseOperator.from_scipy(At, "cpu"); return o
Di, DiA, L = op_of(ops["Di"]), op_of(ops["DiA"]), op_of(ops["L"])
for o in (L, L._t): o.format = "csr"
mods = [(U.DirResNet2(Cc), U.AvgResNet2(Cc)) for _ in range(4)]
v = torch.randn(1, nV, Cc, requires_grad=True); mask = torch.ones(1, nV, 1)
def step():
    x, f = v, None
    for i, (d, a) in enumerate(mods):
        x, f = d(Di, DiA, x, f, f_out_needed=False, num_faces=nF, avg_next=True)
        x = a(None, mask, x)
    x.sum().backward(); kernels.clear_absmax()
for _ in range(5): step()
N = 300
dts = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(N): step()
    dts.append((time.perf_counter() - t0) / N)
dt = min(dts)
print(f"{dt*1e6:.0f} us per step of 8 blocks fwd+bwd = {dt*1e6/16:.1f} us per block direction", plans.stats()["dirac_fwd"])
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N): step()
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(28)
