import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import mesh_mnist as mm
from surfacenetworks_amd.graphs import BatchAhead
dev = torch.device("cuda"); B = 512
rng = np.random.default_rng(2)
ds = mm.MeshDigits(B, seed=2, device=dev, fixed_vertices=150, model="dir")
model = mm.DirModel().to(dev).train(); opt = mm.make_optimizer(model); ids = np.arange(B)
for _ in range(3): mm.train_step(model, opt, ds.sample_batch(B, rng, ids=ids))
g = mm.graphed_train_step(model, opt, ds.sample_batch(B, rng, ids=ids))
def timed(step, n=60, w=10):
    for _ in range(w): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ahead = BatchAhead(lambda: ds.sample_batch(B, rng, ids=ids), dev)
t0 = time.perf_counter()
for _ in range(50): ds.sample_batch(B, rng, ids=ids)
print("host time of sample_batch alone: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3)); torch.cuda.synchronize()
for rep in range(4):
    print("serial  %.3f ms" % timed(lambda: g(ds.sample_batch(B, rng, ids=ids))))
    print("ahead   %.3f ms" % timed(lambda: g(ahead.get())))
