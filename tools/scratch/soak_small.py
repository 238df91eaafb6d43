"""allocator steady state of the small-batch pipelines: 1000 Mesh-MNIST steps with BatchAhead, 500 FAUST pair steps with the
prefetched target; reserved / allocated memory and hipMalloc counts before and after"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import mesh_mnist as mm, dense_correspondence as dc
from surfacenetworks_amd.graphs import BatchAhead
dev = torch.device("cuda")
def stats(tag):
    s = torch.cuda.memory_stats()
    print(f"{tag}: reserved {s['reserved_bytes.all.current'] / 2**20:.0f} MiB, allocated {s['allocated_bytes.all.current'] / 2**20:.0f} MiB, "
          f"device allocs {s['num_device_alloc']}, frees {s['num_device_free']}", flush=True)
B = 512; rng = np.random.default_rng(2)
ds = mm.MeshDigits(B, seed=2, device=dev, fixed_vertices=150, model="dir")
model = mm.DirModel().to(dev).train(); opt = mm.make_optimizer(model); ids = np.arange(B)
g = mm.graphed_train_step(model, opt, ds.sample_batch(B, rng, ids=ids))
ahead = BatchAhead(lambda: ds.sample_batch(B, rng, ids=ids), dev)
for _ in range(50): g(ahead.get())
torch.cuda.synchronize(); stats("mnist after 50")
t0 = time.perf_counter()
for _ in range(1000): loss = g(ahead.get())
torch.cuda.synchronize(); print(f"  1000 steps: {(time.perf_counter() - t0):.2f} s, loss {loss.item():.4f}"); stats("mnist after 1050")
for rep in range(4):
    for _ in range(1000): loss = g(ahead.get())
    torch.cuda.synchronize(); stats(f"mnist after {2050 + 1000 * rep}")
del g, ahead, model, opt, ds; torch.cuda.empty_cache()
ds = dc.TorusBodies(4, device=dev); model = dc.SiameseModel("lap", 15).to(dev).train(); opt = dc.make_optimizer(model)
g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))
for k in range(30): g(dc.PairBatch(ds, k % 4, (k + 1) % 4))
torch.cuda.synchronize(); stats("faust after 30")
t0 = time.perf_counter()
for k in range(500): loss = g(dc.PairBatch(ds, k % 4, (k + 1) % 4))
torch.cuda.synchronize(); print(f"  500 steps: {(time.perf_counter() - t0):.2f} s, loss {loss.item():.4f}"); stats("faust after 530")
