"""Step-time measurements of the other BASELINE configs (parity-test cases, not bench lines):
   python tools/train_bench.py mnist_dir|mnist_lap|faust_lap|arap_lap|arap_ragged [steps]
mnist_dir = config 2 (Mesh-MNIST Dirac, batch 512); mnist_lap = config 1 shape on the GPU; faust_lap = config 4 per-GPU
work (one pair of 6890-vertex bodies padded to 7000); arap_lap = the Laplacian variant of config 3."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import dp, arap, dense_correspondence as dc, mesh_mnist as mm  # noqa: E402


def timed(step, steps, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    what = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = "cuda"
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    if what in ("mnist_dir", "mnist_lap"):
        B = 512 if what == "mnist_dir" else 32
        ds = mm.MeshDigits(B, seed=2, device=dev, fixed_vertices=150 if what == "mnist_dir" else None,
                           model="dir" if what == "mnist_dir" else "lap")
        model = (mm.DirModel() if what == "mnist_dir" else mm.Model()).to(dev).train()
        opt = mm.make_optimizer(model)
        ids = np.arange(B)
        dt = timed(lambda: mm.train_step(model, opt, ds.sample_batch(B, rng, ids=ids)), steps)
        ex = ds.sample_batch(B, rng, ids=ids)
        try:
            g = mm.graphed_train_step(model, opt, ex)
            nb = ds.sample_batch(B, rng, ids=ids)
            if g.matches(nb):
                dtg = timed(lambda: g(ds.sample_batch(B, rng, ids=ids)), steps)
                print(f"{what}: hipGraph replay of fwd+loss+bwd: {dtg * 1e3:.2f} ms/step, {B / dtg:.0f} meshes/s")
            else:
                print(f"{what}: batches have varying signatures (ragged meshes): no graph replay")
        except Exception as exc:  # noqa: BLE001
            print(f"{what}: graph capture failed: {exc}")
        print(f"{what}: batch {B}, {dt * 1e3:.2f} ms/step, {B / dt:.0f} meshes/s")
    elif what == "faust_lap":
        ds = dc.TorusBodies(4, device=dev)
        model = dc.SiameseModel("lap", 15).to(dev).train()
        opt = dc.make_optimizer(model)
        k = [0]

        def step():
            k[0] += 1
            dc.train_step(model, opt, ds, k[0] % 4, (k[0] + 1) % 4)
        dt = timed(step, steps)
        print(f"{what}: 1 pair of 6890-vertex bodies (padded 7000), {dt * 1e3:.2f} ms/step, {2 / dt:.1f} meshes/s")
        try:
            g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))

            def gstep():
                k[0] += 1
                g(dc.PairBatch(ds, k[0] % 4, (k[0] + 1) % 4))
            dtg = timed(gstep, steps)
            print(f"{what}: hipGraph replay of fwd+loss+bwd: {dtg * 1e3:.2f} ms/step, {2 / dtg:.1f} meshes/s")
        except Exception as exc:  # noqa: BLE001
            print(f"{what}: graph capture failed: {exc!r}")
    elif what == "arap_lap":
        ds = arap.ClothSequences([(71, 71)] * 64, frames=44, op_frames=2, seed=3, device=dev, model="lap")
        model = arap.Model(15).to(dev).train()
        opt = arap.make_optimizer(model)
        ids = np.arange(64)
        bucket = dp.FlatGradBucket(model.parameters())          # stored gradients, as bench.py's step
        dt = timed(lambda: arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids), grad_sync=bucket.sync,
                                           zero_grads=bucket.detach_grads), steps)
        print(f"{what}: batch 64 x 71x71, {dt * 1e3:.2f} ms/step, {64 / dt:.0f} meshes/s")
    elif what == "arap_ragged":
        # a ragged batch (config-5-like sizes: 64 cloth meshes with 1 000 .. 10 000 vertices): padded as the reference batches
        # (every mesh to the batch maximum) against PACKED (no padding rows, PackedSegments; BatchNorm over real rows only)
        vs = np.random.default_rng(5).integers(1000, 10001, size=64)
        grids = [(int(np.sqrt(v)), int(v) // int(np.sqrt(v))) for v in vs]
        ds = arap.ClothSequences(grids, frames=44, op_frames=2, seed=3, device=dev, model="dir")
        ids = np.arange(64)
        for packed in (False, True):
            torch.manual_seed(0)
            model = arap.DirModel().to(dev).train()
            opt = arap.make_optimizer(model)
            bucket = dp.FlatGradBucket(model.parameters())
            dt = timed(lambda: arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids, packed=packed),
                                               grad_sync=bucket.sync, zero_grads=bucket.detach_grads), steps)
            rows = int(ds.num_vertices.sum()) if packed else 64 * int(ds.num_vertices.max())
            print(f"{what}: 64 ragged meshes (sum V = {int(ds.num_vertices.sum())}, max V = {int(ds.num_vertices.max())}), "
                  f"{'packed' if packed else 'padded'}: {rows} vertex rows, {dt * 1e3:.2f} ms/step, {64 / dt:.0f} meshes/s")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
