"""A/B of the merged backward launch of the global-average stages (sn_avg_bn_bwd_f32) at 64 meshes: config-3 step with the merge
limit at 8 (three launches at 64 meshes) and lifted (one launch)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap, kernels, plans

def run(limit, meshes=64, steps=30):
    kernels.AVG_BWD_MERGE_MAX = limit
    plans.reset()
    torch.manual_seed(1)
    ds = arap.ClothSequences([(71, 71)] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda", model="dir")
    model = arap.DirModel().cuda().train()
    opt = arap.make_optimizer(model)
    rng = np.random.default_rng(10); ids = np.arange(meshes)
    step = lambda: arap.train_step(model, opt, ds.sample_batch(meshes, rng, seq_ids=ids))
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for m in (16, 32, 64):
    for rep in range(3):
        for lim in (8, 1 << 20):
            print(f"{m} meshes rep {rep} merge limit {lim}: {run(lim, m):.3f} ms/step", flush=True)
