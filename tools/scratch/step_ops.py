"""Which Python lines of the ARAP step issue the small copies / fills (torch.profiler with stacks)."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap, dp  # noqa: E402

dev = torch.device("cuda")
n = 64
ds = arap.ClothSequences([(71, 71)] * n, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device=dev, model="dir",
                         operators="pool")
model = arap.DirModel().to(dev).train()
bucket = dp.FlatGradBucket(model.parameters())
opt = arap.make_optimizer(model)
rng = np.random.default_rng(1)
ids = np.arange(n)


def step():
    b = ds.sample_batch(n, rng, seq_ids=ids)
    return arap.train_step(model, opt, b, global_batch=n, grad_sync=bucket.sync, zero_grads=bucket.detach_grads)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::new_zeros")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if ("surfacenetworks_amd" in f or "bench" in f)][:3]
        cnt[(ev.name, " <- ".join(f.split("/")[-1][:70] for f in frames) or "(autograd engine / no python frame)")] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d} {name:14s} {where}")
