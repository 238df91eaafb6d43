"""Which autograd node launches the large fill_ in the backward of the ARAP Dirac model?"""
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

from surfacenetworks_amd import arap

ds = arap.ClothSequences([(31, 31)] * 8, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=1,
                         device="cuda", model="dir")
model = arap.DirModel().cuda().train()
opt = arap.make_optimizer(model)
rng = np.random.default_rng(0)
for _ in range(2):
    arap.train_step(model, opt, ds.sample_batch(8, rng), global_batch=8)
b = ds.sample_batch(8, rng)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    arap.train_step(model, opt, b, global_batch=8)
evs = [e for e in prof.events() if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like")]
for e in evs:
    shp = e.input_shapes
    par = e.cpu_parent
    chain = []
    while par is not None and len(chain) < 6:
        chain.append(par.name)
        par = par.cpu_parent
    print(e.name, shp, " <- ".join(chain))
    if e.stack:
        print("    ", " | ".join(e.stack[:4]))
