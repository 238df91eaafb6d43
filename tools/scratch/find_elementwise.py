import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap, plans
ds = arap.ClothSequences([(71, 71)] * 64, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda", model="dir")
model = arap.DirModel().cuda().train(); opt = arap.make_optimizer(model)
rng = np.random.default_rng(10); ids = np.arange(64)
for _ in range(4): arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids))
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name == "CUDA" or "CUDA" in str(e.device_type)]
import collections
agg = collections.Counter()
for e in prof.key_averages(group_by_stack_n=6):
    if e.device_time_total > 30 and ("aten::" in e.key):
        print(f"{e.key:40s} n={e.count:4d} cuda {e.device_time_total:9.1f} us", [s for s in e.stack[:6] if "site-packages/torch" not in s][:4])
