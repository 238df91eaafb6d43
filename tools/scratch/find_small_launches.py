"""Which host call sites launch the small fill / copy kernels of the ARAP step?  (torch profiler, CPU side with stacks:
every aten op that ends in a memset / memcpy / fill / copy kernel is listed with the innermost frame of this package.)"""
import collections
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, ".")
from surfacenetworks_amd import arap, dp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ds = arap.ClothSequences([(31, 31)] * n, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=1,
                         device="cuda", model="dir")
model = arap.DirModel().cuda().train()
opt = arap.make_optimizer(model)
bucket = dp.FlatGradBucket(model.parameters(), always_reduce=False)
rng = np.random.default_rng(0)
seq_ids = np.arange(n)


def step():
    b = ds.sample_batch(n, rng, seq_ids=seq_ids)
    return arap.train_step(model, opt, b, global_batch=n, grad_sync=bucket.sync, zero_grads=bucket.detach_grads)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
names = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::_to_copy",
         "aten::empty_like", "aten::contiguous", "aten::cat", "aten::add_", "aten::mul_")
hist = collections.Counter()
for e in prof.events():
    if e.name not in names:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name in names:
        continue                                   # count the outermost aten op only
    frame = next((s for s in (e.stack or []) if "surfacenetworks_amd" in s or "bench.py" in s), (e.stack or ["?"])[0] if e.stack else "?")
    shapes = str(e.input_shapes)[:60] if e.input_shapes else ""
    hist[(e.name, frame.strip()[-110:], shapes)] += 1
for (name, frame, shapes), c in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {name:18s} {frame}  {shapes}")
gpu = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        gpu[e.name[:80]] += 1
print("---- device activities of the step")
for k, c in sorted(gpu.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d}  {k}")
