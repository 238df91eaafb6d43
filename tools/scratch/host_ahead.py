"""Is the host ahead of the GPU in the eager training loop?  Per-iteration host time without any synchronisation."""
import time

import numpy as np
import torch

from surfacenetworks_amd import arap, dp

ds = arap.ClothSequences([(71, 71)] * 64, frames=44, op_frames=2, seed=3, device="cuda", model="dir")
model = arap.DirModel().cuda().train()
bucket = dp.FlatGradBucket(model.parameters())
opt = arap.make_optimizer(model)
rng = np.random.default_rng(0)
ids = np.arange(64)


def step():
    b = ds.sample_batch(64, rng, seq_ids=ids)
    return arap.train_step(model, opt, b, global_batch=64, grad_sync=bucket.sync, zero_grads=bucket.detach_grads).detach()


for _ in range(5):
    step()
torch.cuda.synchronize()
t = [time.perf_counter()]
for _ in range(12):
    step()
    t.append(time.perf_counter())
torch.cuda.synchronize()
t.append(time.perf_counter())
d = np.diff(np.array(t)) * 1e3
print("host ms per iteration:", np.round(d[:-1], 2), " final sync wait:", round(d[-1], 2))
