"""cProfile of eager ARAP steps at a host-bound batch (8 meshes): where the Python time of a step goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap  # noqa: E402

meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ds = arap.ClothSequences([(71, 71)] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device=dev, model="dir")
torch.manual_seed(1)
model = arap.DirModel().to(dev).train()
opt = arap.make_optimizer(model)
rng = np.random.default_rng(10)
ids = np.arange(meshes)
step = lambda: arap.train_step(model, opt, ds.sample_batch(meshes, rng, seq_ids=ids))
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{meshes} meshes: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, with the device drained {1e3 * (t2 - t0) / 20:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
