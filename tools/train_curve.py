"""Loss curve of the ARAP Dirac model over a few dozen steps (same seeds): run once per SN_GEMM_VARIANT to compare the
split-bf16 Linear kernels with the fp32-MFMA ones in actual training.  usage: train_curve.py [steps] [meshes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import arap  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(7)
ds = arap.ClothSequences([(71, 71)] * n, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda", model="dir")
model = arap.DirModel().cuda().train()
opt = arap.make_optimizer(model)
rng = np.random.default_rng(10)
losses = []
for i in range(steps):
    b = ds.sample_batch(n, rng, seq_ids=np.arange(n))
    losses.append(float(arap.train_step(model, opt, b, global_batch=n).item()))
print("variant", os.environ.get("SN_GEMM_VARIANT", "1"), " ".join(f"{l:.6g}" for l in losses[::max(1, steps // 10)]), "last", f"{losses[-1]:.6g}")
