import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap, kernels, plans
kernels.AVG_BWD_MERGE_MAX = int(sys.argv[1])
ds = arap.ClothSequences([(71, 71)] * 64, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda", model="dir")
model = arap.DirModel().cuda().train(); opt = arap.make_optimizer(model)
rng = np.random.default_rng(10); ids = np.arange(64)
for _ in range(8): arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids))
torch.cuda.synchronize()
