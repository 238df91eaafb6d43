"""Per-step allocator behaviour of the ARAP training step (is the caching allocator in steady state?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from surfacenetworks_amd import arap, dp

n = int(os.environ.get("MESHES", 64))
ds = arap.ClothSequences([(71, 71)] * n, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda", model="dir")
model = arap.DirModel().cuda().train()
bucket = dp.FlatGradBucket(model.parameters())
opt = arap.make_optimizer(model)
rng = np.random.default_rng(10)
ids = np.arange(n)
for i in range(12):
    b = ds.sample_batch(n, rng, seq_ids=ids)
    loss = arap.train_step(model, opt, b, global_batch=n, grad_sync=bucket.all_reduce)
    del b
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(i, "allocated GiB %.3f" % (torch.cuda.memory_allocated() / 2**30), "reserved GiB %.3f" % (torch.cuda.memory_reserved() / 2**30),
          "hipMalloc", st["num_device_alloc"], "peak %.3f" % (st["allocated_bytes.all.peak"] / 2**30), flush=True)
