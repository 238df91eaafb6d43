"""What the gradient synchronisation of a step costs at ONE rank through RCCL (dp.FlatGradBucket.sync: one multi-tensor pack
into the 4 MB flat bucket + ncclAllReduce(SUM) + re-pointing the .grad views): device time (events) and host time per call.
The question it answers: is a side stream overlapped with the tail of the backward worth building?"""
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import arap, dp  # noqa: E402

sock = socket.socket(); sock.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sock.getsockname()[1]); sock.close()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank, local, world, dev = dp.init_distributed("nccl", single_rank_group=True)
model = arap.DirModel().to(dev)
bucket = dp.FlatGradBucket(model.parameters(), always_reduce=True)
grads = [torch.randn_like(p) for p in model.parameters()]


def once():
    for p, g in zip(model.parameters(), grads):
        p.grad = g
    bucket.sync()


for _ in range(10):
    once()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
host = 0.0
dev_ms = 0.0
for _ in range(n):
    for p, g in zip(model.parameters(), grads):
        p.grad = g
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.record()
    bucket.sync()
    e.record()
    host += time.perf_counter() - t0
    torch.cuda.synchronize()
    dev_ms += s.elapsed_time(e)
print(f"FlatGradBucket.sync at one rank through RCCL ({bucket.nbytes} bytes, {len(grads)} parameters): device {dev_ms / n * 1e3:.1f} us, "
      f"host {host / n * 1e6:.1f} us per call (pack + ncclAllReduce + re-pointing the views)")
torch.distributed.destroy_process_group()
