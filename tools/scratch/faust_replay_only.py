"""capture the FAUST pair step once, then replay N times (for kernel traces of the replay alone: trace N=10 and N=50, subtract)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import dense_correspondence as dc
n = int(sys.argv[1])
dev = "cuda"
ds = dc.TorusBodies(4, device=dev)
model = dc.SiameseModel("lap", 15).to(dev).train()
opt = dc.make_optimizer(model)
g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))
for k in range(n):
    g(dc.PairBatch(ds, k % 4, (k + 1) % 4))
torch.cuda.synchronize()
