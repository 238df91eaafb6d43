"""cProfile of the eager FAUST pair step (host side): where do the ~16 us per launch go"""
import cProfile, pstats, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import dense_correspondence as dc
dev = "cuda"
ds = dc.TorusBodies(4, device=dev)
model = dc.SiameseModel("lap", 15).to(dev).train()
opt = dc.make_optimizer(model)
for k in range(5):
    dc.train_step(model, opt, ds, k % 4, (k + 1) % 4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(10):
    dc.train_step(model, opt, ds, k % 4, (k + 1) % 4)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
