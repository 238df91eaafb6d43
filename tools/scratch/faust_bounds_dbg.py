import copy, sys, collections
import torch
sys.path.insert(0, ".")
from surfacenetworks_amd import dense_correspondence as dc, kernels, functional as F

cnt = collections.Counter()
real_wgrad, real_seg, real_rb4, real_ring, real_csr_epi = kernels.wgrad, kernels.wgrad_seg, kernels.spmm_rb4, kernels.spmm_ring, kernels.spmm_csr_elubwd
def w(*a, bounds=None, **k):
    cnt["wgrad bounded" if bounds is not None else "wgrad bf16"] += 1
    return real_wgrad(*a, bounds=bounds, **k)
def ws(*a, bounds=None, **k):
    cnt["wgrad_seg bounded" if bounds is not None else "wgrad_seg bf16"] += 1
    return real_seg(*a, bounds=bounds, **k)
def rb4(*a, **k):
    cnt["rb4" + (" epi" if len(a) > 7 and a[7] is not None else "")] += 1
    return real_rb4(*a, **k)
def ring(*a, **k):
    cnt["ring" + (" epi" if len(a) > 7 and a[7] is not None else "")] += 1
    return real_ring(*a, **k)
def csre(*a, **k):
    cnt["csr epi"] += 1
    return real_csr_epi(*a, **k)
kernels.wgrad, kernels.wgrad_seg, kernels.spmm_rb4, kernels.spmm_ring, kernels.spmm_csr_elubwd = w, ws, rb4, ring, csre

torch.manual_seed(3)
ds = dc.TorusBodies(3, n=8, m=9, pad_to=80, seed=4, device="cuda")
model_e = dc.SiameseModel("lap", 15).cuda().train()
model_g = copy.deepcopy(model_e)
opt_e, opt_g = dc.make_optimizer(model_e), dc.make_optimizer(model_g)
cnt.clear()
graphed = dc.graphed_train_step(model_g, opt_g, dc.PairBatch(ds, 0, 1))
print("capture (incl. warm-up runs):", dict(cnt)); cnt.clear()
le = dc.train_step(model_e, opt_e, ds, 1, 2)
print("eager step:", dict(cnt))
