import copy, sys
import torch
sys.path.insert(0, ".")
from surfacenetworks_amd import dense_correspondence as dc, kernels

torch.manual_seed(3)
ds = dc.TorusBodies(3, n=8, m=9, pad_to=80, seed=4, device="cuda")
m0 = dc.SiameseModel("lap", 15).cuda().train()
def grads(model, graph):
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    if graph:
        g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))
        loss = g(dc.PairBatch(ds, 1, 2))
    else:
        loss = dc.train_step(model, opt, ds, 1, 2)
    torch.cuda.synchronize()
    return loss.item(), [p.grad.clone() for p in model.parameters()]
le, ge = grads(copy.deepcopy(m0), False)
le2, ge2 = grads(copy.deepcopy(m0), False)
lg, gg = grads(copy.deepcopy(m0), True)
def md(a, b):
    return max(float((x - y).abs().max()) for x, y in zip(a, b)), max(float(x.abs().max()) for x in a)
print("loss eager/eager2/graph", le, le2, lg)
print("eager vs eager2: max |dgrad| %g (max |grad| %g)" % md(ge, ge2))
print("eager vs graph : max |dgrad| %g (max |grad| %g)" % md(ge, gg))
worst = max(((float((x - y).abs().max()), i) for i, (x, y) in enumerate(zip(ge, gg))))
names = [n for n, _ in m0.named_parameters()]
print("worst parameter:", names[worst[1]], worst[0])
