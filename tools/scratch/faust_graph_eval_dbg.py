import copy, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import deterministic_init
from surfacenetworks_amd import dense_correspondence as dc, dp
dev = torch.device("cuda")
for mode in ("eval", "train"):
    for pad in (256, 1024):
        ds = dc.TorusBodies(3, n=13, m=17, pad_to=pad, seed=5, device=dev)
        model = deterministic_init(dc.SiameseModel("lap", 3), 11).to(dev)
        model = model.eval() if mode == "eval" else model.train()
        mg = copy.deepcopy(model)
        opt, optg = dc.make_optimizer(model), dc.make_optimizer(mg)
        loss = dc.train_step(model, opt, ds, 0, 1)
        ge = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        step = dc.graphed_train_step(mg, optg, dc.PairBatch(ds, 0, 1))
        lg = step(dc.PairBatch(ds, 0, 1))
        gg = torch.cat([p.grad.reshape(-1) for p in mg.parameters()])
        print(mode, pad, "eager loss", loss.item(), "graph loss", lg.item(), "nan grads", int(torch.isnan(gg).sum()), "of", gg.numel(),
              "rel", ((gg - ge).norm() / ge.norm()).item())
        bad = [k for (k, p) in mg.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()]
        print("   first bad:", bad[:6])
