import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import train_bench
from surfacenetworks_amd import plans
from surfacenetworks_amd.resident import resident_cache
for graphs in (True, False):
    plans.reset(); plans.set_graphs(graphs)
    for permute in (False, True, True):
        g0 = plans.graph_stats()
        out = train_bench.faust_swap("cuda", steps=30, permute=permute)
        g1 = plans.graph_stats()
        print("graphs", graphs, "permute", permute, "%.2f ms/step" % out["ms_per_step"], {k: g1[k] - g0[k] for k in g1},
              "sets per plan max", max((len(p.execs) for s in plans._SITES for p in s.plans.values() if p is not None), default=0), flush=True)
        torch.cuda.empty_cache()
        c = resident_cache()
        if c is not None: c.clear()
