# Do the addresses a plan runs on repeat from step to step (the torch caching allocator hands the same blocks to the same requests)?
# Fraction of plan runs whose (plan, slot addresses) were seen in an earlier step, for the three eager loops of tools/plan_probe.py.
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plan_probe
from surfacenetworks_amd import plans
seen, hits, total, last = {}, [0], [0], {}
real = plans.Plan._launch
def spy(self, b):
    k = (id(self), bytes(b))
    total[0] += 1
    if k in seen: hits[0] += 1
    seen[k] = seen.get(k, 0) + 1
    return real(self, b)
plans.Plan._launch = spy
for name, mk in (("arap4", lambda: plan_probe.arap_step(4)), ("mnist", plan_probe.mnist_step), ("faust", plan_probe.faust_step)):
    step = mk()
    for _ in range(5): step()
    seen.clear(); hits[0] = total[0] = 0
    per = []
    for s in range(12):
        h0, t0 = hits[0], total[0]
        step()
        per.append(f"{hits[0]-h0}/{total[0]-t0}")
    torch.cuda.synchronize()
    keys_per_plan = {}
    for (pid, _), n in seen.items(): keys_per_plan.setdefault(pid, []).append(n)
    print(name, "hits/runs per step:", " ".join(per), "| distinct address sets per plan: max", max(len(v) for v in keys_per_plan.values()),
          "mean %.1f" % (sum(len(v) for v in keys_per_plan.values()) / len(keys_per_plan)))
