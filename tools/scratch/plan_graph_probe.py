# What would replaying a plan as a captured hipGraph save on the host?  One Dirac block forward plan (small mesh: the device keeps up),
# host time of sn_plan_run against hipGraphLaunch of the same launches at fixed addresses.
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import arap, plans
DEV = "cuda"
torch.manual_seed(0)
ds = arap.ClothSequences([(9, 8)] * 3, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11, device=DEV, model="dir")
model = arap.DirModel().to(DEV).train()
opt = arap.make_optimizer(model)
calls = []
real = plans.Plan.run
def spy(self, big, small, ext):
    calls.append((self, big, small, list(ext)))
    return real(self, big, small, ext)
plans.Plan.run = spy
for k in range(3):
    calls.clear()
    arap.train_step(model, opt, ds.sample_batch(3, np.random.default_rng(1), seq_ids=np.arange(3)), global_batch=3)
plans.Plan.run = real
torch.cuda.synchronize()
def host_us(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6
rows = []
for plan, big, small, ext in calls:
    direct = host_us(lambda: plan.run(big, small, ext))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run(big, small, ext)
    replay = host_us(g.replay)
    rows.append((plan.launches, direct, replay))
tot_l = sum(r[0] for r in rows); tot_d = sum(r[1] for r in rows); tot_g = sum(r[2] for r in rows)
for l, d, r in rows[:6] + rows[-6:]:
    print(f"{l:3d} launches: sn_plan_run {d:6.1f} us   graph launch {r:6.1f} us")
print(f"step: {len(rows)} plan runs, {tot_l} launches: sn_plan_run {tot_d/1e3:.2f} ms ({tot_d/tot_l:.2f} us per launch), graph launches {tot_g/1e3:.2f} ms")
