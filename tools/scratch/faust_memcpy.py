"""which torch ops issue device-to-device copies in one FAUST pair step"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import dense_correspondence as dc
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
ds = dc.TorusBodies(4, device=dev)
model = dc.SiameseModel("lap", 15).to(dev).train()
opt = dc.make_optimizer(model)
for k in range(3):
    dc.train_step(model, opt, ds, k % 4, (k + 1) % 4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    dc.train_step(model, opt, ds, 0, 1)
    torch.cuda.synchronize()
evs = prof.events()
# device memcpy events and their cpu parents
n = 0
from collections import Counter
c = Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in e.name or "copyBuffer" in e.name):
        n += 1
        c[e.name] += 1
print(n, c)
c2 = Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_foreach_copy_"):
        st = [s for s in (e.stack or []) if "surfacenetworks_amd" in s or "tools/" in s]
        c2[(e.name, str(e.input_shapes)[:60], st[0] if st else "?")] += 1
for k, v in c2.most_common(40):
    print(v, k)
