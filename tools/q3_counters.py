"""The four quaternion-packed Dirac products of config 5 (grid order, packed batch, N = 32), `launches` launches each, for a
counter pass (rocprofv3 --pmc ... -- python tools/q3_counters.py): Di and DiA^T are the FACE-output, write-heavy shapes
(spmm_q3_lds_wide), DiA and Di^T the vertex-output ones (spmm_q3_lds).  Prints the launch order as JSON."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from surfacenetworks_amd import functional as snF  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
Dis, DiAs, _, sumV, sumF, _ = bench._c5_meshes(0, False, False)
sel = np.arange(bench.C5_MESHES_PER_GPU)
plan = []
for name, mats in (("Di", Dis), ("DiA", DiAs)):
    op = OperatorPool(mats, dev, want_bsr4=True).assemble(sel)
    for prod, o in ((name, op), (name + "^T", op.t())):
        M, K = o.shape
        x = torch.randn(K // 4, 128, device=dev, generator=g)
        y = torch.empty(M // 4, 128, device=dev)
        torch.cuda.synchronize()
        for _ in range(launches):
            snF._launch(o, x, y, 4, "c5")
        torch.cuda.synchronize()
        plan.append({"product": prod, "launches": launches, "M": M, "K": K, "blocks": int(o.q3()[1].shape[0])})
print(json.dumps({"sumV": sumV, "sumF": sumF, "plan": plan}))
