import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import deterministic_init
from surfacenetworks_amd import arap, blocks as snB, kernels
snB._TILE_SUMS_MIN_ROWS = 0
ds = arap.ClothSequences([(12, 11), (9, 13), (10, 10)], frames=45, op_frames=2, seed=3, device="cuda", model="dir")
seq, off = np.array([0, 1, 2, 1]), np.array([0, 0, 0, 0])
for packed in (False, True):
    res = []
    for on in (True, False):
        kernels.tile_sums_supported = (lambda: True) if on else (lambda: False)
        model = deterministic_init(arap.DirModel(), 4).cuda().train()
        b = ds.sample_batch(4, None, seq_ids=seq, offsets=off, packed=packed)
        loss, _ = arap.forward_loss(model, b, 4)
        loss.backward()
        res.append((loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double()))
    print("packed" if packed else "padded", "loss", res[0][0], res[1][0], "grad rel diff", float((res[0][1] - res[1][1]).norm() / res[1][1].norm()))
