import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_oracle_golden as T
from helpers import grad_signature
gd = "/root/repo/tests/golden"
rb, ops = T._batch_ops(gd)
g = T.load(gd, "models_reference.npz")
for threads in (1, 8, 64, None):
    if threads: torch.set_num_threads(threads)
    sigs = []
    for rep in range(2):
        loss, out, m = T.run_model("arap_dir", gd, T.ORACLE_LIB, rb, ops, g)
        s = grad_signature(m)
        sigs.append(s)
    k = "conv1.fc.weight"
    print("threads", torch.get_num_threads(), "loss", loss.item(), float(g["arap_dir_loss"]), "sig", sigs[0][k][:2], sigs[1][k][:2], "golden", g[f"arap_dir_psig_{k}"][:2])
    k = "conv2.fc.weight"
    print("   last layer", sigs[0][k][:2], "golden", g[f"arap_dir_psig_{k}"][:2])
    k = "rn14.bn_fc1.fc.weight"
    print("   rn14", sigs[0][k][:2], "golden", g[f"arap_dir_psig_{k}"][:2])
    k = "rn7.bn_fc1.fc.weight"
    print("   rn7", sigs[0][k][:2], "golden", g[f"arap_dir_psig_{k}"][:2])
