"""capture the Mesh-MNIST Dirac step (batch 512) once, then replay N times (kernel traces of the replay alone)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import mesh_mnist as mm
n = int(sys.argv[1]); dev = "cuda"; B = 512
rng = np.random.default_rng(0); torch.manual_seed(0)
ds = mm.MeshDigits(B, seed=2, device=dev, fixed_vertices=150, model="dir")
model = mm.DirModel().to(dev).train(); opt = mm.make_optimizer(model); ids = np.arange(B)
g = mm.graphed_train_step(model, opt, ds.sample_batch(B, rng, ids=ids)) if hasattr(mm, "graphed_train_step") else None
if g is None:
    from surfacenetworks_amd.graphs import GraphedTrainStep
    g = GraphedTrainStep(model, opt, ds.sample_batch(B, rng, ids=ids), lambda m, b: mm.forward_loss(m, b)[0])
for k in range(n):
    g(ds.sample_batch(B, rng, ids=ids))
torch.cuda.synchronize()
