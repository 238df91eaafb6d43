"""How long the Linear / weight-gradient / SpMM kernels take on SMALL operands (launch + prologue + one tile), back to back."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels

def t(f, n=50):
    """device time per launch: n launches captured into one graph, replayed"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (10 * n) * 1e3

