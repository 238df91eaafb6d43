# does torch's own zeroing survive graph replays on this runtime?  (tools/scratch/plan_memset_graph.py: hipMemsetAsync does not)
import torch
for n in (12, 4096, 1 << 20, 1 << 26):
    buf = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty_like(buf); out2 = torch.empty_like(buf)
    fbuf = torch.empty(max(n // 4, 1), device="cuda"); fout = torch.empty_like(fbuf)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.zero_(); out.copy_(buf)
        z = torch.zeros_like(buf); out2.copy_(z)
        fbuf.zero_(); fout.copy_(fbuf)
    bad = []
    for rep in range(4):
        buf.fill_(7 + rep); out.fill_(1); out2.fill_(1); fbuf.fill_(3.0); fout.fill_(1.0)
        g.replay(); torch.cuda.synchronize()
        bad.append((int((out != 0).sum()), int((out2 != 0).sum()), int((fout != 0).sum())))
    print("bytes", n, "nonzero after replays (zero_, zeros_like, float zero_)", bad)
