import torch, numpy as np
from surfacenetworks_amd import kernels
rng=np.random.default_rng(0)
for rows in (1,2,3):
    dy=torch.from_numpy(rng.standard_normal((rows,128)).astype(np.float32)).cuda()
    x=torch.from_numpy((rng.standard_normal((rows,128))*1.5).astype(np.float32)).cuda()
    mean=x.mean(0); var=x.var(0,unbiased=False) if rows>1 else torch.zeros(128,device='cuda')
    invstd=(1/torch.sqrt(var+1e-5))
    b=(dy.abs().max().reshape(1), invstd, rows)
    G,s=kernels.wgrad(dy,x,mean,want_colsum=True,bounds=b)
    G0,s0=kernels.wgrad(dy,x,mean,want_colsum=True)
    print(rows, torch.isnan(G).sum().item(), float((G-G0).abs().max()), float(G0.abs().max()))
