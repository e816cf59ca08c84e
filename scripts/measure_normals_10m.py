import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from cupoch_amd.engine import Engine
eng=Engine(0)
d=torch.rand((10_000_000,3),device='cuda')
for k in (30,):
    eng.estimate_normals_knn(d,k); torch.cuda.synchronize()
    t0=time.perf_counter(); eng.estimate_normals_knn(d,k); torch.cuda.synchronize()
    print("normals 10M k",k,"%.2f ms"%((time.perf_counter()-t0)*1e3))
