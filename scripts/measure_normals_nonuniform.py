import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from cupoch_amd.engine import Engine
eng=Engine(0)
rng = np.random.default_rng(77)
sizes = rng.integers(1, 4000, 60)
centres = rng.random((60, 3)).astype(np.float32)
tgt = np.concatenate([c + rng.normal(0, 0.002, (s, 3)).astype(np.float32) for c, s in zip(centres, sizes)] + [rng.random((300, 3), dtype=np.float32)])
for name, pts in (("clustered 117k", tgt), ("uniform 2M", rng.random((2_000_000,3),dtype=np.float32)), ("clusters of 1..40 pts, 50k clusters", np.concatenate([c + rng.normal(0,1e-4,(s,3)).astype(np.float32) for c,s in zip(rng.random((50000,3)).astype(np.float32), rng.integers(1,40,50000))]))):
    d=torch.from_numpy(pts).cuda()
    for k in (10,30):
        eng.estimate_normals_knn(d,k); torch.cuda.synchronize()
        t0=time.perf_counter(); eng.estimate_normals_knn(d,k); torch.cuda.synchronize()
        print(name, len(pts), "k",k, "%.2f ms"%((time.perf_counter()-t0)*1e3))
