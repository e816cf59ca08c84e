"""set_target + set_source of a small cloud, 50 times: for a kernel trace of where a small call's build time goes"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synth
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
d_src, d_tgt, d_nrm = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
for _ in range(5):
    eng.set_target(d_tgt, d_nrm); eng.set_source(d_src)
torch.cuda.synchronize()
tt, ts = [], []
for _ in range(50):
    t0 = time.perf_counter(); eng.set_target(d_tgt, d_nrm); torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.set_source(d_src); torch.cuda.synchronize(); t2 = time.perf_counter()
    tt.append(t1 - t0); ts.append(t2 - t1)
print("n", n, "set_target ms", round(float(np.median(tt)) * 1e3, 4), "set_source ms", round(float(np.median(ts)) * 1e3, 4), flush=True)
