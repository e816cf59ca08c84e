import sys, time, json, numpy as np, torch
sys.path.insert(0, '.')
from bench import synth
from cupoch_amd.engine import Engine
eng = Engine(0)
n5 = 5_000_000
src, tgt, nrm, T_gt, max_dist = synth(n5)
d_src, d_tgt, d_nrm = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
tcov = eng.covariances_from_normals(d_nrm, 1e-3)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.set_target(d_tgt, d_nrm, tcov); eng.synchronize()
    t1 = time.perf_counter()
    eng.set_source(d_src, None, tcov); eng.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"rep": rep, "set_target_ms": (t1 - t0) * 1e3, "set_source_ms": (t2 - t1) * 1e3}))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.set_target(d_tgt, d_nrm); eng.synchronize()
    print(json.dumps({"rep": rep, "set_target_no_cov_ms": (time.perf_counter() - t0) * 1e3}))
