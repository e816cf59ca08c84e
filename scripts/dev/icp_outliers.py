"""1-NN search / ICP with a correspondence radius far beyond the point spacing and outliers in the source."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
n = 1_000_000
tgt = rng.random((n, 3), dtype=np.float32)
nrm = rng.standard_normal((n, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
src = tgt + np.float32(0.002)
def run(name, s, r):
    eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
    eng.set_source(torch.from_numpy(np.ascontiguousarray(s, np.float32)).cuda())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.registration_icp(_lib.EST_POINT_TO_PLANE, r, None, 0.0, 0.0, 10, -1.0)
    torch.cuda.synchronize()
    print("%-60s radius %6.2f  10 iterations %.2f ms fitness %.4f" % (name, r, (time.perf_counter() - t0) * 1e3, res.fitness), flush=True)
for r in (0.02, 0.5, 50.0):
    run("clean source", src, r)
    out = np.concatenate([src, (rng.random((500, 3), dtype=np.float32) * 40 - 20).astype(np.float32)])
    run("source + 500 points scattered 40x wider", out, r)
