"""First passes of the bench's 10M-point pair (exact correspondences, then the noisy source) -- for a kernel trace of the
group-stationary search (scripts/gpu_r06_d.sh)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synth
from cupoch_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
s = float(n) ** (-1.0 / 3.0)
rng = np.random.default_rng(6)
noisy = (src + rng.normal(0.0, 0.15 * s, src.shape)).astype(np.float32)
for name, cloud in (("exact", src), ("noisy", noisy)):
    eng.set_source(torch.from_numpy(cloud).cuda())
    eng.set_profiling(True)
    ts = []
    for _ in range(5):
        eng.drop_seeds()
        p0 = eng.get_profile()
        r = eng.evaluate_registration(max_dist)
        p1 = eng.get_profile()
        ts.append(p1["nn_ms"] - p0["nn_ms"])
    print(name, "first pass ms", [round(t, 4) for t in ts], "kind", eng.last_search_kind(), "fitness", r.fitness, flush=True)
eng.close()
