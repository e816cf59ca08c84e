"""One sigma of scripts/measure_noisy.py, a fixed number of converged iterations (for counter passes)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
sigma = float(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
eng = Engine(0)
src0, tgt, nrm, T_gt, max_dist = synth(n)
s = n ** (-1.0 / 3.0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
rng = np.random.default_rng(5)
keep = rng.random(n) < 0.6
src = src0[keep] + (rng.normal(0.0, sigma * s, (int(keep.sum()), 3)).astype(np.float32) if sigma > 0 else 0)
eng.set_source(torch.from_numpy(np.ascontiguousarray(src, np.float32)).cuda())
eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
eng.icp_iterate(12)
torch.cuda.synchronize()
