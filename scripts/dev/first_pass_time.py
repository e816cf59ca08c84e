"""The pass without previous matches at 10M points, timed by HIP events: default (wave walk from the root) or, with
MI_ICP_FIRST_SOLO=1, every lane walking on its own; results compared with a fresh default-form search."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
eng.set_source(torch.from_numpy(src).cuda())
eng.set_profiling(True)
ts = []
for _ in range(4):
    eng.drop_seeds()
    p0 = eng.get_profile()
    idx, d2, st = eng.search_radius_1nn(max_dist)
    p1 = eng.get_profile()
    ts.append(p1["nn_ms"] - p0["nn_ms"])
print("first pass %s: %.4f ms (all: %s), matches %d, sum d2 %.9g, idx checksum %d" % (
    "SOLO" if os.environ.get("MI_ICP_FIRST_SOLO") else "wave walk", float(np.median(ts)), ["%.3f" % t for t in ts], int(st[0]), float(st[1]), int(idx.astype(np.int64).sum())))
