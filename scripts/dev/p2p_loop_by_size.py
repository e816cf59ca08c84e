"""Point-to-point loops by size: 30 iterations on the bench's clean clouds (no normals), ms per iteration -- for the limit
of the one-launch iteration (MI_ICP_FUSED_MAX / MI_ICP_NO_FUSED_ITERATION)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
eng = Engine(0)
tag = "FUSED_MAX=%s NO_FUSED=%s" % (os.environ.get("MI_ICP_FUSED_MAX", "-"), os.environ.get("MI_ICP_NO_FUSED_ITERATION", "-"))
out = []
for n in [int(a) for a in sys.argv[1:]] or [50_000, 113_662, 170_000, 250_000, 400_000, 700_000]:
    src, tgt, nrm, T_gt, max_dist = synth(n)
    eng.set_target(torch.from_numpy(tgt).cuda())
    eng.set_source(torch.from_numpy(src).cuda())
    eng.icp_begin(_lib.EST_POINT_TO_POINT, max_dist, None, -1.0)
    eng.icp_iterate(10)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.icp_iterate(32)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 32 * 1e3)
    out.append("%d: %.4f" % (n, float(np.median(ts))))
print(tag, "| ms per point-to-point iteration:", "  ".join(out), flush=True)
