#!/usr/bin/env python3
"""The 10M bench workload with measurement noise and partial overlap: source = a random 60 % of
the target's points + N(0, sigma) per coordinate (sigma in units of the mean spacing s), moved
by the bench's T_gt.  Reports the converged iteration's cost and traversal census per sigma --
how much of the headline rate rests on the synthetic workload's exact correspondences."""
import ctypes as C
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
eng = Engine(0)
src0, tgt, nrm, T_gt, max_dist = synth(n)
s = n ** (-1.0 / 3.0)
d_tgt, d_nrm = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
eng.set_target(d_tgt, d_nrm)
rng = np.random.default_rng(5)
for sigma in (0.0, 0.05, 0.15, 0.3):
    keep = rng.random(n) < 0.6
    src = src0[keep] + (rng.normal(0.0, sigma * s, (int(keep.sum()), 3)).astype(np.float32) if sigma > 0 else 0)
    eng.set_source(torch.from_numpy(np.ascontiguousarray(src, np.float32)).cuda())
    eng.set_profiling(True)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(10)
    p0 = eng.get_profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.icp_iterate(20)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p1 = eng.get_profile()
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    out = (C.c_uint64 * 8)()
    a = np.ascontiguousarray(T.T)
    eng._chk(eng._L.mi_icp_debug_nn_stats8(eng._ctx, a.ctypes.data_as(C.c_void_p), float(max_dist), 1, out))
    print(json.dumps({"row": "10M target, 60 %% of it as source + noise", "sigma_over_spacing": sigma, "n_source": len(src),
                      "ms_per_iter": dt / 20 * 1e3, "nn_ms": (p1["nn_ms"] - p0["nn_ms"]) / 20, "reduce_ms": (p1["reduce_ms"] - p0["reduce_ms"]) / 20,
                      "records_per_packet": out[0] / out[2], "leaf_batches_per_packet": out[1] / out[2],
                      "halo_lines_per_packet": out[4] / out[2], "packets_with_halo_phase": out[5] / out[2],
                      "packets_that_walk": out[6] / out[2], "lanes_walking_per_walking_packet": out[7] / max(out[6], 1),
                      "fitness": res.fitness, "rmse_over_spacing": res.inlier_rmse / s,
                      "T_err": float(np.linalg.norm(T - T_gt))}), flush=True)
