#!/usr/bin/env python3
"""config 2 only (1M -> VoxelDownSample(0.02) -> point-to-plane, r = 0.04) + traversal census."""
import ctypes as C
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
eng = Engine(0)
src, tgt, nrm, T_gt, _ = synth(1_000_000)
d_src, d_tgt, d_nrm = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
vt, vn, _ = eng.voxel_downsample(d_tgt, 0.02, d_nrm)
vs, _, _ = eng.voxel_downsample(d_src, 0.02)
eng.set_target(vt, vn); eng.set_source(vs)
eng.set_profiling(True)
eng.icp_begin(_lib.EST_POINT_TO_PLANE, 0.04, None, -1.0)
eng.icp_iterate(3)
p0 = eng.get_profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
res = eng.icp_iterate(30)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
p1 = eng.get_profile()
T = np.array(res.transformation, np.float32).reshape(4, 4).T
out = (C.c_uint64 * 4)()
a = np.ascontiguousarray(T.T)
eng._chk(eng._L.mi_icp_debug_nn_stats(eng._ctx, a.ctypes.data_as(C.c_void_p), 0.04, 1, out))
print(json.dumps({"cells": os.environ.get("MI_ICP_NO_CELLS") is None, "n": len(vt), "it_per_s": 30 / dt, "us_per_iter": dt / 30 * 1e6,
                  "nn_us": (p1["nn_ms"] - p0["nn_ms"]) / 30 * 1e3, "reduce_us": (p1["reduce_ms"] - p0["reduce_ms"]) / 30 * 1e3,
                  "nodes_per_packet": out[0] / out[2], "leaves_per_packet": out[1] / out[2], "max_visits_of_a_packet": out[3]}))
