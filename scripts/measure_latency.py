#!/usr/bin/env python3
"""End-to-end latency of one RegistrationICP call (tree build +
source staging + 30 point-to-plane iterations + result), inputs resident on the device
(or, with MI_ICP_LATENCY_HOST=1, handed over as HOST arrays: the PCIe-inclusive figure).
The frame-to-frame case of section 8(f)-4 (KinFu / odometry callers: small clouds,
launch latency dominates).  One JSON object per size."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth                      # noqa: E402
from cupoch_amd import _lib                  # noqa: E402
from cupoch_amd.engine import Engine         # noqa: E402

if os.environ.get("MI_ICP_GRAPH") or os.environ.get("MI_ICP_OWN_STREAM"):   # (graph capture needs a non-default stream)
    torch.cuda.set_stream(torch.cuda.Stream())
eng = Engine(0)
sizes = [int(s) for s in sys.argv[1:]] or [20_000, 100_000, 307_200, 1_000_000, 10_000_000]
for n in sizes:
    src, tgt, nrm, T_gt, max_dist = synth(n)
    if os.environ.get("MI_ICP_LATENCY_HOST") == "1":
        d_src, d_tgt, d_nrm = src, tgt, nrm          # numpy arrays: the library stages them (mem_kind = host)
    else:
        d_src, d_tgt, d_nrm = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()

    def call():
        eng.set_target(d_tgt, d_nrm)
        t1 = time.perf_counter()
        eng.set_source(d_src)
        t2 = time.perf_counter()
        res = eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, None, 0.0, 0.0, 30, -1.0)
        t3 = time.perf_counter()
        return res, t1, t2, t3

    call()
    rows = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, t1, t2, t3 = call()
        rows.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
    rows = np.median(np.array(rows), axis=0) * 1e3
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    print(json.dumps({"row": "RegistrationICP call, point-to-plane, 30 iterations" +
                             (", HOST inputs (PCIe-inclusive)" if os.environ.get("MI_ICP_LATENCY_HOST") == "1" else ""), "n": n,
                      "total_ms": round(rows[0], 3), "set_target_ms": round(rows[1], 3),
                      "set_source_ms": round(rows[2], 3), "icp_30_iterations_ms": round(rows[3], 3),
                      "iterations": res.iterations, "T_err": float(np.linalg.norm(T - T_gt))}), flush=True)
eng.close()
