#!/usr/bin/env python3
"""Traversal census of the nearest-neighbour kernel on the bench workload
(node / leaf visits per 64-query packet, seeded vs unseeded).  Tuning aid."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth                      # noqa: E402
from cupoch_amd import _lib                  # noqa: E402
from cupoch_amd.engine import Engine         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
eng.set_source(torch.from_numpy(src).cuda())


def census(seed, T=None):
    out = (C.c_uint64 * 4)()
    tp = None
    if T is not None:
        a = np.ascontiguousarray(np.asarray(T, np.float32).T)
        tp = a.ctypes.data_as(C.c_void_p)
    eng._chk(eng._L.mi_icp_debug_nn_stats(eng._ctx, tp, float(max_dist), int(seed), out))
    steps, leaves, pk = out[0], out[1], out[2]
    return steps / pk, leaves / pk, pk


print("n=%d packets: unseeded identity  nodes/packet=%.1f leaf-batches/packet=%.1f (%d packets)" % ((n,) + census(0)))
print("seeded, same T                  nodes/packet=%.1f leaf-batches/packet=%.1f" % census(1)[:2])
print("seeded, T_gt (converged)        nodes/packet=%.1f leaf-batches/packet=%.1f" % census(1, T_gt)[:2])
print("seeded, T_gt again              nodes/packet=%.1f leaf-batches/packet=%.1f" % census(1, T_gt)[:2])
print("unseeded, T_gt                  nodes/packet=%.1f leaf-batches/packet=%.1f" % census(0, T_gt)[:2])

# the registration loop re-orders the source by its matches after the first pass
eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, T_gt, -1.0)
print("after icp_begin (match-ordered source):")
print("seeded, T_gt                    nodes/packet=%.1f leaf-batches/packet=%.1f" % census(1, T_gt)[:2])
print("unseeded, T_gt                  nodes/packet=%.1f leaf-batches/packet=%.1f" % census(0, T_gt)[:2])
