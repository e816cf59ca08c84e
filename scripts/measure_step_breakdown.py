#!/usr/bin/env python3
"""Where one rank's ~38 us per step go: device-clock stamps (s_memrealtime; csrc/loop.h "where an
iteration's time goes", mi_icp_debug_set_step_stamps) in the stamping instantiations of the seeded search and the
point-to-plane reduction, on the full 10M-point target with 1/N of the source (N = 1, 2, 4, 8 -- a rank's share of the
driver's scaling run, as scripts/measure_shard.py emulates it).  Spans, averaged over 60 iterations:
  boundary: step end -> the next search's first wave       search: first wave start -> last wave end
  boundary: search's last wave -> reduction's first block  reduction, streaming: first block -> last block's ticket
  rows totalled by the finishing block | the ranks' exchange (none here) | solve | state written
-> profiles/r05_shard_step_breakdown.txt"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd import distributed as D
from cupoch_amd.engine import Engine
n = 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
d_all = torch.from_numpy(src).cuda()
names = ["step end -> next search's first wave", "search: first wave start -> last wave end", "search's last wave -> reduction's first block",
         "reduction: first block start -> last ticket", "finishing block: rows totalled", "ranks' exchange", "solve (+ convergence test)", "state written"]
print("# us per iteration, mean over 60 iterations; device clock (s_memrealtime); full 10M target, 1/N of the 10M source; step = wall clock / iterations of the same window")
for world in (1, 2, 4, 8):
    mine = D.device_shard_source(eng, d_all, 0, world)
    eng.set_source(torch.from_numpy(np.ascontiguousarray(src[mine])).cuda())
    eng.set_global_source_count(n)
    eng.set_profiling(False)
    eng.set_step_stamps(False)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(48)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.icp_iterate(60)
    torch.cuda.synchronize(); plain = (time.perf_counter() - t0) / 60 * 1e6
    eng.set_step_stamps(True)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(24)
    a, tpu = eng.get_step_stamps()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.icp_iterate(60)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 60 * 1e6
    b, _ = eng.get_step_stamps()
    cnt = int(b[24] - a[24])
    spans = (b[16:24].astype(np.float64) - a[16:24].astype(np.float64)) / max(cnt, 1) / tpu
    print("N = %d: %d source points on this rank; step %.2f us with the stamps (%.2f us without); %d iterations counted; clock %.0f ticks/us"
          % (world, len(mine), wall, plain, cnt, tpu))
    for k in range(8):
        print("    %-52s %7.2f" % (names[k], spans[k]))
    print("    %-52s %7.2f   (host-side enqueue gaps between chunks of 8 iterations are inside the first span)" % ("sum of the spans", float(spans.sum())))
