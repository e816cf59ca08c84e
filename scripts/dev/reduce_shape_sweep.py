"""The point-to-plane reduction's launch shape by shard size: grid x elements in flight, on 1/N Morton shards of the
bench's 10M source (needs the MI_ICP_AB_REDUCE_GRID / _U knobs of commit e056bb3..: a -DMI_AB_REDUCE_SWEEP build, taken out
again once profiles/r06_reduce_shape_sweep.txt was measured).  Prints reduce_ms (HIP events)
and the step (wall clock / iterations, no events)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
shapes = [(None, None)] + [(g, u) for u in (2, 4, 8) for g in (256, 384, 512, 768, 1024, 1536, 2048)]
for world in (8, 4, 2, 1):
    mine = D.device_shard_source(eng, d_all, 0, world)
    eng.set_source(torch.from_numpy(np.ascontiguousarray(src[mine])).cuda())
    eng.set_global_source_count(n)
    eng.set_profiling(False)
    os.environ.pop("MI_ICP_AB_REDUCE_GRID", None); os.environ.pop("MI_ICP_AB_REDUCE_U", None)
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(48)
    rows = []
    for g, u in shapes:
        if g is None:
            os.environ.pop("MI_ICP_AB_REDUCE_GRID", None); os.environ.pop("MI_ICP_AB_REDUCE_U", None)
        else:
            os.environ["MI_ICP_AB_REDUCE_GRID"] = str(g); os.environ["MI_ICP_AB_REDUCE_U"] = str(u)
        eng.icp_iterate(8)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.icp_iterate(32)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 32)
        eng.set_profiling(True)
        p0 = eng.get_profile(); eng.icp_iterate(16); p1 = eng.get_profile()
        eng.set_profiling(False)
        red = (p1["reduce_ms"] - p0["reduce_ms"]) / max(1, p1["reduce_launches"] - p0["reduce_launches"])
        rows.append((g, u, best * 1e3, red))
    base = rows[0]
    print("ranks %d (%d points): as shipped step %.4f ms reduce %.4f ms" % (world, len(mine), base[2], base[3]))
    for g, u, st, red in sorted(rows[1:], key=lambda r: r[2])[:8]:
        print("    grid %5d x %d in flight: step %.4f ms  reduce %.4f ms" % (g, u, st, red))
    sys.stdout.flush()
