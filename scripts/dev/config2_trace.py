"""BASELINE config 2 (1M -> VoxelDownSample(0.02) both -> point-to-plane, r = 0.04) call by call: where do the iterations' times go?"""
import ctypes as C, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
eng = Engine(0)
gpu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
src, tgt, nrm, T_gt, _ = synth(1_000_000)
d_src, d_tgt, d_nrm = gpu(src), gpu(tgt), gpu(nrm)
vt, vn, _ = eng.voxel_downsample(d_tgt, 0.02, d_nrm)
vs, _, _ = eng.voxel_downsample(d_src, 0.02)


def counters():
    out = (C.c_int32 * 4)()
    eng._chk(eng._L.mi_icp_debug_loop_counters(eng._ctx, out))
    return list(out)


def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


for rep in range(2):
    eng.set_target(vt, vn)
    eng.set_source(vs)
    eng.set_profiling(True)
    line = ["rep %d (lib %s):" % (rep, os.path.basename(os.environ.get("MI_ICP_LIB_PATH", "default")))]
    ms, r = t(lambda: eng.icp_begin(_lib.EST_POINT_TO_PLANE, 0.04, None, -1.0)); line.append("begin %.3f ms" % ms)
    ms, r = t(lambda: eng.icp_iterate(3)); line.append("iterate(3) %.3f" % ms)
    p0 = eng.get_profile()
    ms, r = t(lambda: eng.icp_iterate(30)); p1 = eng.get_profile()
    line.append("iterate(30) %.3f ms (nn %.3f, reduce %.3f by events) counters %s rmse %.4g" % (ms, p1["nn_ms"] - p0["nn_ms"], p1["reduce_ms"] - p0["reduce_ms"], counters(), r.inlier_rmse))
    for k in range(6):
        p0 = eng.get_profile()
        ms, r = t(lambda: eng.icp_iterate(1)); p1 = eng.get_profile()
        line.append("| 1: %.3f (nn %.3f)" % (ms, p1["nn_ms"] - p0["nn_ms"]))
    eng.set_profiling(False)
    print(" ".join(line), flush=True)
