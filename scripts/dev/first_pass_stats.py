"""Census of the walk from the root on the bench's first pass (10M-vs-10M under the initial displacement)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
eng = Engine(0)
eng.set_target(torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda())
eng.set_source(torch.from_numpy(src).cuda())
out = (C.c_uint64 * 16)()
T = np.ascontiguousarray(np.eye(4, dtype=np.float32))
eng._chk(eng._L.mi_icp_debug_nn_stats8(eng._ctx, T.ctypes.data_as(C.c_void_p), float(max_dist), 0, out))
print("first pass from the root: records per packet %.2f, 64-item batches per packet %.2f, slowest packet %d" % (out[0] / out[2], out[1] / out[2], out[3]))
