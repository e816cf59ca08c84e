"""The bench's transient (bench.py secondary) once, for a kernel timeline."""
import os, sys, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth
from cupoch_amd import _lib
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt, nrm, T_gt, max_dist = synth(n)
s = n ** (-1.0 / 3.0)
eng = Engine(0)
d_tgt, d_nrm = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
init = np.eye(4, dtype=np.float32)
init[:3, 3] = (1.5 * s / np.sqrt(3.0)) * np.array([1.0, -1.0, 1.0], np.float32)
ang = 0.5 * s
init[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], np.float32)
rng = np.random.default_rng(6)
noisy_all = (src + rng.normal(0.0, 0.15 * s, src.shape)).astype(np.float32)
d_noisy = torch.from_numpy(np.ascontiguousarray(noisy_all)).cuda()
for rep in range(2):
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_noisy)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, init, 0.0, 0.0, 30, -1.0)
    torch.cuda.synchronize(); print("loop %.3f ms fitness %.4f" % ((time.perf_counter() - t0) * 1e3, r.fitness))
