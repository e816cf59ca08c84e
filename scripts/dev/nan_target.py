"""Does a TARGET with NaN / inf points survive set_target, the search, EstimateNormals and VoxelDownSample?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cupoch_amd.engine import Engine
kind = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
rng = np.random.default_rng(10)
tgt = rng.random((n, 3), dtype=np.float32)
src = rng.random((100_000, 3), dtype=np.float32)
if kind == "nan": tgt[1000:1020] = np.nan
if kind == "inf": tgt[1000:1020] = np.inf
if kind == "nan1": tgt[1000:1020, 1] = np.nan
eng = Engine(0)
eng.set_target(tgt); eng.synchronize(); print("target ok", flush=True)
eng.set_source(src); eng.synchronize()
idx, d2, st = eng.search_radius_1nn(0.02); print("search ok", st, flush=True)
idx, d2, st = eng.search_radius_1nn(3.0); print("wide search ok", st, flush=True)
bad = np.isin(idx, np.arange(1000, 1020)).sum(); print("matches onto non-finite points:", bad, flush=True)
nrm = eng.estimate_normals_knn(tgt, 30); print("normals ok, finite rows", int(np.isfinite(nrm).all(1).sum()), flush=True)
p, _, _ = eng.voxel_downsample(tgt, 0.02); print("voxel ok", len(p), flush=True)
