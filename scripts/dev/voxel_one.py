"""VoxelDownSample alone, for a kernel timeline: 10M points (voxel 0.01 -> 1.03M voxels), with and without normals."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
rng = np.random.default_rng(42)
pts = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
nrm = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda()
eng = Engine(0)
for label, nn in (("points only", None), ("points + normals", nrm)):
    eng.voxel_downsample(pts, voxel, nn)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        v = eng.voxel_downsample(pts, voxel, nn)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("voxel %s: n %d voxel %g -> %d voxels, %.3f ms (min %.3f)" % (label, n, voxel, len(v[0]), np.median(ts) * 1e3, min(ts) * 1e3))
