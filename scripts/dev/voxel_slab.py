"""VoxelDownSample of 10M points in a slab (1 x 1 x 0.45: a 20-bit key) -- for MI_ICP_VOXEL_HB=10 against 11 (1024 buckets with runs of 8
points per tile against 2048 with runs of 4)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd.engine import Engine
n, voxel = 10_000_000, 0.01
rng = np.random.default_rng(42)
pts = torch.from_numpy((rng.random((n, 3), dtype=np.float32) * np.float32([1.0, 1.0, 0.45])).astype(np.float32)).cuda()
eng = Engine(0)
for _ in range(3):
    v = eng.voxel_downsample(pts, voxel)
torch.cuda.synchronize()
ts = []
for _ in range(9):
    t0 = time.perf_counter()
    v = eng.voxel_downsample(pts, voxel)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("slab hb=%s: %d voxels, path %d, %.3f ms (min %.3f)" % (os.environ.get("MI_ICP_VOXEL_HB", "auto"), len(v[0]), eng._L.mi_icp_debug_last_voxel_path(eng._ctx), np.median(ts) * 1e3, min(ts) * 1e3))
