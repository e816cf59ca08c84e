"""VoxelDownSample 10M with and without normals: two libraries alternating in one process (same box).
usage: voxel_ab.py libA.so libB.so"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib
libs = sys.argv[1:3]
n, voxel = 10_000_000, 0.01
rng = np.random.default_rng(42)
pts = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
nrm = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda()
import subprocess, json
if len(sys.argv) > 3:      # child: one library
    from cupoch_amd.engine import Engine
    eng = Engine(0)
    out = {}
    for label, nn in (("points", None), ("normals", nrm)):
        for _ in range(3):
            eng.voxel_downsample(pts, voxel, nn)
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            eng.voxel_downsample(pts, voxel, nn)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[label] = round(float(np.median(ts)) * 1e3, 4)
    print(json.dumps(out))
    sys.exit(0)
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, MI_ICP_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, lib, lib, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(os.path.basename(lib), line[-1] if line else r.stderr[-300:], flush=True)
