"""The dense-grid VoxelDownSample (voxel_dense.h) against the general path and the oracle, then timed at 10M points.
usage: voxel_dense_check.py [quick]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd.engine import Engine
from oracle import oracle as orc

eng = Engine(0)


def both(pts, voxel, nrm=None, col=None):
    os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
    a = eng.voxel_downsample(pts, voxel, nrm, col)
    os.environ["MI_ICP_NO_DENSE_VOXEL"] = "1"
    b = eng.voxel_downsample(pts, voxel, nrm, col)
    os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
    return a, b


def cuda(x):
    return None if x is None else torch.from_numpy(x).cuda()


bad = 0
for n, voxel, scale in ((800000, 0.01, 1.0), (200000, 0.02, 1.0), (1000000, 0.02, 1.0), (400000, 0.05, 1.0), (3000000, 0.01, 1.0),
                        (500000, 0.03, (1.0, 0.3, 2.0)), (700000, 0.011, 1.0)):
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3), dtype=np.float32) * np.asarray(scale, np.float32) - np.float32(0.25)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    col = rng.random((n, 3), dtype=np.float32)
    a, b = both(cuda(pts), voxel, cuda(nrm), cuda(col))
    rp, rn, rc = orc.voxel_downsample(pts, voxel, nrm, col)
    ok_len = len(a[0]) == len(rp) == len(b[0])
    ap, an, ac = (x.cpu().numpy() for x in a)
    exact = ok_len and np.array_equal(ap, rp) and np.array_equal(ac, rc)
    nexact = ok_len and np.array_equal(an, rn)
    dn = float(np.abs(an - rn).max()) if ok_len else -1
    dold = float(np.abs(b[0].cpu().numpy() - rp).max()) if ok_len else -1
    print("n %d voxel %g: voxels %d / oracle %d / general %d; points+colours bit-equal to the oracle: %s, normals: %s (max diff %.2e); general path max diff %.2e"
          % (n, voxel, len(a[0]), len(rp), len(b[0]), exact, nexact, dn, dold), flush=True)
    bad += 0 if (exact and dn < 2e-5) else 1
    a1, _ = both(cuda(pts), voxel)
    if not (ok_len and np.array_equal(a1[0].cpu().numpy(), rp)):
        print("  points-only call differs"); bad += 1
    # host arrays
    os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
    h = eng.voxel_downsample(pts, voxel, None, col)
    if not (np.array_equal(np.asarray(h[0]), rp) and np.array_equal(np.asarray(h[2]), rc)):
        print("  host-array call differs"); bad += 1
# a crowded cloud: the skew flag sends it to the general path
rng = np.random.default_rng(5)
pts = np.concatenate([rng.random((600000, 3), dtype=np.float32) * np.float32(0.02), rng.random((2000, 3), dtype=np.float32)]).astype(np.float32)
a, b = both(cuda(pts), 0.01)
rp, _, _ = orc.voxel_downsample(pts, 0.01)
print("crowded: %d / %d voxels, max diff %.2e" % (len(a[0]), len(rp), float(np.abs(a[0].cpu().numpy() - rp).max())))
bad += 0 if len(a[0]) == len(rp) else 1
# run-to-run
pts = cuda(np.random.default_rng(1).random((2000000, 3), dtype=np.float32))
r0 = eng.voxel_downsample(pts, 0.01)[0].clone()
same = all(torch.equal(r0, eng.voxel_downsample(pts, 0.01)[0]) for _ in range(5))
print("run to run identical:", same)
bad += 0 if same else 1
print("FAILED" if bad else "ALL OK", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(1 if bad else 0)
for n, voxel in ((10_000_000, 0.01), (1_000_000, 0.02)):
    rng = np.random.default_rng(42)
    pts = cuda(rng.random((n, 3), dtype=np.float32))
    nrm = cuda(rng.standard_normal((n, 3)).astype(np.float32))
    for label, nn in (("points only", None), ("points + normals", nrm)):
        for env in (None, "1"):
            if env:
                os.environ["MI_ICP_NO_DENSE_VOXEL"] = env
            else:
                os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
            eng.voxel_downsample(pts, voxel, nn)
            torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                v = eng.voxel_downsample(pts, voxel, nn)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print("voxel %s [%s]: n %d voxel %g -> %d voxels, %.3f ms (min %.3f)" % (label, "general" if env else "dense", n, voxel, len(v[0]), np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
os.environ.pop("MI_ICP_NO_DENSE_VOXEL", None)
sys.exit(1 if bad else 0)
