"""Phase clocks of vx_scatter / vx_finish (a -DMI_VX_CLOCKS build: MI_ICP_LIB_PATH=cupoch_amd/lib/libmi_icp_vxclk.so)."""
import os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd.engine import Engine
from cupoch_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
with_normals = len(sys.argv) > 3
rng = np.random.default_rng(42)
pts = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
nrm = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda() if with_normals else None
eng = Engine(0)
for _ in range(3):
    eng.voxel_downsample(pts, voxel, nrm)
torch.cuda.synchronize()
L = C.CDLL(os.environ["MI_ICP_LIB_PATH"])
buf = np.zeros((2, 4096, 12), np.uint64)
assert L.mi_vx_clocks_dump(buf.ctypes.data_as(C.c_void_p)) == 0
names = ((0, ("start", "loaded", "ranked", "bin scan", "staged", "written", "end")),
         (1, ("start", "loaded", "ranked", "scanned", "staged", "chunk end", "means", "stored")))
for k, label in ((0, "vx_scatter"), (1, "vx_finish")):
    first, nm = names[k]
    t = buf[k].astype(np.int64)[:, first:first + len(nm)]
    t = t[(t > 0).all(axis=1)]
    t0 = t[:, 0].min()
    print("%s: %d work items, span %.1f us" % (label, len(t), (t[:, -1].max() - t0) / 100.0))
    d = np.diff(t, axis=1) / 100.0
    for j in range(len(nm) - 1):
        print("   %-10s -> %-10s mean %7.2f us  median %7.2f  max %7.2f" % (nm[j], nm[j + 1], d[:, j].mean(), np.median(d[:, j]), d[:, j].max()))
    tot = (t[:, -1] - t[:, 0]) / 100.0
    print("   whole item: mean %.2f us, median %.2f, max %.2f; start times: median %.1f, last %.1f us" % (tot.mean(), np.median(tot), tot.max(), np.median(t[:, 0] - t0) / 100.0, (t[:, 0].max() - t0) / 100.0))
