"""The reference benchmark's registration_icp call alone (fragment.pcd, source and target the same cloud turned by 30
degrees, init = the same 30 degrees, threshold 0.02, default criteria), a few times: for a kernel trace."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd import pybind as cph
pts = np.load(os.path.join(ROOT, "tests", "golden", "fragment_points.npz"))["points"]
a = np.deg2rad(30.0)
T30 = np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
pc2 = cph.geometry.PointCloud(pts)
pc2.transform(T30)
est = cph.registration.TransformationEstimationPointToPoint()
ts = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = cph.registration.registration_icp(pc2, pc2, 0.02, T30, est)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("registration_icp: %s ms; fitness %.4f rmse %.4g" % (" ".join("%.2f" % t for t in ts), res.fitness, res.inlier_rmse))
