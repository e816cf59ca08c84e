"""GPU: the C++ drop-in surface (namespace cupoch, cupoch_amd/cpp) compiled with
g++ against libcupoch_amd.so and run as a user program."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_surface_end_to_end(tmp_path):
    from cupoch_amd import _lib
    _lib.build()
    cpp = os.path.join(ROOT, "cupoch_amd", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp])
    exe = str(tmp_path / "test_registration")
    libdir = os.path.join(ROOT, "cupoch_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(cpp, "include"), "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "test_registration.cpp"),
                           "-o", exe, "-L" + libdir, "-lcupoch_amd", "-lmi_icp", "-L/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    # files the C++ readers must understand: written by the Python mirror
    import numpy as np
    from cupoch_amd.io import read_point_cloud_arrays, write_pcd_arrays, write_ply_arrays
    rng = np.random.default_rng(3)
    py_pts = rng.random((3000, 3), dtype=np.float32)
    py_nrm = rng.standard_normal((3000, 3)).astype(np.float32)
    write_pcd_arrays(str(tmp_path / "py_comp.pcd"), py_pts, py_nrm, None, compressed=True)
    write_ply_arrays(str(tmp_path / "py_bin.ply"), py_pts * 2, None, None)
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["roundtrip_max"] < 1e-6
    for k in ("p2p_err", "pt2pl_err", "sym_err", "gicp_err", "custom_err"):
        assert r[k] < 2e-5, (k, r[k])
    assert r["p2p_fitness"] > 0.999 and r["eval_fitness"] > 0.999 and r["p2p_ncorr"] > 49000
    assert r["p2p_corr_ascending"] and r["kabsch_ok"] and r["no_normals_is_identity"]
    assert r["custom_calls"] >= 1
    assert r["kdtree_knn_ok"] and r["kdtree_batch"] == 4 * 50000 and r["kdtree_empty"] == -1
    assert r["kdtree_radius"][0] == r["kdtree_radius"][1] == r["kdtree_radius"][2] > 0, r["kdtree_radius"]
    # colored ICP recovers the in-plane motion of a textured plane; point-to-plane cannot
    assert r["colored_err"] < 0.05 * r["colored_motion"], r
    assert r["colored_vs_plane_err"] > 10 * r["colored_err"], r
    assert r["colored_no_colors_is_identity"]
    assert 1000 < r["voxels"] <= 9261 and r["voxel_normals_unit"] and r["voxel_zero_empty"] and r["has_normals"]
    assert "require pre-computed target normal vectors" in out.stderr      # LogError path
    # GeometryBase3D virtuals on the cloud, called through the base class
    for k in ("base3d_bounds", "base3d_box", "base3d_translate", "base3d_translate_abs", "base3d_scale",
              "base3d_empty_center_zero"):
        assert r[k], k
    assert r["base3d_rotate_err"] < 1e-5 and r["base3d_rotate_normals_err"] < 1e-6, r
    # io::ReadPointCloud / WritePointCloud: every PCD / PLY flavour round-trips in C++ (binary flavours bit for bit),
    # C++ reads what Python wrote, Python reads what C++ wrote
    io = r["io"]
    assert io["all_ok"], io
    assert io["py_points"] == 3000 + 1 + 3000
    w = np.array([1.0, 2.0, 3.0])
    assert io["py_checksum"] == pytest.approx(float((py_pts.astype(np.float64) @ w).sum() * 3), rel=1e-9)
    assert io["nan_removed"] == 2 and io["nan_kept"] == 4 and io["unknown_ext_fails"]
    first = None
    for name in ("cpp_bin.pcd", "cpp_comp.pcd", "cpp_ascii.pcd", "cpp_bin.ply", "cpp_ascii.ply"):
        a = read_point_cloud_arrays(str(tmp_path / name))
        assert a["points"].shape == (5000, 3) and a["normals"].shape == (5000, 3) and a["colors"].shape == (5000, 3)
        if first is None:
            first = a
        tol = 1e-6 if "ascii.pcd" in name else 0
        np.testing.assert_allclose(a["points"], first["points"], rtol=tol)
        np.testing.assert_allclose(a["normals"], first["normals"], rtol=tol, atol=tol)
        np.testing.assert_allclose(a["colors"], first["colors"], atol=1e-6)
    assert "unknown file extension" in out.stderr
    # depth frames -> cloud pyramids -> kinfu::PoseEstimation recovers the camera motion
    assert r["kinfu_ok"] and r["kinfu_err"] < 2e-3, r["kinfu_err"]
    assert r["kinfu_points"] == [320 * 240, 160 * 120] and r["kinfu_normals"]
    assert r["depth_strided"] == 80 * 60 and r["depth_bad_empty"]
    assert "[PointCloud::CreateFromDepthImage] Unsupported image format." in out.stderr
    # RGB-D odometry recovers most of the motion (hybrid term better than the colour term alone)
    assert r["odometry_ok"] and r["odometry_hybrid_err"] < 0.15 * r["odometry_motion"], r
    assert r["odometry_color_err"] < 0.6 * r["odometry_motion"] and r["odometry_mismatch_fails"], r
    assert "[RGBDOdometry] Two RGBD pairs should be same in size." in out.stderr
