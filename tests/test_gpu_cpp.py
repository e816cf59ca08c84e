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
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
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
    # depth frames -> cloud pyramids -> kinfu::PoseEstimation recovers the camera motion
    assert r["kinfu_ok"] and r["kinfu_err"] < 2e-3, r["kinfu_err"]
    assert r["kinfu_points"] == [320 * 240, 160 * 120] and r["kinfu_normals"]
    assert r["depth_strided"] == 80 * 60 and r["depth_bad_empty"]
    assert "[PointCloud::CreateFromDepthImage] Unsupported image format." in out.stderr
    # RGB-D odometry recovers most of the motion (hybrid term better than the colour term alone)
    assert r["odometry_ok"] and r["odometry_hybrid_err"] < 0.15 * r["odometry_motion"], r
    assert r["odometry_color_err"] < 0.6 * r["odometry_motion"] and r["odometry_mismatch_fails"], r
    assert "[RGBDOdometry] Two RGBD pairs should be same in size." in out.stderr
