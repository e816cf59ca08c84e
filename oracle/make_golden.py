#!/usr/bin/env python3
"""Regenerate tests/golden/reference_tests.json from the reference checkout.

Run HERE (the container that has /root/reference); the GPU box only sees the
committed JSON.  Inputs are produced by the reference's own deterministic
generator (unit_test::Raw, compiled into oracle/_ref/libref_raw.so); expected
values are the literal golden vectors of the reference's googletest files,
parsed out of the .cpp sources (cited per entry).
"""
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as orc  # noqa: E402

REF = os.environ.get("CUPOCH_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "reference_tests.json")


def test_body(path, suite, case):
    src = open(os.path.join(REF, path)).read()
    m = re.search(r"TEST\(\s*%s\s*,\s*%s\s*\)\s*\{" % (suite, case), src)
    assert m, (path, suite, case)
    i, depth = m.end(), 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def brace_list(body, name):
    m = re.search(r"%s\s*\[\s*\]\s*=\s*\{([^}]*)\}" % name, body)
    return [float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]


def vec3_pushes(body, var):
    pat = r"%s\.push_back\(Vector3f\(([^)]*)\)\)" % var
    return [[float(v) for v in m.split(",")] for m in re.findall(pat, body)]


def main():
    orc.build()
    g = {}
    pts100 = orc.ref_rand_vec3f(100, [0, 0, 0], [10, 10, 10], 0)

    b = test_body("src/tests/knn/kdtree_flann.cpp", "KDTreeFlann", "SearchKNN")
    g["kdtree_search_knn"] = {
        "cite": "src/tests/knn/kdtree_flann.cpp:47-91",
        "points": pts100.tolist(), "query": [1.647059, 4.392157, 8.784314], "knn": 30,
        "ref_indices": [int(x) for x in brace_list(b, "indices0")],
        "ref_distance2": brace_list(b, "distances0"), "ref_return": 30, "tol": 1e-4}

    b = test_body("src/tests/knn/kdtree_flann.cpp", "KDTreeFlann", "SearchRadius")
    g["kdtree_search_radius"] = {
        "cite": "src/tests/knn/kdtree_flann.cpp:93-135",
        "points": pts100.tolist(), "query": [1.647059, 4.392157, 8.784314],
        "radius": 5.0, "max_nn": 15,
        "ref_indices": [int(x) for x in brace_list(b, "indices0")],
        "ref_distance2": brace_list(b, "distances0"), "ref_return": 15, "tol": 1e-4}

    b = test_body("src/tests/knn/lbvh_knn.cpp", "LinearBoundingVolumeHierarchyKNN", "SearchKNN")
    g["lbvh_search_nn"] = {
        "cite": "src/tests/knn/lbvh_knn.cpp:47-86",
        "points": pts100.tolist(), "query": [1.647059, 4.392157, 8.784314],
        "ref_index": int(brace_list(b, "indices0")[0]),
        "ref_distance2": brace_list(b, "distances0")[0], "ref_return": 1, "tol": 1e-9}

    rad = np.float32(30.0) / np.float32(180.0) * np.pi
    c, s = float(np.cos(np.float32(rad))), float(np.sin(np.float32(rad)))
    g["kabsch"] = {
        "cite": "src/tests/registration/kabsch.cpp:35-55",
        "points": orc.ref_rand_vec3f(20, [0, 0, 0], [1000, 1000, 1000], 0).tolist(),
        "ref_tf": [[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
        "tol_rel": 1e-3}

    c4, s4 = float(np.cos(np.pi / 4)), float(np.sin(np.pi / 4))
    g["pointcloud_transform"] = {
        "cite": "src/tests/geometry/pointcloud.cpp:143-174",
        "points": orc.ref_rand_vec3f(10, [0, 0, 0], [1000, 1000, 1000], 0).tolist(),
        "normals": orc.ref_rand_vec3f(10, [0, 0, 0], [1000, 1000, 1000], 0).tolist(),
        "transformation": [[1, 0, 0, 1], [0, c4, -s4, 2], [0, s4, c4, 3], [0, 0, 0, 1]],
        "tol": 5e-4}

    b = test_body("src/tests/geometry/pointcloud.cpp", "PointCloud", "VoxelDownSample")
    g["voxel_down_sample"] = {
        "cite": "src/tests/geometry/pointcloud.cpp:371-469",
        "points": orc.ref_rand_vec3f(20, [0, 0, 0], [1000, 1000, 1000], 0).tolist(),
        "normals": orc.ref_rand_vec3f(20, [0, 0, 0], [10, 10, 10], 0).tolist(),
        "colors": orc.ref_rand_vec3f(20, [0, 0, 0], [255, 255, 255], 0).tolist(),
        "voxel_size": 0.5,
        "ref_points": vec3_pushes(b, "ref_points"),
        "ref_normals": vec3_pushes(b, "ref_normals"),
        "ref_colors": vec3_pushes(b, "ref_colors"), "tol": 1e-4}

    b = test_body("src/tests/geometry/pointcloud.cpp", "PointCloud", "EstimateNormals")
    g["estimate_normals"] = {
        "cite": "src/tests/geometry/pointcloud.cpp:535-597",
        "points": orc.ref_rand_vec3f(40, [0, 0, 0], [1000, 1000, 1000], 0).tolist(),
        "knn": 30, "ref_normals": vec3_pushes(b, "ref"), "tol": 1e-4}

    def vec6_rows(body, var):
        pat = r"%s\[\d+\]\s*<<([^;]*);" % var
        return [[float(v) for v in m.replace("\n", " ").split(",")] for m in re.findall(pat, body)]

    # the two Jacobian functors of the RGB-D odometry: inputs as the tests build them
    # (odometry_tools.cpp GenerateImage / ShiftLeft / ShiftUp over unit_test::Rand)
    img = lambda vmin, vmax, seed: orc.ref_rand_floats(100, vmin, vmax, seed).reshape(10, 10)
    shift_left = lambda a, s: np.stack([[a[h, (w + s) % 10] for w in range(10)] for h in range(10)]).astype(np.float32)
    shift_up = lambda a, s: np.stack([[a[(h + s) % 10, w] for w in range(10)] for h in range(10)]).astype(np.float32)
    for name, case, hybrid in (("odometry_jacobian_color", "RGBDOdometryJacobianFromColorTerm", 0),
                               ("odometry_jacobian_hybrid", "RGBDOdometryJacobianFromHybridTerm", 1)):
        path = "src/tests/odometry/rgbdodometry_jacobian_from_%s_term.cpp" % ("hybrid" if hybrid else "color")
        b = test_body(path, case, "ComputeJacobianAndResidual")
        tgt_color = shift_up(shift_left(img(0.0, 1.0, 1), 10), 5)
        dx_color, dy_color = shift_left(img(0.0, 1.0, 1), 10), shift_up(img(0.0, 1.0, 1), 5)
        tgt_depth = img(1.0, 2.0, 0)
        # (both tests pass the target depth image as the depth-gradient images)
        g[name] = {
            "cite": path + ":31-" + ("134" if hybrid else "112"), "hybrid": hybrid,
            "source_color": img(0.0, 1.0, 1).tolist(), "source_depth": img(0.0, 1.0, 0).tolist(),
            "target_color": tgt_color.tolist(), "target_depth": tgt_depth.tolist(),
            "dx_color": dx_color.tolist(), "dy_color": dy_color.tolist(),
            "source_xyz": orc.ref_rand_floats(300, 0.0, 1.0, 0).reshape(10, 10, 3).tolist(),
            "intrinsic": [[0.5, 0, 0.75], [0, 0.65, 0.35], [0, 0, 0]],
            "extrinsic": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 0]],
            "corresps": orc.ref_rand_vec4i(10, 0, 3, 0).tolist(),
            "ref_J_r": vec6_rows(b, "ref_J_r"), "ref_r": brace_list(b, "ref_r_raw"), "tol": 1e-4}

    for k, v in g.items():
        for name in ("ref_points", "ref_normals", "ref_colors", "ref_indices"):
            if name in v:
                assert len(v[name]) > 0, (k, name)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=0)
    print("wrote", os.path.normpath(OUT), {k: len(json.dumps(v)) for k, v in g.items()})


if __name__ == "__main__":
    main()
