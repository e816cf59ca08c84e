"""CPU-side checks of the drop-in boundary: the library builds/loads, exports
exactly what include/*.h declares, mirrors the reference's Python surface, and
the host solver (no GPU involved) agrees with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as orc


def _declared():
    names = []
    for h in ("mi_icp.h", "mi_icp_debug.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"MI_ICP_API\s+[\w\s\*]+?\b(mi_icp_\w+)\s*\(", src)
    return names


def test_library_builds_and_exports_every_declared_symbol():
    from cupoch_amd import _lib
    _lib.build()
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (mi_icp_\w+)", out))
    assert set(declared) == exported
    assert set(declared) == set(_lib.SIGNATURES)        # the ctypes table binds all of them
    assert b"gfx950" in lib.mi_icp_version()


def test_library_contains_gfx950_code_object_only():
    from cupoch_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + _lib.LIB_PATH], capture_output=True, text=True).stdout
    if out.strip():
        targets = [l for l in out.split() if "amdgcn" in l]
        assert targets and all("gfx950" in t for t in targets)


def test_no_device_is_a_status_not_a_crash():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cupoch_amd import _lib, MiIcpError
    from cupoch_amd.engine import Engine
    ctx = C.c_void_p()
    assert _lib.load().mi_icp_create(0, C.byref(ctx)) == -5      # MI_ICP_ERR_NO_DEVICE
    with pytest.raises(MiIcpError):
        Engine(0)                                                # fails loudly, no CPU fallback


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing shipped may import, link, dlopen
    or read it (comments may mention it)."""
    pkg = os.path.join(ROOT, "cupoch_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+oracle|liboracle|oracle/|"
                     r"icp_oracle|oracle\.py|dlopen\([^)]*oracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".txt", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), os.path.join(dirpath, f)


def test_python_surface_mirrors_reference_names_and_defaults():
    from cupoch_amd import registration as reg
    c = reg.ICPConvergenceCriteria()
    assert (c.relative_fitness, c.relative_rmse, c.max_iteration) == (1e-6, 1e-6, 30)
    assert reg.TransformationEstimationPointToPlane().det_thresh == pytest.approx(1e-6)
    assert reg.TransformationEstimationSymmetricMethod().det_thresh == pytest.approx(1e-6)
    assert reg.TransformationEstimationForGeneralizedICP().epsilon == pytest.approx(1e-3)
    T = reg.TransformationEstimationType
    assert [int(T.Unspecified), int(T.PointToPoint), int(T.PointToPlane), int(T.SymmetricMethod),
            int(T.ColoredICP), int(T.GeneralizedICP)] == [0, 1, 2, 3, 4, 5]
    assert reg.TransformationEstimationPointToPoint().get_transformation_estimation_type() == T.PointToPoint
    r = reg.RegistrationResult()
    np.testing.assert_array_equal(r.transformation, np.eye(4))
    assert r.fitness == 0.0 and r.inlier_rmse == 0.0 and r.correspondence_set.shape == (0, 2)
    for name in ("registration_icp", "evaluate_registration", "registration_generalized_icp"):
        assert callable(getattr(reg, name))


def test_host_solver_matches_oracle():
    from cupoch_amd import engine
    rng = np.random.default_rng(5)
    for trial in range(20):
        J = rng.standard_normal((300, 6)) * rng.uniform(0.1, 3.0, 6)
        r = rng.standard_normal(300) * 1e-2
        sys = np.zeros(32)
        sys[:21] = (J.T @ J)[np.triu_indices(6)]
        sys[21:27] = J.T @ r
        sys[29] = 300
        for det in (-1.0, 1e-6):
            ok1, T1 = engine.solve_system(sys, det)
            ok2, T2 = orc.solve_system(sys, det)
            assert ok1 == ok2
            np.testing.assert_allclose(T1, T2, atol=2e-6)
    ok, T = engine.solve_system(np.zeros(32), 1e-6)
    assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))
    big = sys.copy()
    big[:21] *= 1e7                                   # fp32 determinant overflow -> failure (quirk 6)
    assert engine.solve_system(big, 1e-6)[0] is False and engine.solve_system(big, -1.0)[0] is True
    x = np.array([0.3, -0.1, 0.2, 1, 2, 3], np.float32)
    np.testing.assert_allclose(engine.vector6_to_matrix4(x), orc.vector6_to_matrix4(x), atol=1e-7)
    np.testing.assert_array_equal(engine.vector6_to_matrix4([0, 0, 0, 1, 2, 3])[:3, :3], np.eye(3))


def test_host_kabsch_matches_oracle_and_golden(golden):
    from cupoch_amd import engine
    g = golden["kabsch"]
    src = np.asarray(g["points"], np.float32)
    ref = np.asarray(g["ref_tf"], np.float32)
    tgt = orc.transform_points(ref, src)
    cor = np.stack([np.arange(20), np.arange(20)], 1).astype(np.int32)
    sys = orc.compute_system(orc.EST_P2P, src, tgt, cor)
    T = engine.kabsch_from_sums(sys, 20)
    assert np.linalg.norm(T - ref) <= g["tol_rel"] * min(np.linalg.norm(T), np.linalg.norm(ref))
    np.testing.assert_allclose(T, orc.kabsch_from_sums(sys, 20), atol=1e-6)
    # the reference divides by model.size(), not by the number of pairs (kabsch.cu:76,107)
    np.testing.assert_allclose(engine.kabsch_from_sums(sys, 40), orc.kabsch_from_sums(sys, 40), atol=1e-6)


def test_every_environment_switch_is_in_the_design_table():
    """DESIGN.md section 7 lists every MI_ICP_* environment variable the library, the Python loader and bench.py read
    (A/B measurements and tests only: none changes results) -- a switch that is not there is clutter nobody can find."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    design = open(os.path.join(root, "DESIGN.md")).read()
    seen = set()
    for base, _, files in os.walk(os.path.join(root, "cupoch_amd")):
        if os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".h", ".hip", ".cpp", ".py")):
                src = open(os.path.join(base, f), errors="replace").read()
                seen |= set(re.findall(r'getenv\(\s*"(MI_ICP_[A-Z0-9_]+)"', src))
                seen |= set(re.findall(r'environ(?:\.get)?\(?\[?\s*"(MI_ICP_[A-Z0-9_]+)"', src))
    src = open(os.path.join(root, "bench.py")).read()
    seen |= set(re.findall(r'environ(?:\.get|\.setdefault)?\(?\[?\s*"(MI_ICP_[A-Z0-9_]+)"', src))
    assert len(seen) >= 20, sorted(seen)
    missing = sorted(v for v in seen if v not in design)
    assert not missing, missing
