"""GPU: randomised differential test of the search and of the registration loop against
the oracle -- sizes, densities, radii, duplicates, initial transforms and iteration counts
drawn from a fixed-seed generator (the traversal has many data-dependent paths: bottom-up
start, early stop on disjoint boxes, leaf batches, overflowing cells, padded groups)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import check_nn, rigid

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def random_cloud(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:      # uniform volume
        p = rng.random((n, 3))
    elif kind == 1:    # surface with noise
        uv = rng.random((n, 2))
        p = np.stack([uv[:, 0], uv[:, 1], 0.2 * np.sin(5 * uv[:, 0]) * np.cos(4 * uv[:, 1])], 1)
        p += rng.normal(0, 0.002, p.shape)
    elif kind == 2:    # clusters of very different density
        c = rng.random((8, 3))
        s = 10.0 ** rng.uniform(-3, -0.7, 8)
        w = rng.integers(0, 8, n)
        p = c[w] + rng.normal(0, 1, (n, 3)) * s[w, None]
    else:              # lattice with exact duplicates
        g = max(2, int(round(n ** (1 / 3))))
        p = rng.integers(0, g, (n, 3)) / g
    if rng.random() < 0.3 and n > 20:          # some exact duplicates
        k = n // 10
        p[rng.integers(0, n, k)] = p[rng.integers(0, n, k)]
    return p.astype(np.float32)


# MI_ICP_FUZZ_CASES=500 turns this into a soak test
N_SEARCH = int(os.environ.get("MI_ICP_FUZZ_CASES", "24"))
N_REG = max(8, N_SEARCH // 3)


@pytest.mark.parametrize("case", range(N_SEARCH))
def test_search_matches_oracle_on_random_configurations(eng, case):
    rng = np.random.default_rng(1000 + case)
    nt = int(10 ** rng.uniform(0, 5.2))
    ns = int(10 ** rng.uniform(0, 4.8))
    tgt = random_cloud(rng, nt)
    if rng.random() < 0.5 and nt > 10:
        src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, 10.0 ** rng.uniform(-4, -1.5), (ns, 3))).astype(np.float32)
    else:
        src = random_cloud(rng, ns)
    ext = float(np.ptp(tgt, axis=0).max()) or 1.0
    eng.set_target(tgt)
    eng.set_source(src)
    T = rigid(rng.uniform(0, 0.2), rng.normal(size=3), rng.normal(0, 0.02, 3) * ext)
    src_t = orc.transform_points(T, src)
    # three searches on the same clouds: the 2nd and 3rd are seeded by the one before
    for radius in (ext * 10.0 ** rng.uniform(-2.5, -0.5), ext * 10.0 ** rng.uniform(-2.5, 0.5), ext * 3.0):
        idx, d2, st = eng.search_radius_1nn(radius, T)
        cnt, oi, od = orc.search_radius(tgt, src_t, radius, 1)
        assert st[0] == cnt, (case, nt, ns, radius)
        check_nn(idx, d2, oi, od, src_t, tgt)


@pytest.mark.parametrize("case", range(N_REG))
def test_registration_matches_oracle_on_random_configurations(eng, case):
    rng = np.random.default_rng(5000 + case)
    nt = int(10 ** rng.uniform(3.3, 5.0))
    tgt = random_cloud(rng, nt)
    ext = float(np.ptp(tgt, axis=0).max()) or 1.0
    nrm = rng.normal(size=(nt, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    spacing = ext * nt ** (-1 / 3)
    T = rigid(rng.uniform(0.0, 0.3) * spacing / ext, rng.normal(size=3), rng.normal(0, 0.3, 3) * spacing)
    take = rng.permutation(nt)[: max(10, int(nt * rng.uniform(0.3, 1.0)))]
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt[take])
    src = (src + rng.normal(0, 0.05 * spacing, src.shape)).astype(np.float32)
    est = [1, 2][case % 2]
    radius = spacing * rng.uniform(1.5, 4.0)
    iters = int(rng.integers(1, 25))
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    res = eng.registration_icp(est, radius, None, max_iteration=iters, det_thresh=-1.0)
    o = orc.registration_icp(src, tgt, radius, est=est, tgt_nrm=nrm, det_thresh=-1.0, max_iteration=iters)
    Tg = np.array(res.transformation, np.float32).reshape(4, 4).T
    # These inputs are deliberately nasty (clusters, duplicates, 30-100 % overlap, noise, radii of several
    # spacings, loops cut off after 1-24 iterations).  What can separate the engine from the reference's form:
    #  * the engine applies the composed T to the pristine source, the reference transforms its copy
    #    incrementally (DESIGN.md deviation 1): positions differ by ~1e-7, and a match that flips between two
    #    near-equidistant targets moves the solution of a few-thousand-point cloud by ~spacing / count.
    # So EVERY case is held (a) to the oracle's restatement of the engine's OWN form (composed=True: same
    # positions bit for bit, hence the same matches) at 1e-6 -- rounding level, both estimators -- and (b) to the
    # reference's incremental form at north_star's 1e-5, or at 1e-4 if (a) holds and the two FORMS of the oracle
    # themselves are further apart than 1e-5 on this input.
    #  * exact DUPLICATES in the target are no exception (round 5; they were held to 1e-4 only): coincident
    #    points keep their original order in the engine's kd order (the cell sort is stable and the local index
    #    sits in the low bits of every split key: kd_refine.h), so "lowest slot among equal distances" picks the
    #    lowest ORIGINAL index among coincident points -- which is exactly the oracle's tie rule
    #    (icp_oracle.c kd_offer).  Point-to-point could not tell anyway (same position); point-to-plane gets the
    #    same normal.  (FLANN itself keeps the first point visited: SURVEY quirk, DESIGN section 6 "Ties".)
    scale = max(1.0, ext)
    has_dups = len(np.unique(tgt, axis=0)) < nt
    assert abs(res.fitness - o.fitness) <= 2e-3
    assert abs(res.inlier_rmse - o.inlier_rmse) <= 1e-3 * max(o.inlier_rmse, spacing)
    err_ref = float(np.linalg.norm(Tg - o.transformation))
    oc = orc.registration_icp(src, tgt, radius, est=est, tgt_nrm=nrm, det_thresh=-1.0, max_iteration=iters,
                              composed=True)
    err_own = float(np.linalg.norm(Tg - oc.transformation))
    forms = float(np.linalg.norm(oc.transformation - o.transformation))
    assert err_own <= 1e-6 * scale, (case, est, err_own, has_dups)
    assert res.iterations == oc.iterations
    if forms <= 1e-5 * scale:
        rule = "1e-6 vs the composed form, 1e-5 vs the reference form"
        assert err_ref <= 1e-5 * scale, (case, est, err_ref, forms)
    else:
        rule = "1e-6 vs the composed form; the oracle's two forms are %.2g apart here: 1e-4 vs the reference form" % forms
        assert err_ref <= 1e-4 * scale, (case, est, err_ref, forms)
    if has_dups:
        rule += " (target with exact duplicates)"
    _RULES.append({"case": case, "estimator": "point-to-point" if est == 1 else "point-to-plane", "target_points": nt,
                   "iterations": iters, "rule": rule, "err_vs_reference_form": err_ref, "err_vs_composed_form": err_own})
    if case == N_REG - 1:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "fuzz_registration_rules.json"), "w") as f:
                import json
                json.dump(_RULES, f, indent=1)
        except OSError:
            pass


_RULES = []


@pytest.mark.parametrize("case", range(N_REG))
def test_voxel_downsample_matches_oracle_on_random_configurations(eng, case):
    """Random cloud kinds (volume, noisy sheet, clusters of very different density, lattice with duplicates), sizes from
    one point to 400k, voxel sizes from 'every point alone' to 'one voxel', offsets far from the origin, with and
    without normals / colours: the grid's key width, the number of radix passes, the bits left to the means kernel and
    the run lengths all follow from these (geometry_kernels.h) -- same voxels, same order, same means as the oracle."""
    rng = np.random.default_rng(9000 + case)
    n = int(10 ** rng.uniform(0, 5.6))
    pts = random_cloud(rng, n)
    pts = (pts * np.float32(10.0 ** rng.uniform(-1, 1.5)) + rng.normal(0, 10.0 ** rng.uniform(-1, 2), 3).astype(np.float32)).astype(np.float32)
    ext = float(np.ptp(pts, axis=0).max()) or 1.0
    voxel = ext * 10.0 ** rng.uniform(-3.0, 0.5)
    if ext / voxel > 5e5:                       # (keep the oracle's and the engine's grid inside what float32 indices resolve)
        voxel = ext / 5e5
    nrm = rng.standard_normal((n, 3)).astype(np.float32) if rng.random() < 0.6 else None
    col = rng.random((n, 3), dtype=np.float32) if rng.random() < 0.5 else None
    p, nn, c = eng.voxel_downsample(pts, voxel, nrm, col)
    rp, rn, rc = orc.voxel_downsample(pts, voxel, nrm, col)
    assert len(p) == len(rp), (case, n, voxel, len(p), len(rp))
    scale = float(np.abs(pts).max()) or 1.0
    np.testing.assert_allclose(p, rp, atol=2e-6 * scale)
    if nrm is None:
        assert nn is None
    else:
        np.testing.assert_allclose(nn, rn, atol=4e-5)
    if col is None:
        assert c is None
    else:
        np.testing.assert_allclose(c, rc, atol=2e-6)


@pytest.mark.skipif(os.environ.get("MI_ICP_WAIT_LINKS") is not None, reason="this IS the forced run")
def test_the_same_cases_with_every_first_pass_from_its_own_seeds():
    """The loops normally get their target's halos on demand and start a first pass from the queries' own
    seeds only when halos exist and the source is large (launch_nn / loop_run); the library reads its A/B
    switches once per process, so the forced variant -- halos for every loop, own seeds at every size down
    to one point -- runs this file and the loop-level parity tests in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, MI_ICP_WAIT_LINKS="1", MI_ICP_COARSE_MIN="0")
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_gpu_fuzz.py"),
                          os.path.join(here, "test_gpu_seeded.py")], env=env, capture_output=True, text=True,
                         timeout=900, cwd=os.path.dirname(here))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout
