"""The GPU's half of tests/test_outside_checks.py: the HIP engine against OUTSIDE implementations (LAPACK through
numpy, scipy's cKDTree, oracle/icp_numpy.py) instead of against the author's own CPU oracle.

  eigen3.h on the device (the GPU's acosf / cosf / sqrtf / divisions)   numpy.linalg.eigh
  the registration loop through the C ABI, three estimators            oracle/icp_numpy.py (cKDTree + fp64 + LAPACK)
  the colour-gradient kernel                                           numpy.linalg.lstsq per point
"""
import numpy as np
import pytest

from conftest import make_pair, make_colored
from oracle import icp_numpy as inp
from test_outside_checks import check_eigen_against_lapack, engine_eig, symmetric_cases

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_fast_eigen3x3_on_the_device_against_lapack():
    assert check_eigen_against_lapack(lambda A: engine_eig(A, device=0)[:2], "engine (device)") >= 10000
    # device against host code of the same functions: libm differs in the last ulp of acosf / cosf, which the closed
    # form amplifies at repeated eigenvalues exactly as it amplifies its own roundings -- same bounds as above
    for name, (A32, sep) in symmetric_cases(np.random.default_rng(9), 500).items():
        eh, _, Sh = engine_eig(A32)
        ed, _, Sd = engine_eig(A32, device=0)
        unit = np.maximum(np.abs(eh).max(1), 1e-30)
        assert (np.abs(eh - ed).max(1) <= (5e-6 if sep else 4e-4) * unit).all(), name
        assert np.array_equal(Sh, Sd, equal_nan=True), name        # gicp_weight: divisions only, correctly rounded


def _engine_run(eng, d, est, max_iteration=30, rel=1e-6, covs=None):
    eng.set_target(d["tgt"], d["tgt_nrm"] if est == inp.PT2PL else None, None if covs is None else covs[1])
    eng.set_source(d["src"], None, None if covs is None else covs[0])
    res = eng.registration_icp(est, d["max_dist"], None, rel, rel, max_iteration, -1.0)
    T = np.array(res.transformation, F32).reshape(4, 4).T
    return res, T, eng.get_correspondences()


@pytest.mark.parametrize("est,n,noise", [(inp.P2P, 100000, 0.0), (inp.P2P, 100000, 0.1), (inp.PT2PL, 200000, 0.0),
                                         (inp.PT2PL, 200000, 0.1), (inp.GICP, 200000, 0.0), (inp.GICP, 200000, 0.1)])
def test_registration_loop_against_the_second_restatement(eng, est, n, noise):
    """final transform <= 1e-5 Frobenius (north_star's bound; the engine applies the composed transform to the pristine
    source where the reference -- and icp_numpy -- transform a copy incrementally: DESIGN section 6, deviation 1),
    equal iteration counts under the default criteria, correspondence sets equal up to near-ties"""
    d = make_pair(n, seed=42, noise=noise)
    covs = None
    kw = {}
    if est == inp.GICP:
        covs = (inp.covariances_from_normals(d["src_nrm"]), inp.covariances_from_normals(d["tgt_nrm"]))
        kw = dict(src_cov=covs[0], tgt_cov=covs[1])
    if est == inp.PT2PL:
        kw = dict(tgt_nrm=d["tgt_nrm"])
    res, T, cor = _engine_run(eng, d, est, covs=covs)
    ref = inp.registration_icp(d["src"], d["tgt"], d["max_dist"], est=est, det_thresh=-1.0, **kw)
    err = float(np.linalg.norm(T.astype(np.float64) - ref.transformation.astype(np.float64)))
    assert err <= 1e-5, err
    assert res.iterations == ref.iterations, (res.iterations, ref.iterations)
    assert abs(res.fitness - ref.fitness) <= 2e-5
    assert abs(res.inlier_rmse - ref.inlier_rmse) <= 1e-7 + 1e-3 * ref.inlier_rmse
    a, b = dict(cor.tolist()), dict(ref.correspondence_set.tolist())
    differing = [i for i in set(a) | set(b) if a.get(i) != b.get(i)]
    assert len(differing) <= 1e-4 * n, len(differing)
    src64 = d["src"].astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
    tgt64 = d["tgt"].astype(np.float64)
    r2 = float(d["max_dist"]) ** 2
    for i in differing:
        da = ((src64[i] - tgt64[a[i]]) ** 2).sum() if i in a else r2
        db = ((src64[i] - tgt64[b[i]]) ** 2).sum() if i in b else r2
        assert abs(da - db) <= 1e-5 * max(da, db), "source %d: %s vs %s is not a near-tie" % (i, a.get(i), b.get(i))


def test_colour_gradient_kernel_against_lstsq(eng):
    from scipy.spatial import cKDTree
    tgt, col, _ = make_colored(6000, seed=4, planar=False)
    nrm = np.asarray(eng.estimate_normals_knn(tgt, 20), F32)
    radius, max_nn = 6.0, 30
    eng.set_target(tgt, nrm)
    eng.set_source(tgt[:10])
    eng.set_target_colors(col)
    eng.set_source_colors(col[:10])
    g = np.asarray(eng.compute_color_gradients(radius, max_nn), np.float64)
    c = col.astype(F32)
    inten = (((c[:, 0] + c[:, 1]) + c[:, 2]).astype(np.float64) / 3.0).astype(F32).astype(np.float64)
    P, N = tgt.astype(np.float64), nrm.astype(np.float64)
    dist, idx = cKDTree(P).query(P, k=max_nn, distance_upper_bound=radius)
    worst, checked = 0.0, 0
    for i in range(0, len(P), 7):
        nb = idx[i][(idx[i] < len(P)) & (dist[i] ** 2 < radius * radius)][1:]
        if len(nb) < 4:
            assert (g[i] == 0).all()
            continue
        dd = P[nb] - P[i]
        v = dd - (dd @ N[i])[:, None] * N[i]
        nn = len(nb)
        A = np.concatenate([v, (nn - 1) * N[i][None, :], 1e-3 * np.eye(3)])
        b = np.concatenate([inten[nb] - inten[i], np.zeros(4)])
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        lam = np.linalg.eigvalsh(v.T @ v + (nn - 1) ** 2 * np.outer(N[i], N[i]))
        tol = 3e-7 * (lam[2] / lam[0]) * max(np.abs(ref).max(), 1e-4) + 1e-7
        worst = max(worst, np.abs(g[i] - ref).max() / tol)
        checked += 1
    assert checked > 500 and worst <= 1.0, worst


def test_symmetric_and_colored_systems_of_the_engine_against_numpy_rows(eng):
    """the two estimators the second loop does not run: the engine's 6x6 systems against rows written out in numpy fp64
    (transformation_estimation.cu:58-90; colored_icp.cu:150-216), correspondences from the engine's own search"""
    d = make_pair(30000, seed=8, noise=0.05)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"], d["src_nrm"])
    eng.search_radius_1nn(d["max_dist"])
    cor = eng.get_correspondences()
    got = eng.compute_system(3)
    J, r = inp.rows_symmetric(d["src"][cor[:, 0]], d["src_nrm"][cor[:, 0]], d["tgt"][cor[:, 1]], d["tgt_nrm"][cor[:, 1]])
    ref = inp.system_of_rows(J, r)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(got[:21], ref[:21]) <= 1e-6 and rel(got[21:27], ref[21:27]) <= 1e-5 and abs(got[27] - ref[27]) <= 1e-5 * ref[27]
    assert got[29] == len(cor)
    # colored: the engine's own gradients (held against lstsq above), intensities as colored_icp.cu:91 forms them
    tgt, col, T = make_colored(12000, seed=6, planar=False)
    nrm = np.asarray(eng.estimate_normals_knn(tgt, 20), F32)
    src = (tgt.astype(np.float64) @ np.linalg.inv(T.astype(np.float64))[:3, :3].T + np.linalg.inv(T.astype(np.float64))[:3, 3]).astype(F32)
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    eng.set_target_colors(col)
    eng.set_source_colors(col)
    grad = np.asarray(eng.compute_color_gradients(6.0, 30), F32)
    eng.search_radius_1nn(3.0)
    cor = eng.get_correspondences()
    got = eng.compute_system(4)
    c = col.astype(F32)
    inten = (((c[:, 0] + c[:, 1]) + c[:, 2]).astype(np.float64) / 3.0).astype(F32)
    J, r = inp.rows_colored(src[cor[:, 0]], tgt[cor[:, 1]], nrm[cor[:, 1]], inten[cor[:, 0]], inten[cor[:, 1]], grad[cor[:, 1]], 0.968)
    ref = inp.system_of_rows(J, r)
    assert rel(got[:21], ref[:21]) <= 2e-6 and rel(got[21:27], ref[21:27]) <= 2e-5 and abs(got[27] - ref[27]) <= 2e-5 * ref[27]


def test_non_finite_target_points_are_never_matched(eng):
    """NaN / inf points in the TARGET: the tree is built, nobody matches them, EstimateNormals and VoxelDownSample survive
    (a grid whose extent overflows: no voxels, down_sample.cu:186-189)"""
    rng = np.random.default_rng(10)
    for kind in ("nan", "inf", "one coordinate"):
        tgt = rng.random((200_000, 3), dtype=np.float32)
        if kind == "nan":
            tgt[1000:1020] = np.nan
        elif kind == "inf":
            tgt[1000:1020] = np.inf
        else:
            tgt[1000:1020, 1] = np.nan
        src = rng.random((50_000, 3), dtype=np.float32)
        eng.set_target(tgt)
        eng.set_source(src)
        for radius in (0.03, 3.0):
            idx, d2, st = eng.search_radius_1nn(radius)
            assert not np.isin(idx, np.arange(1000, 1020)).any() and np.isfinite(d2[idx >= 0]).all(), kind
        nrm = np.asarray(eng.estimate_normals_knn(tgt, 30))
        assert np.isfinite(nrm[np.isfinite(tgt).all(1)]).all(), kind
        p, _, _ = eng.voxel_downsample(tgt, 0.02)
        assert (len(p) == 0) if kind == "inf" else (len(p) > 1000), kind
