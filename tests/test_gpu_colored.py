"""GPU parity for the colored-ICP row (registration/colored_icp.cu): colour
gradients, the two-row system, ComputeRMSE's plain sum, and
RegistrationColoredICP end to end, against the CPU oracle on identical inputs.
The reference ships no golden vectors for this estimator, so the oracle is pinned
only by its own analytic checks (tests/test_oracle_selfcheck.py)."""
import numpy as np
import pytest
import torch

from conftest import make_colored
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

COLORED = 4


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def scene(n, seed, planar=False, noise=0.0, subsample=None):
    tgt, col, T = make_colored(n, seed=seed, planar=planar)
    if planar:
        nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    else:
        nrm = orc.estimate_normals_knn(tgt, 20)
        nrm[nrm[:, 2] < 0] *= -1
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt)
    scol = col.copy()
    rng = np.random.default_rng(seed + 100)
    if subsample:
        p = rng.permutation(n)[:subsample]
        src, scol = np.ascontiguousarray(src[p]), np.ascontiguousarray(col[p])
    if noise:
        src = (src + rng.normal(0, noise, src.shape)).astype(np.float32)
    return src, scol, tgt, col, nrm, T


def load(eng, src, scol, tgt, col, nrm):
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    eng.set_target_colors(col)
    eng.set_source_colors(scol)


@pytest.mark.parametrize("n,planar", [(3000, False), (20000, False), (20000, True), (200000, False)])
def test_colour_gradients_match_oracle(eng, n, planar):
    scale_r = 6.0 * (20000.0 / n) ** 0.5
    src, scol, tgt, col, nrm, T = scene(n, seed=n % 97, planar=planar)
    load(eng, src, scol, tgt, col, nrm)
    g = eng.compute_color_gradients(scale_r, 30)
    og = orc.color_gradients(tgt, nrm, col, scale_r, 30)
    assert np.isfinite(g).all()
    # same neighbour sets, same fp32 formulas; only the summation order over the
    # neighbours differs (the oracle adds them nearest-first)
    ref = np.abs(og).max()
    err = np.abs(g - og).max(axis=1)
    assert np.quantile(err, 0.999) < 2e-3 * ref, (np.quantile(err, 0.999), ref)
    assert np.median(err) < 1e-5 * ref, (np.median(err), ref)
    # fewer than 4 neighbours <-> exactly zero on both sides
    assert np.array_equal((g == 0).all(1), (og == 0).all(1))


def test_gradient_is_zero_with_fewer_than_four_neighbours(eng):
    src, scol, tgt, col, nrm, T = scene(2000, seed=5)
    load(eng, src, scol, tgt, col, nrm)
    g = eng.compute_color_gradients(1e-3, 30)      # nobody has neighbours
    assert not g.any()
    g = eng.compute_color_gradients(50.0, 4)       # max_nn = 4: self + 3 others -> nn = 3 < 4
    assert not g.any()
    g = eng.compute_color_gradients(50.0, 5)       # nn = 4: defined
    assert np.abs(g).max() > 0
    assert np.allclose(g, orc.color_gradients(tgt, nrm, col, 50.0, 5), atol=2e-3 * np.abs(g).max())


@pytest.mark.parametrize("lam", [0.968, 0.5, 1.0, 0.0, 7.0])
def test_colored_system_and_rmse_match_oracle(eng, lam):
    src, scol, tgt, col, nrm, T = scene(20000, seed=11, noise=0.02, subsample=15000)
    load(eng, src, scol, tgt, col, nrm)
    g = eng.compute_color_gradients(6.0, 30)
    eng.set_lambda_geometric(lam)
    idx = eng.search_radius_1nn(3.0)[0]
    cor = np.stack([np.arange(len(src)), idx], 1)[idx >= 0].astype(np.int32)
    assert len(cor) > 14000
    eng.set_correspondences(cor)
    sys_g = eng.compute_system(COLORED)
    rmse_g = eng.compute_rmse(COLORED)
    # the oracle gets the engine's gradients so that only the estimator is compared
    orc.set_colored_context(scol, col, g, lam)
    sys_o = orc.compute_system(COLORED, src, tgt, cor, tgt_nrm=nrm)
    rmse_o = orc.compute_rmse(COLORED, src, tgt, cor, tgt_nrm=nrm)
    scale = np.abs(sys_o[:27]).max()
    assert np.abs(sys_g[:30] - sys_o[:30]).max() <= 1e-9 * max(scale, 1.0), np.abs(sys_g[:30] - sys_o[:30]).max()
    assert sys_g[29] == len(cor)
    assert abs(rmse_g - rmse_o) <= 1e-5 * max(rmse_o, 1e-12)
    Tg = eng.compute_transformation(COLORED, det_thresh=-1.0)
    _, To = orc.solve_system(sys_o, -1.0)
    assert np.linalg.norm(Tg - To) <= 1e-5 * max(1.0, np.linalg.norm(To))


@pytest.mark.parametrize("planar,noise,sub", [(False, 0.0, None), (False, 0.02, 15000), (True, 0.0, None),
                                              (True, 0.01, 12000)])
def test_registration_colored_icp_matches_oracle(eng, planar, noise, sub):
    src, scol, tgt, col, nrm, T = scene(20000, seed=21, planar=planar, noise=noise, subsample=sub)
    load(eng, src, scol, tgt, col, nrm)
    res = eng.registration_colored_icp(3.0, None, det_thresh=-1.0)
    Tg = np.array(res.transformation, np.float32).reshape(4, 4).T
    ores = orc.registration_colored_icp(src, tgt, 3.0, scol, col, nrm, det_thresh=-1.0)
    motion = np.linalg.norm(T - np.eye(4))
    # both recover the motion ...
    assert np.linalg.norm(Tg - T) < 0.05 * motion
    # ... and agree with each other to the path's tolerance, relative to the scene's scale (100)
    assert np.linalg.norm(Tg - ores.transformation) <= 1e-5 * 100.0, np.linalg.norm(Tg - ores.transformation)
    assert res.iterations == ores.iterations
    assert abs(res.fitness - ores.fitness) <= 1e-6
    # coordinates are ~100, so 4 ulp of a coordinate (3e-5) is the floor of a residual's accuracy
    assert abs(res.inlier_rmse - ores.inlier_rmse) <= max(1e-5 * ores.inlier_rmse, 3e-5)


def test_colored_icp_without_colours_or_normals_returns_init(eng):
    src, scol, tgt, col, nrm, T = scene(5000, seed=2)
    init = np.eye(4, dtype=np.float32)
    init[:3, 3] = [0.1, 0.0, -0.1]
    # no source colours
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    eng.set_target_colors(col)
    res = eng.registration_colored_icp(3.0, init)
    assert np.array_equal(np.array(res.transformation, np.float32).reshape(4, 4).T, init)
    assert res.fitness > 0.9       # the correspondences are still evaluated
    # no target colours
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    eng.set_source_colors(scol)
    res = eng.registration_colored_icp(3.0, init)
    assert np.array_equal(np.array(res.transformation, np.float32).reshape(4, 4).T, init)
    # colours without target normals are refused loudly
    eng.set_target(tgt)
    from cupoch_amd.engine import MiIcpError
    with pytest.raises(MiIcpError):
        eng.set_target_colors(col)


def test_python_mirror_registration_colored_icp():
    from cupoch_amd import geometry, registration, utility
    src, scol, tgt, col, nrm, T = scene(20000, seed=33, planar=True)
    s, t = geometry.PointCloud(), geometry.PointCloud()
    s.points, s.colors = utility.Vector3fVector(src), utility.Vector3fVector(scol)
    t.points, t.colors = utility.Vector3fVector(tgt), utility.Vector3fVector(col)
    t.normals = utility.Vector3fVector(nrm)
    res = registration.registration_colored_icp(s, t, 3.0)          # reference defaults
    ores = orc.registration_colored_icp(src, tgt, 3.0, scol, col, nrm, det_thresh=1e-6)
    assert np.linalg.norm(res.transformation - ores.transformation) <= 1e-3
    assert np.linalg.norm(res.transformation - T) < 0.05 * np.linalg.norm(T - np.eye(4))
    pl = registration.registration_icp(s, t, 3.0, np.eye(4, dtype=np.float32),
                                       registration.TransformationEstimationPointToPlane(-1.0))
    assert np.linalg.norm(pl.transformation - T) > 10 * np.linalg.norm(res.transformation - T)
    assert len(res.correspondence_set) == 20000


def test_colored_icp_is_deterministic_and_survives_the_source_resort(eng):
    # 200k points: the loop re-sorts the source by match after the first pass, so the
    # staged intensities have to follow the permutation
    src, scol, tgt, col, nrm, T = scene(200000, seed=8, noise=0.005)
    out = []
    for _ in range(2):
        load(eng, src, scol, tgt, col, nrm)
        res = eng.registration_colored_icp(1.0, None, det_thresh=-1.0, max_iteration=10)
        out.append(np.array(res.transformation, np.float32))
    assert np.array_equal(out[0], out[1])
    Tg = out[0].reshape(4, 4).T
    assert np.linalg.norm(Tg - T) < 0.05 * np.linalg.norm(T - np.eye(4))
    ores = orc.registration_colored_icp(src, tgt, 1.0, scol, col, nrm, det_thresh=-1.0, max_iteration=10)
    assert np.linalg.norm(Tg - ores.transformation) <= 1e-5 * 100.0


def test_example_flow_on_the_real_coloured_fragment_matches_oracle():
    """examples/python/advanced/colored_pointcloud_registration.py, written against the
    Python mirror, on the reference's own RGB-D fragment (metre scale, where the fp32
    gradient fit of the reference is ill conditioned -- DESIGN.md section 6)."""
    from conftest import colored_fragment_pair
    from test_io_and_real_data import colored_example_flow_oracle
    from cupoch_amd import geometry, registration, utility
    src, scol, tgt, tcol, T = colored_fragment_pair()
    source, target = geometry.PointCloud(), geometry.PointCloud()
    source.points, source.colors = utility.Vector3fVector(src), utility.Vector3fVector(scol)
    target.points, target.colors = utility.Vector3fVector(tgt), utility.Vector3fVector(tcol)
    scales = [(0.04, 50), (0.02, 30)]
    cur = np.eye(4, dtype=np.float32)
    for radius, iters in scales:
        source_down = source.voxel_down_sample(radius)
        target_down = target.voxel_down_sample(radius)
        source_down.estimate_normals(geometry.KDTreeSearchParamRadius(radius=radius * 2, max_nn=30))
        target_down.estimate_normals(geometry.KDTreeSearchParamRadius(radius=radius * 2, max_nn=30))
        res = registration.registration_colored_icp(
            source_down, target_down, radius, cur,
            registration.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=iters))
        cur = res.transformation
    ores = colored_example_flow_oracle(src, scol, tgt, tcol, scales)
    motion = np.linalg.norm(T - np.eye(4))
    err_g, err_o = np.linalg.norm(res.transformation - T), np.linalg.norm(ores.transformation - T)
    print("real fragment: |T - T_gt| engine %.3g oracle %.3g, engine vs oracle %.3g, fitness %.4f / %.4f"
          % (err_g, err_o, np.linalg.norm(res.transformation - ores.transformation), res.fitness, ores.fitness))
    assert err_g < 0.15 * motion and err_o < 0.15 * motion
    assert abs(res.fitness - ores.fitness) < 0.01
    # same accuracy class as the oracle; bitwise-level agreement is not defined here because the
    # reference's gradient arithmetic amplifies summation-order differences by ~1e4
    assert err_g < 2.0 * err_o + 1e-3
