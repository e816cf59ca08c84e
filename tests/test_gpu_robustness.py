"""GPU: the search/registration on inputs that are NOT a cloud against a moved copy of
itself -- partial overlap, outliers, clusters, lopsided sizes, large initial transforms,
context reuse -- always against the oracle on identical inputs."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_gpu_parity import check_nn, rigid

pytestmark = pytest.mark.gpu
P2P, PT2PL = 1, 2


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def T_of(res):
    return np.array(res.transformation, np.float32).reshape(4, 4).T


def clustered(n, seed):
    rng = np.random.default_rng(seed)
    centers = rng.random((12, 3)) * 4
    scale = rng.uniform(0.01, 0.3, 12)
    which = rng.integers(0, 12, n)
    pts = centers[which] + rng.standard_normal((n, 3)) * scale[which, None]
    pts[: n // 50] = pts[n // 50: 2 * (n // 50)]            # exact duplicates
    pts[-5:] = rng.random((5, 3)) * 1000                     # far outliers
    return pts.astype(np.float32)


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_search_on_clustered_data_with_duplicates_and_outliers(eng):
    tgt = clustered(150000, 1)
    src = clustered(90000, 2)
    eng.set_target(tgt)
    eng.set_source(src)
    for radius in (0.005, 0.05, 5.0):
        idx, d2, st = eng.search_radius_1nn(radius)
        cnt, oi, od = orc.search_radius(tgt, src, radius, 1)
        assert st[0] == cnt
        check_nn(idx, d2, oi, od, src, tgt)


def test_partial_overlap_and_lopsided_sizes(eng):
    rng = np.random.default_rng(3)
    tgt = rng.random((120000, 3), dtype=np.float32)
    nrm = unit(rng.standard_normal((120000, 3)))
    T = rigid(0.004, [1, 1, 0], [0.002, -0.001, 0.001])
    # source: half of the target's volume moved by T^-1, plus 30 % points with no partner at all
    keep = tgt[:, 0] < 0.5
    inside = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt[keep])
    lost = (rng.random((len(inside) * 3 // 10, 3), dtype=np.float32) + np.float32([2, 2, 2]))
    src = np.concatenate([inside, lost])[rng.permutation(len(inside) + len(lost))]
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    res = eng.registration_icp(PT2PL, 0.02, None, 1e-6, 1e-6, 30, -1.0)
    ref = orc.registration_icp(src, tgt, 0.02, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0)
    assert np.linalg.norm(T_of(res) - ref.transformation) <= 1e-5
    assert res.fitness == pytest.approx(ref.fitness, abs=1e-5) and 0.6 < res.fitness < 0.8
    np.testing.assert_array_equal(eng.get_correspondences(), ref.correspondence_set)
    # tiny target / big source and the other way round
    for ns, nt in ((200000, 3000), (3000, 200000)):
        a = rng.random((nt, 3), dtype=np.float32)
        b = rng.random((ns, 3), dtype=np.float32)
        eng.set_target(a)
        eng.set_source(b)
        r = 2.0 * nt ** (-1 / 3)
        idx, d2, st = eng.search_radius_1nn(r)
        cnt, oi, od = orc.search_radius(a, b, r, 1)
        assert st[0] == cnt
        check_nn(idx, d2, oi, od, b, a)


def test_large_initial_transform_and_context_reuse(eng):
    rng = np.random.default_rng(5)
    tgt = rng.random((80000, 3), dtype=np.float32)
    nrm = unit(rng.standard_normal((80000, 3)))
    T_true = rigid(2.1, [0.2, -1, 0.4], [3.0, -2.0, 0.5])      # 120 degrees + a big shift
    src = orc.transform_points(np.linalg.inv(T_true).astype(np.float32), tgt)[rng.permutation(80000)]
    init = (rigid(0.003, [0, 0, 1], [0.001, 0.001, -0.001]) @ T_true).astype(np.float32)
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    for est, kw in ((PT2PL, dict(tgt_nrm=nrm, det_thresh=-1.0)), (P2P, {})):
        res = eng.registration_icp(est, 0.03, init, 1e-6, 1e-6, 30, -1.0)     # same context, new call
        ref = orc.registration_icp(src, tgt, 0.03, init=init, est=est, **kw)
        assert np.linalg.norm(T_of(res) - ref.transformation) <= 1e-5
        assert np.linalg.norm(T_of(res) - T_true) < 1e-4 and res.fitness > 0.999
    # a second target on the same context, identity start that does NOT converge to anything useful
    eng.set_target(tgt[:30000] + np.float32(10.0))
    res = eng.registration_icp(P2P, 0.03, None, 1e-6, 1e-6, 5, -1.0)
    ref = orc.registration_icp(src, tgt[:30000] + np.float32(10.0), 0.03, est=orc.EST_P2P, max_iteration=5)
    assert res.fitness == ref.fitness == 0.0
    np.testing.assert_array_equal(T_of(res), np.eye(4))


def test_iteration_limits_and_immediate_convergence(eng):
    rng = np.random.default_rng(7)
    tgt = rng.random((40000, 3), dtype=np.float32)
    nrm = unit(rng.standard_normal((40000, 3)))
    T = rigid(0.004, [1, 0, 1], [0.001, 0.002, 0.0])
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt)
    eng.set_target(tgt, nrm)
    eng.set_source(src)
    for max_it, rel in ((0, 1e-6), (1, 1e-6), (3, 0.0), (30, 10.0)):
        res = eng.registration_icp(PT2PL, 0.04, None, rel, rel, max_it, -1.0)
        ref = orc.registration_icp(src, tgt, 0.04, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0,
                                   relative_fitness=rel, relative_rmse=rel, max_iteration=max_it)
        assert res.iterations == ref.iterations, (max_it, rel)
        assert res.nn_passes == ref.iterations + 1
        assert np.linalg.norm(T_of(res) - ref.transformation) <= 1e-5
    # stepping API == whole call
    eng.icp_begin(PT2PL, 0.04, None, -1.0)
    a = eng.icp_iterate(2)
    b = eng.icp_iterate(3)
    whole = eng.registration_icp(PT2PL, 0.04, None, 0.0, 0.0, 5, -1.0)
    assert a.iterations == 2 and b.iterations == 5 and whole.iterations == 5
    assert np.linalg.norm(T_of(b) - T_of(whole)) <= 1e-6


# ---- the kd-cell layout of the target (kd_cells.h): group boundaries, overflowing cells
@pytest.mark.parametrize("n", [3299, 3300, 3301, 4095, 4096, 4097, 6600, 6601, 8191, 8193, 9900, 9901, 13200, 13201, 19801, 26401,
                               39600, 39601, 52801, 79201, 105601])
def test_search_around_cell_and_group_boundaries(eng, n):
    # 3300 is the mean cell fill beyond which the layout takes more cells -- 2^d or 3 * 2^k of them, whichever is fewer
    # (kd_cells.h cell_layout_for: 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48 cells change over at these sizes); 4096 the group size
    rng = np.random.default_rng(n)
    tgt = rng.random((n, 3), dtype=np.float32)
    src = (tgt[rng.permutation(n)[: max(1, n // 2)]] + rng.normal(0, 0.01, (max(1, n // 2), 3))).astype(np.float32)
    eng.set_target(tgt)
    eng.set_source(src)
    for radius in (0.02, 2.0):
        idx, d2, st = eng.search_radius_1nn(radius)
        cnt, oi, od = orc.search_radius(tgt, src, radius, 1)
        assert st[0] == cnt
        check_nn(idx, d2, oi, od, src, tgt)


def test_cells_that_overflow_a_group(eng):
    """20 000 copies of ONE point cannot be split by any plane: their cell takes five
    4096-slot groups and shifts the groups behind it against the kd hierarchy.  Speed
    may suffer, answers may not."""
    rng = np.random.default_rng(77)
    base = rng.random((60000, 3), dtype=np.float32)
    dup = np.tile(np.array([[0.31, 0.62, 0.47]], np.float32), (20000, 1))
    line = np.stack([np.full(9000, 0.8, np.float32), np.full(9000, 0.1, np.float32),
                     np.linspace(0, 1, 9000, dtype=np.float32) // 0.25 * 0.25], 1)   # 4 distinct points x 2250
    tgt = np.concatenate([base[:30000], dup, base[30000:], line]).astype(np.float32)
    src = np.concatenate([rng.random((30000, 3), dtype=np.float32),
                          dup[:100] + rng.normal(0, 1e-3, (100, 3)).astype(np.float32),
                          line[::50] + np.float32(1e-4)]).astype(np.float32)
    eng.set_target(tgt)
    eng.set_source(src)
    for radius in (0.01, 0.2):
        idx, d2, st = eng.search_radius_1nn(radius)
        cnt, oi, od = orc.search_radius(tgt, src, radius, 1)
        assert st[0] == cnt
        # duplicates: the index may be ANY of the copies (same distance); d2 must be bit-exact
        assert np.array_equal(d2[oi[:, 0] >= 0], od[:, 0][oi[:, 0] >= 0])
        assert np.array_equal(idx < 0, oi[:, 0] < 0)
        hit = idx >= 0
        dd = src[hit] - tgt[idx[hit]]
        d_chk = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
        assert np.allclose(d_chk, d2[hit], rtol=1e-6, atol=0)
    # and a registration through it
    nrm = unit(np.random.default_rng(78).standard_normal((len(tgt), 3)))
    T = rigid(0.01, [1, 2, 3], [0.002, -0.001, 0.0015])
    src2 = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt[::3])
    eng.set_target(tgt, nrm)
    eng.set_source(src2)
    res = eng.registration_icp(PT2PL, 0.05, None, det_thresh=-1.0)
    ores = orc.registration_icp(src2, tgt, 0.05, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0)
    assert np.linalg.norm(T_of(res) - ores.transformation) <= 1e-5
    assert res.iterations == ores.iterations


def test_normals_and_correspondences_on_the_padded_layout(eng):
    """everything that indexes the target's sorted order has to skip the padding slots:
    EstimateNormals' queries, the correspondence export, explicit pairs (inverse map)."""
    rng = np.random.default_rng(5)
    n = 30000
    tgt = rng.random((n, 3), dtype=np.float32)
    nrm_g = np.asarray(eng.estimate_normals_knn(tgt, 12))
    nrm_o = orc.estimate_normals_knn(tgt, 12)
    agree = np.abs(np.sum(nrm_g * nrm_o, axis=1))
    assert (agree > 0.999).mean() > 0.995
    T = rigid(0.005, [0, 0, 1], [0.001, 0.001, -0.001])
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), tgt[rng.permutation(n)[:20000]])
    eng.set_target(tgt, nrm_o)
    eng.set_source(src)
    res = eng.evaluate_registration(0.01, T)
    cor = eng.get_correspondences()
    cnt, oi, _ = orc.search_radius(tgt, orc.transform_points(T, src), 0.01, 1)
    assert len(cor) == cnt == res.n_correspondences
    assert cor[:, 1].max() < n and cor[:, 1].min() >= 0
    assert np.array_equal(cor[:, 1], oi[:, 0][oi[:, 0] >= 0])
    eng.set_correspondences(cor[::2])
    sys_g = eng.compute_system(PT2PL, T)
    sys_o = orc.compute_system(PT2PL, orc.transform_points(T, src), tgt, cor[::2], tgt_nrm=nrm_o)
    assert np.abs(sys_g[:30] - sys_o[:30]).max() <= 1e-9 * max(np.abs(sys_o[:27]).max(), 1.0)


def test_estimate_normals_leaves_a_registration_in_flight_alone(eng):
    """EstimateNormals builds its tree in a private scratch context; target, source,
    correspondences and the stepping loop's state of the calling context survive it."""
    from conftest import make_pair
    d = make_pair(40000, seed=31, noise=0.05)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    eng.icp_begin(PT2PL, d["max_dist"], None, -1.0)
    eng.icp_iterate(2)
    other = np.random.default_rng(0).random((5000, 3), dtype=np.float32)
    n1 = eng.estimate_normals_knn(other, 10)
    n2 = eng.estimate_normals_radius(other, 0.2, 30)
    assert np.isfinite(n1).all() and np.isfinite(n2).all()
    res = eng.icp_iterate(4)
    cor = eng.get_correspondences()
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    eng.icp_begin(PT2PL, d["max_dist"], None, -1.0)
    ref = eng.icp_iterate(6)
    np.testing.assert_array_equal(np.array(res.transformation), np.array(ref.transformation))
    assert res.fitness == ref.fitness and res.iterations == ref.iterations == 6
    np.testing.assert_array_equal(cor, eng.get_correspondences())
    # a new cloud ends a stepping loop: iterate() is then a no-op on stale state
    eng.set_source(d["src"][:1000])
    out = eng.icp_iterate(3)
    assert out.iterations == 6 and eng.evaluate_registration(d["max_dist"]).n_correspondences <= 1000
    # search against an empty target forgets an explicit correspondence set
    eng.set_target(d["tgt"])
    eng.set_source(d["src"])
    eng.set_correspondences(np.array([[0, 1], [2, 3]], np.int32))
    eng.set_target(np.zeros((0, 3), np.float32))
    eng.search_radius_1nn(1.0)
    assert len(eng.get_correspondences()) == 0


def test_list_build_in_flight_survives_retargeting_knn_and_destroy():
    """A context that has registered before starts the halos of a target below 2M points on its private
    stream in mi_icp_set_target.  Whatever comes next while that build is in flight -- another target, a
    k-NN search, normals, a registration on a tiny source that does not wait for it, destroying the
    context -- must neither hang nor change results."""
    import numpy as np
    from cupoch_amd.engine import Engine
    from conftest import make_pair
    d = make_pair(300_000, seed=17, noise=0.02)
    small = make_pair(5_000, seed=18, noise=0.02)
    ref_eng = Engine(0)
    ref_eng.set_target(d["tgt"], d["tgt_nrm"])
    ref_eng.set_source(d["src"])
    ref = ref_eng.registration_icp(2, d["max_dist"], None, 0.0, 0.0, 6, -1.0)       # fresh context: no head start
    k_ref = ref_eng.search_knn(d["src"][:1000], 8)
    ref_eng.close()
    eng = Engine(0)
    eng.set_target(small["tgt"], small["tgt_nrm"])
    eng.set_source(small["src"])
    eng.registration_icp(2, small["max_dist"], None, 0.0, 0.0, 3, -1.0)              # now the context "has registered"
    for _ in range(3):
        eng.set_target(d["tgt"], d["tgt_nrm"])                                       # halos start ...
        eng.set_target(small["tgt"], small["tgt_nrm"])                               # ... and are dropped
        eng.set_target(d["tgt"], d["tgt_nrm"])
        k = eng.search_knn(d["src"][:1000], 8)                                       # k-NN next to the build
        assert k[0] == k_ref[0] and np.array_equal(k[1], k_ref[1]) and np.array_equal(k[2], k_ref[2])
        eng.set_source(d["src"])
        got = eng.registration_icp(2, d["max_dist"], None, 0.0, 0.0, 6, -1.0)
        assert np.array_equal(np.array(got.transformation), np.array(ref.transformation))
        assert got.fitness == ref.fitness and got.inlier_rmse == ref.inlier_rmse
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.close()                                                                      # destroyed with the build in flight
