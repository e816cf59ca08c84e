"""GPU: knn::KDTreeFlann as a search object (SearchKNN / SearchRadius with up to
knn::NUM_MAX_NN = 100 neighbours): the reference's own golden vectors (src/tests/knn/kdtree_flann.cpp:47-135)
and the oracle's exact k-NN on random data, through the C ABI, the Python class and
batches of queries."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_golden_search_knn_and_radius(golden):
    from cupoch_amd import geometry
    g = golden["kdtree_search_knn"]                    # TEST(KDTreeFlann, SearchKNN)
    pc = geometry.PointCloud(np.asarray(g["points"], np.float32))
    tree = geometry.KDTreeFlann(pc)
    k, idx, d2 = tree.search_knn_vector_3f(g["query"], g["knn"])
    assert k == g["ref_return"] == 30
    assert sorted(idx) == sorted(g["ref_indices"])     # what the reference test compares
    assert [int(i) for i in idx] == g["ref_indices"]   # and the order, too
    np.testing.assert_allclose(d2, g["ref_distance2"], atol=g["tol"])

    g = golden["kdtree_search_radius"]                 # TEST(KDTreeFlann, SearchRadius)
    tree.set_geometry(geometry.PointCloud(np.asarray(g["points"], np.float32)))
    k, idx, d2 = tree.search_radius_vector_3f(g["query"], g["radius"], g["max_nn"])
    assert k == g["ref_return"] == 15
    assert [int(i) for i in idx] == g["ref_indices"]
    np.testing.assert_allclose(d2, g["ref_distance2"], atol=g["tol"])
    k2, idx2, _ = tree.search_vector_3f(g["query"], geometry.KDTreeSearchParamRadius(g["radius"], g["max_nn"]))
    assert k2 == k and idx2 == idx
    # empty tree
    with pytest.raises(RuntimeError):
        geometry.KDTreeFlann().search_knn_vector_3f([0, 0, 0], 3)


def rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry):
    """distances bit-exact; indices equal except where equal distances allow a choice"""
    fin = np.isfinite(od)
    assert np.array_equal(np.isfinite(d2), fin)
    assert np.array_equal(d2[fin], od[fin])
    assert np.array_equal(idx < 0, oi < 0)
    bad = np.flatnonzero((idx != oi).any(axis=1))
    for r in bad:                                      # only ties may differ
        k = int(fin[r].sum())
        dd = tgt[idx[r, :k]] - qry[r]
        chk = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
        np.testing.assert_allclose(chk, od[r, :k], rtol=5e-7, err_msg=str(r))   # (numpy has no fma: last-ulp slack)
        # (a tie inside the row, or between its last entry and the first point left out: the
        # distances are the oracle's bit for bit either way, so this is a valid answer)
    assert len(bad) <= max(2, len(idx) // 200)


@pytest.mark.parametrize("nt,nq,k", [(1, 5, 3), (7, 100, 8), (100, 1, 30), (5000, 3000, 1), (5000, 3000, 17),
                                     (200000, 50000, 32), (200000, 50000, 20),
                                     (50, 40, 100), (3000, 2000, 33), (100000, 20000, 64), (100000, 20000, 100)])
def test_search_knn_matches_oracle(eng, nt, nq, k):
    rng = np.random.default_rng(nt * 31 + nq + k)
    tgt = rng.random((nt, 3), dtype=np.float32)
    qry = (rng.random((nq, 3), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
    eng.set_target(tgt)
    found, idx, d2 = eng.search_knn(qry, k)
    ret, oi, od = orc.search_knn(tgt, qry, k)
    assert found == ret == nq * min(k, nt)
    rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)
    # rows are ascending and padded at the end
    assert np.all(np.diff(np.where(np.isfinite(d2), d2, np.float32(3e38)), axis=1) >= 0)


@pytest.mark.parametrize("radius,max_nn", [(0.01, 10), (0.03, 32), (0.1, 5), (1e-4, 8), (0.05, 100), (0.03, 50)])
def test_search_radius_matches_oracle(eng, radius, max_nn):
    rng = np.random.default_rng(int(radius * 1e5) + max_nn)
    tgt = rng.random((120000, 3), dtype=np.float32)
    qry = tgt[rng.permutation(len(tgt))[:20000]] + rng.normal(0, 0.003, (20000, 3)).astype(np.float32)
    eng.set_target(tgt)
    found, idx, d2 = eng.search_knn(torch.from_numpy(qry).cuda(), max_nn, radius)     # device in, device out
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    ret, oi, od = orc.search_radius(tgt, qry, radius, max_nn)
    assert found == ret
    rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)
    assert (d2[np.isfinite(d2)] < np.float32(radius * radius)).all()                   # strict


def test_errors_and_limits(eng):
    from cupoch_amd.engine import MiIcpError
    eng.set_target(np.random.default_rng(0).random((100, 3), dtype=np.float32))
    with pytest.raises(MiIcpError):
        eng.search_knn(np.zeros((4, 3), np.float32), 101)                  # knn::NUM_MAX_NN = 100
    found, idx, d2 = eng.search_knn(np.zeros((0, 3), np.float32), 5)
    assert found == 0 and idx.shape == (0, 5)
    # duplicates in the target: ties come out ascending in index
    tgt = np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (40, 1))
    eng.set_target(tgt)
    found, idx, d2 = eng.search_knn(np.array([[0.5, 0.5, 0.6]], np.float32), 8)
    assert found == 8 and len(set(idx[0].tolist())) == 8 and np.all(d2[0] == d2[0, 0])
    # the longer lists (64 and 104 slots) place every entry at its rank by (distance, index): equal
    # distances everywhere, and two distance classes with the row ending inside the second
    tgt = np.concatenate([np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (90, 1)),
                          np.tile(np.array([[0.5, 0.5, 0.25]], np.float32), (90, 1))])
    eng.set_target(tgt)
    for k in (40, 64, 70, 100):
        found, idx, d2 = eng.search_knn(np.array([[0.5, 0.5, 0.6], [0.5, 0.5, 0.2]], np.float32), k)
        assert found == 2 * k
        for r, near in ((0, 0), (1, 90)):
            row = idx[r].tolist()
            assert len(set(row)) == k and np.all(np.diff(d2[r]) >= 0)
            first = [j for j in row if near <= j < near + 90]
            assert row[:len(first)] == first and len(first) == min(k, 90)       # the nearer class first ...
            assert first == sorted(first) and row[len(first):] == sorted(row[len(first):])   # ... ascending in index


def test_sparse_and_clustered_targets_stay_exact_and_fast(eng):
    """Nodes in a group's padded tail hold any number of real points and a radius search may
    find fewer than max_nn: neither may leave a query with an unbounded walk (the seeding in
    knn_search_kernel widens through the ancestors / keeps the radius bound).  Exactness on
    strongly non-uniform data, and a wall-clock guard: the unbounded walk takes seconds."""
    import time
    rng = np.random.default_rng(77)
    # 60 tight clusters of very different sizes + a sparse background
    sizes = rng.integers(1, 4000, 60)
    centres = rng.random((60, 3)).astype(np.float32)
    tgt = np.concatenate([c + rng.normal(0, 0.002, (s, 3)).astype(np.float32) for c, s in zip(centres, sizes)] +
                         [rng.random((300, 3), dtype=np.float32)])
    qry = np.concatenate([tgt[rng.permutation(len(tgt))[:20000]] + rng.normal(0, 0.001, (20000, 3)).astype(np.float32),
                          rng.random((5000, 3), dtype=np.float32) * 3 - 1])
    eng.set_target(tgt)
    for k in (1, 4, 9, 32):
        found, idx, d2 = eng.search_knn(qry, k)
        ret, oi, od = orc.search_knn(tgt, qry, k)
        assert found == ret
        rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)
    found, idx, d2 = eng.search_knn(qry, 32, 0.004)
    ret, oi, od = orc.search_radius(tgt, qry, 0.004, 32)
    assert found == ret
    rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)

    big = rng.random((2_000_000, 3), dtype=np.float32)
    q = torch.from_numpy(big[:400_000] + np.float32(1e-3)).cuda()
    eng.set_target(torch.from_numpy(big).cuda())
    for k, r in ((4, 0.0), (30, 0.0), (30, 0.004)):       # radius 0.004: ~0.5 neighbours per query
        eng.search_knn(q, k, r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.search_knn(q, k, r)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.5, (k, r)


def test_outliers_around_a_dense_cloud_walk_alone(eng):
    """A few points scattered far around a dense cloud have their k-th neighbours at the distance of the cloud: in
    the wave-uniform walk their cubes hold all of it and their packets were offered every point (EstimateNormals of
    2M + 1000 such points: 177 ms).  They leave their packets -- by their bounds (knn_walks_alone) or, where whole
    packets are like that, by the probe of the tree's upper levels (knn_packet_reaches_too_far) -- and walk alone
    with L2 pruning (knn_normals.h knn_solo_walk): same rows as the oracle's, in a time of the usual order."""
    import time
    rng = np.random.default_rng(123)
    dense = (rng.random((300_000, 3), dtype=np.float32) * np.float32(0.1)).astype(np.float32)
    far = (rng.random((400, 3), dtype=np.float32) * np.float32(100.0) - np.float32(50.0)).astype(np.float32)
    tgt = np.concatenate([dense, far])
    # queries: every outlier itself, points next to outliers, points inside the cloud, points in between
    qry = np.concatenate([far, far[:200] + np.float32(0.5), dense[:3000] + np.float32(1e-4),
                          (rng.random((500, 3), dtype=np.float32) * np.float32(4.0) - np.float32(2.0)).astype(np.float32)])
    eng.set_target(tgt)
    for k in (1, 8, 30, 70):
        found, idx, d2 = eng.search_knn(qry, k)
        ret, oi, od = orc.search_knn(tgt, qry, k)
        assert found == ret
        rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)
    found, idx, d2 = eng.search_knn(qry, 30, 20.0)                   # a radius that reaches the cloud from many outliers
    ret, oi, od = orc.search_radius(tgt, qry, 20.0, 30)
    assert found == ret
    rows_equal_up_to_ties(idx, d2, oi, od, tgt, qry)
    # EstimateNormals of the whole cloud (queries = its own points), against the neighbour sets just verified
    nrm = eng.estimate_normals_knn(tgt, 30)
    assert np.isfinite(nrm).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-4)
    big = np.concatenate([(rng.random((2_000_000, 3), dtype=np.float32) * np.float32(0.1)).astype(np.float32),
                          (rng.random((1000, 3), dtype=np.float32) * np.float32(100.0)).astype(np.float32)])
    d = torch.from_numpy(big).cuda()
    eng.estimate_normals_knn(d, 30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.estimate_normals_knn(d, 30)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.05          # (3 ms; 177 ms with every lane in the wave's walk)
