"""GPU: the SEEDED correspondence search (what every ICP iteration after the first runs) must
return exactly what an unseeded search and the oracle return, after the transform changed by a
lot or by an ulp, on uniform / clustered / surface data, before and after the loop's
match-order re-sort of the source.  Bit-exact d2; an index may differ only on an exact tie.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
P2P, PT2PL = 1, 2


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rigid(angle, axis, t):
    axis = np.asarray(axis, np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    T[:3, 3] = t
    return T.astype(np.float32)


def cloud(kind, n, rng):
    if kind == "uniform":
        return rng.random((n, 3), dtype=np.float32)
    if kind == "clustered":          # blobs whose densities differ by three orders of magnitude
        k = 12
        centres = rng.random((k, 3))
        sig = 10.0 ** rng.uniform(-3.5, -1.0, k)
        which = rng.integers(0, k, n)
        return (centres[which] + rng.standard_normal((n, 3)) * sig[which, None]).astype(np.float32)
    if kind == "surface":            # a thin sheet: leaves are flat, regions are slabs
        uv = rng.random((n, 2))
        z = 0.15 * np.sin(5 * uv[:, 0]) * np.cos(4 * uv[:, 1]) + rng.standard_normal(n) * 2e-4
        return np.stack([uv[:, 0], uv[:, 1], z], 1).astype(np.float32)
    raise ValueError(kind)


def compare(idx, d2, q, tgt, tree, radius):
    _, oi, od = tree.search_radius(q, radius, 1)
    oi, od = oi[:, 0], od[:, 0]
    assert np.array_equal(idx < 0, oi < 0), "hit/miss pattern differs"
    hit = oi >= 0
    assert np.array_equal(d2[hit], od[hit]), "d2 not bit-exact"
    diff = np.flatnonzero(idx != oi)
    if len(diff):
        dd = q[diff] - tgt[idx[diff]]
        alt = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
        assert np.array_equal(alt, od[diff]), "index mismatch that is not an exact tie"
    return int(hit.sum())


@pytest.mark.parametrize("kind", ["uniform", "clustered", "surface"])
@pytest.mark.parametrize("noise", [0.0, 0.2])
def test_seeded_search_equals_oracle_after_the_transform_changed(eng, kind, noise):
    n = 200_000
    rng = np.random.default_rng({"uniform": 10, "clustered": 20, "surface": 30}[kind] + int(noise * 10))
    tgt = cloud(kind, n, rng)
    spacing = n ** (-1.0 / 3.0) if kind != "surface" else n ** (-0.5)
    T0 = rigid(0.02, [1, 2, 3], [0.004, -0.003, 0.002])
    take = rng.permutation(n)[: int(0.7 * n)]          # partial overlap, source order incoherent
    src = orc.transform_points(np.linalg.inv(T0).astype(np.float32), tgt[take])
    src = (src + rng.standard_normal(src.shape).astype(np.float32) * np.float32(noise * spacing)).astype(np.float32)
    radius = 3.0 * spacing
    tree = orc.Tree(tgt)
    eng.set_target(cuda(tgt))
    eng.set_source(cuda(src))

    def check(T, seeded_hits=None):
        idx, d2, st = eng.search_radius_1nn(radius, T)
        q = src if T is None else orc.transform_points(T, src)
        h = compare(idx, d2, q, tgt, tree, radius)
        assert st[0] == h
        return idx, d2

    # before any re-sort: unseeded, then seeded under other transforms
    check(None)
    check(T0)                                              # far from the identity result's matches
    step = rigid(1e-7, [0, 1, 0], [1e-8, 0, 0])
    check((step @ T0).astype(np.float32))                   # ~1 ulp away
    # the loop's state: first pass under a poor initial guess + match-order re-sort
    init = rigid(0.012, [1, 2, 3], [0.002, -0.001, 0.001])
    eng.icp_begin(P2P, radius, init, -1.0)
    a_idx, a_d2 = check(T0)                                 # large step from `init`
    check((step @ T0).astype(np.float32))
    far = rigid(0.05, [3, -1, 2], [0.01, 0.01, -0.02])
    b_idx, b_d2 = check(far)                                # seeds mostly useless
    # unseeded from scratch under the same transform: identical arrays
    eng.drop_seeds()
    u_idx, u_d2 = check(far)
    assert np.array_equal(b_d2, u_d2)
    ne = b_idx != u_idx
    assert not ne.any() or np.array_equal(b_d2[ne], u_d2[ne])
    # a few real iterations, then a search under the loop's own transform and under T0 again
    res = eng.icp_iterate(3)
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    check(T)
    c_idx, c_d2 = check(T0)
    assert np.array_equal(a_d2, c_d2)
    tree.close()


def test_seeded_search_tiny_radius_and_all_misses(eng):
    rng = np.random.default_rng(3)
    n = 150_000
    tgt = cloud("uniform", n, rng)
    src = (tgt[rng.permutation(n)[:100_000]] + rng.standard_normal((100_000, 3)).astype(np.float32) *
           np.float32(0.3 * n ** (-1 / 3))).astype(np.float32)
    tree = orc.Tree(tgt)
    eng.set_target(tgt)
    eng.set_source(src)
    r_small = 0.3 * n ** (-1 / 3)
    eng.icp_begin(P2P, 2.0 * n ** (-1 / 3), None, -1.0)     # seeds from a WIDER search than the next one
    for radius, T in ((r_small, None), (r_small, rigid(0.01, [1, 0, 0], [0.001, 0, 0])),
                      (1e-6, None), (5.0 * n ** (-1 / 3), rigid(0.3, [0, 0, 1], [0.2, 0, 0]))):
        idx, d2, st = eng.search_radius_1nn(radius, T)
        q = src if T is None else orc.transform_points(T, src)
        h = compare(idx, d2, q, tgt, tree, radius)
        assert st[0] == h
    tree.close()


def build_halos(e):
    """Have the target's halos built now (the debug export waits for them)."""
    import ctypes as C
    info = (C.c_int64 * 5)()
    e._chk(e._L.mi_icp_debug_get_tree(e._ctx, info, None, None))
    out = np.empty((int(info[1]), 8, 32), np.float32)
    e._chk(e._L.mi_icp_debug_get_leaf_halos(e._ctx, out.ctypes.data_as(C.c_void_p)))


@pytest.mark.parametrize("kind", ["uniform", "clustered", "surface"])
def test_first_pass_from_its_own_seeds_equals_the_walk_from_the_root(kind):
    """Where the target's halos exist already, a registration loop's FIRST pass starts every query from the
    leaf a greedy descent lands in (launch_nn: locate_leaves) instead of walking the tree from the root.
    Same correspondences, bit for bit, as the first pass of a loop without halos and a one-shot search (both
    walk from the root) and as the oracle -- under a poor initial guess, with partial overlap and noise."""
    from cupoch_amd.engine import Engine
    n = 160_000
    rng = np.random.default_rng({"uniform": 41, "clustered": 42, "surface": 43}[kind])
    tgt = cloud(kind, n, rng)
    spacing = n ** (-1.0 / 3.0) if kind != "surface" else n ** (-0.5)
    take = rng.permutation(n)[: int(0.8 * n)]
    src = (tgt[take] + rng.standard_normal((len(take), 3)).astype(np.float32) * np.float32(0.2 * spacing)).astype(np.float32)
    init = rigid(0.015, [1, -2, 1], [0.003, 0.002, -0.002])
    radius = 3.0 * spacing
    tree = orc.Tree(tgt)
    _, oi, od = tree.search_radius(orc.transform_points(init, src), radius, 1)
    tree.close()
    oi = oi[:, 0]
    never = os.environ.get("MI_ICP_NO_COARSE_FIRST") is not None    # (A/B switches of the library, read once per process)
    always = os.environ.get("MI_ICP_WAIT_LINKS") is not None
    got = {}
    for name in ("root", "loop", "seeds+halos"):
        e = Engine(0)
        e.set_target(cuda(tgt))
        e.set_source(cuda(src))
        if name == "root":                                   # a one-shot search has no seeds of any kind
            dense, d2, st = e.search_radius_1nn(radius, init)
            assert e.last_search_kind() == 0
            fit, rmse = st[0] / len(src), float(np.sqrt(st[1] / max(st[0], 1)))
        else:
            if name == "seeds+halos":
                build_halos(e)
            res = e.icp_begin(P2P, radius, init, -1.0)
            assert e.last_search_kind() == (2 if (name == "seeds+halos" or always) and not never else 0)
            corr = e.get_correspondences()
            dense = np.full(len(src), -1, np.int32)
            dense[corr[:, 0]] = corr[:, 1]
            fit, rmse = res.fitness, res.inlier_rmse
        got[name] = (np.asarray(dense), fit, rmse)
        e.close()
    for name, (dense, fit, rmse) in got.items():
        assert np.array_equal(dense < 0, oi < 0), name
        ne = np.flatnonzero(dense != oi)
        if len(ne):                                          # only exact ties may differ
            q = orc.transform_points(init, src)[ne]
            dd = q - tgt[dense[ne]]
            alt = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
            assert np.array_equal(alt, od[ne, 0]), name
    for name in ("loop", "seeds+halos"):
        assert np.array_equal(got["root"][0], got[name][0]), name
        assert got["root"][1] == pytest.approx(got[name][1], abs=1e-7) and got["root"][2] == pytest.approx(got[name][2], rel=1e-5)


def _run_small_loop(path, est="pt2pl"):
    """(child process) a 40k-point point-to-plane (or point-to-point: no normals) registration, results to `path`"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cupoch_amd.engine import Engine
    from conftest import make_pair
    d = make_pair(40_000, seed=77, noise=0.05)
    e = Engine(0)
    if est == "p2p":
        e.set_target(d["tgt"])
    else:
        e.set_target(d["tgt"], d["tgt_nrm"])
    e.set_source(d["src"])
    res = e.registration_icp(P2P if est == "p2p" else PT2PL, d["max_dist"], None, 1e-6, 1e-6, 25, -1.0)
    corr = e.get_correspondences()
    np.savez(path, T=np.array(res.transformation, np.float32), stat=np.array([res.fitness, res.inlier_rmse, res.iterations]),
             corr=corr)
    e.close()


@pytest.mark.parametrize("est", ["pt2pl", "p2p"])
def test_one_launch_iteration_equals_the_two_kernel_form(tmp_path, est):
    """Sources of up to ~170k points run search + system rows + reduction + step as ONE kernel per
    iteration (fused_small.h; point-to-plane and -- late in round 5 -- point-to-point, whose rows are the Kabsch
    sums).  Same loop with MI_ICP_NO_FUSED_ITERATION=1 in a child process (the switch
    is read once per process): same iteration count, same correspondence set, transformation and statistics
    equal to the rounding of the sums' order."""
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    outs = {}
    for name, extra in (("fused", {}), ("split", {"MI_ICP_NO_FUSED_ITERATION": "1"})):
        path = str(tmp_path / (name + ".npz"))
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_seeded as t; t._run_small_loop(%r, %r)"
                % (os.path.dirname(here), os.path.dirname(os.path.dirname(here)), path, est))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(path)
    a, b = outs["fused"], outs["split"]
    assert int(a["stat"][2]) == int(b["stat"][2])
    assert np.linalg.norm(a["T"] - b["T"]) <= 1e-6
    assert a["stat"][0] == pytest.approx(b["stat"][0], abs=1e-7) and a["stat"][1] == pytest.approx(b["stat"][1], rel=1e-5)
    assert np.array_equal(a["corr"], b["corr"])


def _loop_counters(e):
    import ctypes as C
    out = (C.c_int32 * 4)()
    e._chk(e._L.mi_icp_debug_loop_counters(e._ctx, out))
    return [int(v) for v in out]


def test_relocation_after_large_steps_changes_nothing_but_the_seeds():
    """Round 5: a step that moves the source by more than about a leaf's width (~1.9 point spacings; sized on the device from the update
    and the source's box, loop.h) makes the gated launch in front of the next search replace every seed by the leaf the
    moved query falls into (nn_search.h locate_by_planes).  On a surface started 6 spacings above it the first steps are that
    large: re-locations happen, stop once the steps are small (the launches are disarmed), and the loop's every number
    equals the oracle's restatement of the engine's form -- seeds never change an answer."""
    from cupoch_amd.engine import Engine
    rng = np.random.default_rng(91)
    n = 200_000
    tgt = cloud("surface", n, rng)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    spacing = n ** (-0.5)
    src = (tgt[rng.permutation(n)[: int(0.7 * n)]] + rng.standard_normal((int(0.7 * n), 3)).astype(np.float32) * np.float32(0.05 * spacing)).astype(np.float32)
    init = rigid(0.004, [0, 0, 1], [0.5 * spacing, -0.5 * spacing, 6 * spacing])   # (off the sheet: the first step is ~6 spacings long)
    radius = 12.0 * spacing
    e = Engine(0)
    e.set_target(cuda(tgt), cuda(nrm))
    e.set_source(cuda(src))
    build_halos(e)
    res = e.registration_icp(P2P, radius, init, 0.0, 0.0, 25, -1.0)
    it, passes, reloc, armed = _loop_counters(e)
    have_planes = os.environ.get("MI_ICP_NO_CELLS") is None and os.environ.get("MI_ICP_NO_LOCATE_PLANES") is None
    if have_planes:
        assert e.last_search_kind() in (1, 2)
        assert 1 <= reloc < it, (reloc, it)            # the early steps re-located, the late ones did not
        assert armed == 0                              # ... and the launches were taken off again
    else:
        assert reloc == 0
    o = orc.registration_icp(src, tgt, radius, init=init, est=orc.EST_P2P, det_thresh=-1.0, max_iteration=25,
                             relative_fitness=0.0, relative_rmse=0.0, composed=True)
    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    assert res.iterations == o.iterations == 25
    assert np.linalg.norm(T - o.transformation) <= 1e-6
    assert abs(res.fitness - o.fitness) <= 1e-6
    e.close()


def test_step_stamps_account_for_the_iteration():
    """mi_icp_debug_set_step_stamps (csrc/loop.h): the stamping instantiations of the seeded search and the point-to-plane
    reduction leave the eight spans of every iteration in the loop's stamp words; they are positive where something runs
    and add up to no more than the wall clock of the same iterations."""
    import time
    from cupoch_amd.engine import Engine
    from conftest import make_pair
    d = make_pair(400_000, seed=5)
    e = Engine(0)
    e.set_target(cuda(d["tgt"]), cuda(d["tgt_nrm"]))
    e.set_source(cuda(d["src"]))
    e.set_step_stamps(True)
    e.icp_begin(PT2PL, d["max_dist"], None, -1.0)
    e.icp_iterate(16)
    t0 = time.perf_counter()       # (right behind the call whose last act was to wait for the 16th iteration, and ahead of
    torch.cuda.synchronize()       # everything else: the first counted iteration's "since the last step" span reaches back
    a, tpu = e.get_step_stamps()   # to the end of that iteration, across this whole host-side pause)
    e.icp_iterate(40)
    torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) * 1e6
    b, _ = e.get_step_stamps()
    cnt = int(b[24] - a[24])
    spans = (b[16:24].astype(np.float64) - a[16:24].astype(np.float64)) / tpu
    assert cnt == 40 and tpu > 0
    assert spans[1] > 0 and spans[3] > 0 and spans[6] > 0          # search, reduction's streaming phase, solve
    assert spans.sum() <= 1.05 * wall_us + 50.0
    e.set_step_stamps(False)
    e.close()
