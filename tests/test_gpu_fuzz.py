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
    # These inputs are deliberately nasty (clusters, duplicates, 30-100 % overlap, noise, radii of
    # several spacings): a match that flips between two near-equidistant targets -- the engine
    # applies the composed T to the pristine source, the reference restatement transforms its copy
    # incrementally, the positions differ by ~1e-7 -- moves a few-thousand-point centroid by ~1e-5,
    # and with duplicates a tie resolved differently feeds a different random normal into
    # point-to-plane.  So: same statistics, transforms to 1e-4 (the 1e-5 bar is held on the
    # BASELINE-style inputs of test_gpu_parity / test_gpu_scale).
    assert abs(res.fitness - o.fitness) <= 2e-3
    assert abs(res.inlier_rmse - o.inlier_rmse) <= 1e-3 * max(o.inlier_rmse, spacing)
    if est == 1:
        assert np.linalg.norm(Tg - o.transformation) <= 1e-4 * max(1.0, ext)


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
