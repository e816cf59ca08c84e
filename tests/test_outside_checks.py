"""Outside implementations against everything no reference vector pins.

The engine (csrc/host_solver.h, eigen3.h) and the CPU oracle (icp_oracle.c) are the same author's restatements of
Eigen / closed-form routines that are absent from the reference checkout, so GPU-vs-oracle tests compare like with
like.  Here both are held against somebody else's code -- LAPACK through numpy in fp64, scipy's cKDTree -- piece by
piece, and the loop as a whole against a second restatement written on those (oracle/icp_numpy.py).

  FastEigen3x3 (eigenvalue.inl:93-154)                       numpy.linalg.eigh
  SqrtMatrix3x3 / the GICP weight (generalized_icp.cu:91-92) numpy.linalg.inv + eigh
  A.determinant(), A.ldlt().solve(b) (eigen.cu:92-103)       numpy.linalg.det / solve / lstsq, incl. singular systems
  the colour-gradient fit (colored_icp.cu:72-121)            numpy.linalg.lstsq on the stacked system
  RegistrationICP (registration.cu:121-172)                  oracle/icp_numpy.py on BASELINE configs 1, 2, slices of 3, 5

The engine's half of this file runs its __host__ code (no GPU needed); tests/test_gpu_outside_checks.py repeats the
eigen-solver and the whole loop on the device.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import make_pair
from oracle import icp_numpy as inp
from oracle import oracle as orc

F32 = np.float32


# ------------------------------------------------------------------------------------------------------------------
# symmetric 3x3 matrices of every kind the closed form treats differently
def _rand_rot(rng, m):
    q = rng.standard_normal((m, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(m, 3, 3)


def _from_eigs(rng, ev):
    R = _rand_rot(rng, len(ev))
    return np.einsum("nij,nj,nkj->nik", R, ev, R)


def symmetric_cases(rng, n):
    """name -> (n, 3, 3) float32 symmetric matrices, and whether the eigenvalues are well separated"""
    out = {}
    e = np.sort(rng.uniform(0.05, 1, (n, 3)), 1) * np.array([1.0, 2.0, 4.0])
    out["separated"] = (_from_eigs(rng, e * 10 ** rng.uniform(-3, 3, (n, 1))), True)
    out["general"] = (_from_eigs(rng, rng.uniform(0.01, 1, (n, 3))), False)
    e = rng.uniform(0.1, 1, (n, 3))
    e[:, 1] = e[:, 0]
    out["repeated"] = (_from_eigs(rng, e), False)
    e = rng.uniform(0.1, 1, (n, 3))
    e[:, :] = e[:, :1]
    out["triple"] = (_from_eigs(rng, e), False)
    e = np.zeros((n, 3))
    e[:, 2] = rng.uniform(0.1, 1, n)
    out["rank1"] = (_from_eigs(rng, e), False)
    e = np.sort(rng.uniform(0.1, 1, (n, 3)), 1) * np.array([0.0, 1.0, 3.0])
    out["rank2"] = (_from_eigs(rng, e), True)
    out["negative_definite"] = (-_from_eigs(rng, np.sort(rng.uniform(0.05, 1, (n, 3)), 1) * np.array([1.0, 2.0, 4.0])), True)
    out["indefinite"] = (_from_eigs(rng, rng.uniform(-1, 1, (n, 3))), False)
    d = np.zeros((n, 3, 3))
    d[:, [0, 1, 2], [0, 1, 2]] = rng.uniform(-1, 1, (n, 3))
    out["diagonal"] = (d, True)
    # what GICP feeds it: (Ct + Cs)^-1 of two plane-like covariances with nearly the same normal
    out["gicp_like"] = (_from_eigs(rng, np.stack([np.full(n, 0.5), 0.5 * rng.uniform(1, 1.01, n),
                                                  rng.uniform(100, 500, n)], 1)), False)
    res = {}
    for k, (A, sep) in out.items():
        A = A.astype(F32)
        res[k] = (np.ascontiguousarray((A + A.transpose(0, 2, 1)) * F32(0.5)), sep)
    return res


def oracle_eig(A32):
    n = len(A32)
    ev, vec = np.empty((n, 3), F32), np.empty((n, 3, 3), F32)
    orc.lib().oracle_fast_eigen3x3(orc._p(A32), C.c_int64(n), orc._p(ev), orc._p(vec))
    return ev, vec


def engine_eig(A32, device=-1):
    from cupoch_amd import _lib
    n = len(A32)
    ev, vec, S = np.empty((n, 3), F32), np.empty((n, 3, 3), F32), np.empty((n, 3, 3), F32)
    rc = _lib.load().mi_icp_debug_eigen3(device, A32.ctypes.data, n, ev.ctypes.data, vec.ctypes.data, S.ctypes.data)
    assert rc == 0
    return ev, vec, S


def check_eigen_against_lapack(eig_fn, who, n_per_kind=1500, seed=5):
    """eigenvalues (as a set), residual |B v - l v| and orthonormality against numpy.linalg.eigh in fp64, where B is
    what the reference's routine decomposes: A / A.maxCoeff() (signed maximum) in its general branch, A itself when A
    has no off-diagonal entries; maxCoeff == 0 -> (0, I) (eigenvalue.inl:100-104,151-153).
    Bounds: the closed form finds the eigenvalues through acos(det(B) / 2): at (nearly) repeated eigenvalues its
    argument is (nearly) +-1 and one rounding there moves the angle by sqrt(2 ulp) ~ 3.5e-4 -- 4e-4 |B| is that
    algorithm's own accuracy, everywhere; matrices with separated eigenvalues are held to 5e-6."""
    rng = np.random.default_rng(seed)
    total = 0
    for name, (A32, separated) in symmetric_cases(rng, n_per_kind).items():
        A64 = A32.astype(np.float64)
        mx = A64.reshape(-1, 9).max(1)
        off = (A32[:, 0, 1] * A32[:, 0, 1] + A32[:, 0, 2] * A32[:, 0, 2] + A32[:, 1, 2] * A32[:, 1, 2]) > 0
        zero = mx == 0
        scale = np.where(off & ~zero, mx, 1.0)
        B = A64 / scale[:, None, None]
        B[zero] = 0.0
        w = np.linalg.eigvalsh(B)
        ev, vec = eig_fn(A32)
        assert np.isfinite(ev).all(), (who, name)
        ev64, v = ev.astype(np.float64), vec.astype(np.float64)
        nrm = np.maximum(np.abs(w).max(1), 1e-30)
        dval = np.abs(np.sort(ev64, 1) - w).max(1) / nrm
        if name == "triple":
            # A = c I up to rounding: every row of A - eval I is noise of ~1e-8, the cross products of
            # ComputeEigenvector0 (eigenvalue.inl:30-49) square it to below the smallest fp32 and `rxr / sqrtf(d)`
            # is 0 / 0 -- the reference's routine returns NaN eigenvectors there.  Its eigenvalues are held; its
            # eigenvectors are whatever that division gives (the two restatements agree on it bit for bit, below).
            assert dval.max() <= 4e-4, (who, name, float(dval.max()))
            total += len(A32)
            continue
        assert np.isfinite(vec).all(), (who, name)
        res = np.linalg.norm(np.einsum("nij,njk->nik", B, v) - v * ev64[:, None, :], axis=1).max(1) / nrm
        orth = np.abs(np.einsum("nji,njk->nik", v, v) - np.eye(3)).reshape(-1, 9).max(1)
        dval[zero], res[zero] = np.abs(ev64[zero]).max(1) if zero.any() else 0, 0.0
        if zero.any():
            assert (vec[zero] == np.eye(3, dtype=F32)).all(), (who, name)
        bound = 5e-6 if separated else 4e-4
        assert dval.max() <= bound, (who, name, "eigenvalues", float(dval.max()))
        assert res.max() <= bound, (who, name, "residual", float(res.max()))
        assert orth.max() <= 2e-6, (who, name, "orthonormality", float(orth.max()))
        if name == "diagonal":           # eval = the diagonal as it stands, evec = I: exact
            nz = ~zero
            assert np.array_equal(ev[nz], A32[nz][:, [0, 1, 2], [0, 1, 2]]) and (vec[nz] == np.eye(3, dtype=F32)).all()
        total += len(A32)
    return total


def test_fast_eigen3x3_against_lapack_oracle_and_engine():
    assert check_eigen_against_lapack(oracle_eig, "oracle") >= 10000
    assert check_eigen_against_lapack(lambda A: engine_eig(A)[:2], "engine (host code)") >= 10000
    # (and the two restatements agree bit for bit on the host: same libm, same order of operations)
    for name, (A32, _) in symmetric_cases(np.random.default_rng(9), 300).items():
        eo, vo = oracle_eig(A32)
        ee, ve, _ = engine_eig(A32)
        assert np.array_equal(eo, ee) and np.array_equal(vo, ve, equal_nan=True), name


def _plane_covs(rng, n, eps=1e-3, spread=0.05):
    """pairs of GICP covariances (generalized_icp.cu:37-61: R diag(eps, 1, 1) R^T) whose normals differ by ~spread"""
    a = rng.standard_normal((n, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + spread * rng.standard_normal((n, 3))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return inp.covariances_from_normals(a.astype(F32), eps), inp.covariances_from_normals(b.astype(F32), eps)


def test_gicp_weight_against_lapack_inverse_and_root():
    """W = SqrtMatrix3x3((Ct + Cs)^-1): the oracle's W and the engine's S = W W (reduce.h never forms the root) against
    numpy.linalg.inv + eigh in fp64, on covariances of the kind GICP makes (nearly equal normals -> a repeated small
    eigenvalue, the closed form's worst case) and on general SPD pairs."""
    rng = np.random.default_rng(3)
    for kind in ("planes", "planes_identical", "general"):
        n = 4000
        if kind == "general":
            Cs = _from_eigs(rng, rng.uniform(0.01, 1, (n, 3))).astype(F32)
            Ct = _from_eigs(rng, rng.uniform(0.01, 1, (n, 3))).astype(F32)
        else:
            Cs, Ct = _plane_covs(rng, n, spread=0.05 if kind == "planes" else 0.0)
        W_np = inp.gicp_weights(Cs, Ct)
        S_np = np.einsum("nij,njk->nik", W_np, W_np)
        unit = np.abs(S_np).reshape(n, -1).max(1)
        W_or = np.stack([orc.gicp_weight(Cs[i], Ct[i]) for i in range(n)]).astype(np.float64)
        assert np.isfinite(W_or).all()
        # the eigenvalues of the closed form are sqrt(ulp)-accurate where they repeat (check_eigen_against_lapack):
        # 4e-4 of the largest; the ROOT of a small one (0.001 of the largest here: the two in-plane directions)
        # moves by d / (2 sqrt(lambda)) -- up to 5e-3 of |W|.  That is the reference's algorithm, not a slip of
        # the restatement: W W, which is all the system needs, is back within 8e-4.
        assert (np.abs(W_or - W_np).reshape(n, -1).max(1) <= 6e-3 * np.sqrt(unit)).all(), kind
        S_or = np.einsum("nij,njk->nik", W_or, W_or)
        assert (np.abs(S_or - S_np).reshape(n, -1).max(1) <= 8e-4 * unit).all(), kind
        # engine: S from the fp32 cofactor inverse of (Ct + Cs), as reduce.h forms it
        M = (Ct + Cs).astype(F32)
        Mi = np.linalg.inv(M.astype(np.float64)).astype(F32)
        _, _, S_en = engine_eig(np.ascontiguousarray((Mi + Mi.transpose(0, 2, 1)) * F32(0.5)))
        assert (np.abs(S_en.astype(np.float64) - S_np).reshape(n, -1).max(1) <= 2e-6 * unit).all(), kind


# ------------------------------------------------------------------------------------------------------------------
def _system(J, r):
    A = J.T @ J
    s = np.zeros(32)
    s[:21] = A[np.triu_indices(6)]
    s[21:27] = J.T @ r
    return s, A


def _pt2pl_rows(p, n):
    return np.concatenate([np.cross(p, n), n], 1)


def _solvers():
    from cupoch_amd import engine
    return (("oracle", orc.solve_system), ("engine", engine.solve_system))


def _x_of(T):
    """the 6-vector back out of TransformVector6fToMatrix4f's result (small angles)"""
    R = T[:3, :3].astype(np.float64)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    s = np.linalg.norm(w)
    if s > 0:
        w *= np.arcsin(min(s, 1.0)) / s
    return np.concatenate([w, T[:3, 3].astype(np.float64)])


def test_ldlt_and_determinant_against_lapack_also_on_singular_systems():
    """SolveLinearSystemPSD<6> (eigen.cu:76-105): x against numpy.linalg.solve in fp64 with a bound that scales with
    the condition number; the determinant against numpy.linalg.det incl. fp32 overflow (quirk 6) and exact zero; a WALL
    (all target normals equal: rank 3 -- the point-to-plane twin of the planar-Kabsch case) axis-aligned and tilted."""
    rng = np.random.default_rng(1)
    det6 = orc.lib().oracle_det6
    det6.restype = C.c_float
    for case in range(300):
        n = int(rng.integers(20, 400))
        cond_target = 10 ** rng.uniform(0, 5)
        J = rng.standard_normal((n, 6)) * np.geomspace(1.0, 1.0 / np.sqrt(cond_target), 6) * 10 ** rng.uniform(-1, 2)
        J = J @ np.linalg.qr(rng.standard_normal((6, 6)))[0]
        x_true = rng.standard_normal(6) * 1e-2
        r = -(J @ x_true) + 1e-4 * rng.standard_normal(n)
        s, _ = _system(J, r)
        A32 = np.zeros((6, 6))
        A32[np.triu_indices(6)] = s[:21].astype(F32)
        A32 = A32 + A32.T - np.diag(np.diag(A32))
        b32 = (-s[21:27]).astype(F32).astype(np.float64)
        x_ref = np.linalg.solve(A32, b32)
        cond = np.linalg.cond(A32)
        for who, solve in _solvers():
            ok, T = solve(s, -1.0)
            assert ok
            x = _x_of(T)
            assert np.abs(x - x_ref).max() <= 4e-7 * cond * max(np.abs(x_ref).max(), 1e-3) + 1e-6, (who, case, cond)
        d_ref = np.linalg.det(A32)
        d = float(det6(orc._p(np.ascontiguousarray(A32.astype(F32)))))
        if abs(d_ref) < 3e38 and abs(d_ref) > 1e-30:
            assert abs(d - d_ref) <= 1e-4 * abs(d_ref) * max(1.0, cond * 1e-3), (case, d, d_ref)
        # the det check (both implementations, through solve_system): accepted iff 1e-6 <= |det| <= FLT_MAX
        expect = (abs(d_ref) >= 1e-6) and (abs(d_ref) <= 3.3e38)
        if abs(d_ref) > 1.1e-6 and abs(d_ref) < 3e38 or abs(d_ref) < 0.9e-6 or abs(d_ref) > 3.6e38:
            for who, solve in _solvers():
                assert solve(s, 1e-6)[0] == expect, (who, case, d_ref)
    # ---- walls
    for tilt, aligned in ((0.0, True), (0.0, False), (3e-2, False), (1e-1, False)):
        n = 5000
        p = rng.uniform(-1, 1, (n, 3))
        p[:, 2] = 0.3
        nr = np.tile([0, 0, 1.0], (n, 1)) + tilt * rng.standard_normal((n, 3))
        nr /= np.linalg.norm(nr, axis=1, keepdims=True)
        if not aligned:
            Q = _rand_rot(rng, 1)[0]
            p, nr = p @ Q.T, nr @ Q.T
        p32, n32 = p.astype(F32).astype(np.float64), nr.astype(F32).astype(np.float64)
        J = _pt2pl_rows(p32, n32)
        x_true = np.linalg.lstsq(J, J @ np.array([0.01, -0.02, 0.005, 0.01, 0.02, 0.03]), rcond=1e-9)[0]
        r = -(J @ x_true)
        s, A = _system(J, r)
        A32 = A.astype(F32).astype(np.float64)
        b32 = (-s[21:27]).astype(F32).astype(np.float64)
        sv = np.linalg.svd(A32, compute_uv=False)
        rank = int((sv > 1e-5 * sv[0]).sum())
        assert rank == (3 if tilt == 0.0 else 6)
        for who, solve in _solvers():
            if aligned:
                # exact zeros in the matrix: determinant exactly 0 -> rejected by the check; without the check the
                # zero pivots are skipped (Eigen's ldlt().solve() takes D's pseudo-inverse) and x is the
                # minimum-norm solution LAPACK's lstsq gives
                assert solve(s, 1e-6)[0] is False, who
                ok, T = solve(s, -1.0)
                x_ref = np.linalg.lstsq(A32, b32, rcond=1e-6)[0]
                assert ok and np.abs(_x_of(T) - x_ref).max() <= 2e-6, (who, _x_of(T), x_ref)
            elif tilt == 0.0:
                # rank 3 with rounding-noise pivots: whatever comes back must still solve the consistent system
                # in the directions the data sees (the range of A); the null-space part is noise in the reference too
                ok, T = solve(s, -1.0)
                x = _x_of(T)
                if np.isfinite(x).all() and np.abs(x).max() < 0.5:
                    U = np.linalg.svd(A32)[0][:, :3]
                    assert np.abs(U.T @ (A32 @ x - b32)).max() <= 1e-3 * np.abs(b32).max(), (who, x)
            else:
                ok, T = solve(s, -1.0)
                x_ref = np.linalg.solve(A32, b32)
                cond = sv[0] / sv[-1]
                assert ok and np.abs(_x_of(T) - x_ref).max() <= 4e-7 * cond * np.abs(x_ref).max() + 1e-6, (who, tilt)


# ------------------------------------------------------------------------------------------------------------------
def test_colour_gradient_fit_against_lstsq():
    """InitializePointCloudForColoredICP (colored_icp.cu:72-148): the oracle's per-point fit against
    numpy.linalg.lstsq on the stacked system [v_k^T; (nn - 1) n^T; 1e-3 I] g = [di_k; 0; 0] (whose normal equations are
    the reference's AtA, Atb), neighbours from scipy's cKDTree (the max_nn nearest within the radius, nearest first,
    the first -- the point itself -- skipped)."""
    from conftest import make_colored
    from scipy.spatial import cKDTree
    tgt, col, _ = make_colored(6000, seed=4, planar=False)
    nrm = orc.estimate_normals_knn(tgt, 20)
    radius, max_nn = 6.0, 30
    g = orc.color_gradients(tgt, nrm, col, radius, max_nn).astype(np.float64)
    inten = orc.intensity(col).astype(np.float64)
    P, N = tgt.astype(np.float64), nrm.astype(np.float64)
    dist, idx = cKDTree(P).query(P, k=max_nn, distance_upper_bound=radius)
    worst, checked = 0.0, 0
    for i in range(0, len(P), 7):
        nb = idx[i][(idx[i] < len(P)) & (dist[i] ** 2 < radius * radius)][1:]
        if len(nb) < 4:
            assert (g[i] == 0).all()
            continue
        d = P[nb] - P[i]
        v = d - (d @ N[i])[:, None] * N[i]
        nn = len(nb)
        A = np.concatenate([v, (nn - 1) * N[i][None, :], 1e-3 * np.eye(3)])
        b = np.concatenate([inten[nb] - inten[i], np.zeros(4)])
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        # fp32 normal equations: their condition number is ((nn-1)^2 / smallest tangent eigenvalue)
        lam = np.linalg.eigvalsh(v.T @ v + (nn - 1) ** 2 * np.outer(N[i], N[i]))
        tol = 3e-7 * (lam[2] / lam[0]) * max(np.abs(ref).max(), 1e-4) + 1e-7
        err = np.abs(g[i] - ref).max()
        worst = max(worst, err / tol)
        checked += 1
    assert checked > 500 and worst <= 1.0, worst


def _relative(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_symmetric_and_colored_systems_against_numpy_rows():
    """The 6x6 systems of the two estimators the second loop does not run: the oracle's compute_system against rows
    written out in numpy fp64 (transformation_estimation.cu:58-90; colored_icp.cu:150-216)."""
    from conftest import make_colored
    d = make_pair(30000, seed=8, noise=0.05)
    _, idx, _ = orc.search_radius(d["tgt"], d["src"], d["max_dist"], 1)
    j = idx[:, 0]
    cor = np.stack([np.arange(len(j)), j], 1)[j >= 0].astype(np.int32)
    sys_or = orc.compute_system(orc.EST_SYM, d["src"], d["tgt"], cor, src_nrm=d["src_nrm"], tgt_nrm=d["tgt_nrm"])
    J, r = inp.rows_symmetric(d["src"][cor[:, 0]], d["src_nrm"][cor[:, 0]], d["tgt"][cor[:, 1]], d["tgt_nrm"][cor[:, 1]])
    sys_np = inp.system_of_rows(J, r)
    assert _relative(sys_or[:21], sys_np[:21]) <= 1e-6 and _relative(sys_or[21:27], sys_np[21:27]) <= 1e-5
    assert abs(sys_or[27] - sys_np[27]) <= 1e-5 * sys_np[27]
    # colored: gradients from the (lstsq-checked) oracle fit, intensities as colored_icp.cu:91 forms them
    tgt, col, T = make_colored(12000, seed=6, planar=False)
    nrm = orc.estimate_normals_knn(tgt, 20)
    grad = orc.color_gradients(tgt, nrm, col, 6.0, 30)
    src = orc.transform_points(np.linalg.inv(T).astype(F32), tgt)
    orc.set_colored_context(col, col, grad, 0.968)
    _, idx, _ = orc.search_radius(tgt, src, 3.0, 1)
    j = idx[:, 0]
    cor = np.stack([np.arange(len(j)), j], 1)[j >= 0].astype(np.int32)
    sys_or = orc.compute_system(orc.EST_COLORED, src, tgt, cor, tgt_nrm=nrm)
    inten = orc.intensity(col)
    J, r = inp.rows_colored(src[cor[:, 0]], tgt[cor[:, 1]], nrm[cor[:, 1]], inten[cor[:, 0]], inten[cor[:, 1]],
                            grad[cor[:, 1]], 0.968)
    sys_np = inp.system_of_rows(J, r)
    assert _relative(sys_or[:21], sys_np[:21]) <= 2e-6 and _relative(sys_or[21:27], sys_np[21:27]) <= 2e-5
    assert abs(sys_or[27] - sys_np[27]) <= 2e-5 * sys_np[27]
    assert abs(orc.compute_rmse(orc.EST_COLORED, src, tgt, cor, tgt_nrm=nrm) - sys_np[27]) <= 1e-4 * sys_np[27]


# ------------------------------------------------------------------------------------------------------------------
# The loop as a whole: icp_oracle.c against oracle/icp_numpy.py
def _both_loops(src, tgt, max_dist, est, **kw):
    kw_np = dict(kw)
    kw_or = dict(kw)
    kw_np.pop("src_nrm", None)
    a = orc.registration_icp(src, tgt, max_dist, est=est, **kw_or)
    b = inp.registration_icp(src, tgt, max_dist, est=est, **kw_np)
    return a, b


def _assert_same_run(a, b, tol=1e-6):
    assert a.iterations == b.iterations, (a.iterations, b.iterations)
    assert np.array_equal(a.correspondence_set, b.correspondence_set)
    assert abs(a.fitness - b.fitness) <= 1e-7 and abs(a.inlier_rmse - b.inlier_rmse) <= 2e-8 + 1e-5 * a.inlier_rmse
    err = float(np.linalg.norm(a.transformation.astype(np.float64) - b.transformation.astype(np.float64)))
    assert err <= tol, err
    return err


def test_second_loop_config1_point_to_point_100k():
    """BASELINE config 1: 100k-vs-100k point-to-point, 30 iterations allowed"""
    d = make_pair(100000, seed=42)
    a, b = _both_loops(d["src"], d["tgt"], d["max_dist"], inp.P2P)
    _assert_same_run(a, b)
    assert np.linalg.norm(b.transformation - d["T_gt"]) < 2e-5


def test_second_loop_config2_voxel_then_point_to_plane():
    """BASELINE config 2 at a quarter of its size: VoxelDownSample(0.02) on both clouds (the oracle's, itself pinned by
    the reference's golden vector), then point-to-plane with r = 0.04"""
    d = make_pair(250000, seed=42)
    sp, _, _ = orc.voxel_downsample(d["src"], 0.02)
    tp, tn, _ = orc.voxel_downsample(d["tgt"], 0.02, normals=d["tgt_nrm"])
    a, b = _both_loops(sp, tp, 0.04, inp.PT2PL, tgt_nrm=tn, det_thresh=-1.0)
    _assert_same_run(a, b, tol=2e-6)
    assert a.iterations >= 3


@pytest.mark.parametrize("noise", [0.0, 0.1])
def test_second_loop_config3_slice_point_to_plane_200k(noise):
    """a 200k-point pair of config 3's kind (exact correspondences; and with measurement noise 0.1 spacings, where the
    loop runs longer and correspondences change between iterations)"""
    d = make_pair(200000, seed=42, noise=noise)
    a, b = _both_loops(d["src"], d["tgt"], d["max_dist"], inp.PT2PL, tgt_nrm=d["tgt_nrm"], det_thresh=-1.0)
    _assert_same_run(a, b)
    # the default det_thresh (1e-6): the determinant check is live and passes at this size ...
    a, b = _both_loops(d["src"][:50000], d["tgt"], d["max_dist"], inp.PT2PL, tgt_nrm=d["tgt_nrm"], max_iteration=3)
    _assert_same_run(a, b)
    assert not np.array_equal(b.transformation, np.eye(4, dtype=F32))
    # ... and the same cloud in millimetres overflows the fp32 determinant (~1e46): isinf -> identity updates (quirk 6;
    # at 10M points it happens in metres too)
    k = F32(1000.0)
    a, b = _both_loops(d["src"][:50000] * k, d["tgt"] * k, d["max_dist"] * 1000.0, inp.PT2PL, tgt_nrm=d["tgt_nrm"],
                       max_iteration=3)
    _assert_same_run(a, b)
    assert np.array_equal(b.transformation, np.eye(4, dtype=F32)) and b.iterations == 1


@pytest.mark.parametrize("noise", [0.0, 0.1])
def test_second_loop_config5_slice_generalized_icp_200k(noise):
    """a 200k-point pair of config 5's kind: covariances from the synthetic normals (epsilon 1e-3), GICP.  The second
    loop takes W from LAPACK's eigh; the oracle from the closed form, whose small eigenvalues are sqrt(eps)-accurate
    (test_gicp_weight_against_lapack_inverse_and_root) -- the systems agree to ~1e-5, the transforms to 1e-6."""
    d = make_pair(200000, seed=42, noise=noise)
    sc, tc = orc.covariances_from_normals(d["src_nrm"]), orc.covariances_from_normals(d["tgt_nrm"])
    assert np.abs(sc - inp.covariances_from_normals(d["src_nrm"])).max() <= 5e-7
    a, b = _both_loops(d["src"], d["tgt"], d["max_dist"], inp.GICP, src_cov=sc, tgt_cov=tc)
    _assert_same_run(a, b, tol=2e-6)
