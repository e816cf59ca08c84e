"""A SECOND restatement of registration::RegistrationICP, written separately from icp_oracle.c and on
outside numerics: scipy.spatial.cKDTree for the search, numpy fp64 sums, LAPACK (numpy.linalg solve / det /
svd / eigh) where the reference calls Eigen.

TEST INFRASTRUCTURE ONLY (tests/test_outside_checks.py): nothing under cupoch_amd/ may import this.

Why it exists: the engine's host_solver.h / eigen3.h and icp_oracle.c's ldlt6_solve / svd3 / fast_eigen3x3 are the
same author's restatements of Eigen routines that are absent from the reference checkout (third_party/eigen is an
empty submodule), so a GPU-vs-oracle test compares like with like.  Here every such routine is somebody else's:

  reference                                             here
  knn::KDTreeFlann::SearchRadius (kdtree_flann.inl:96)  cKDTree.query(k=2), the strict fp32 `d2 < r*r` applied after
  ComputeJTJandJTr (utility/eigen.inl:34-145)           rows formed in fp32, J^T J / J^T r as fp64 matrix products
  A.determinant(), A.ldlt().solve(b) (eigen.cu:92-103)  numpy.linalg.det / solve on the fp32 system, in fp64
  Eigen::JacobiSVD (kabsch.cu:108)                      numpy.linalg.svd
  FastEigen3x3 / SqrtMatrix3x3 (eigenvalue.inl:93-177)  numpy.linalg.eigh (batched), incl. the reference's rule of
                                                        taking the root of A / A.maxCoeff() (generalized_icp.cu:91-92)

The loop itself follows registration/registration.cu:121-172 line by line: the source copy is transformed
incrementally in fp32, `transformation = update * transformation`, fitness / rmse from the Euclidean NN distances,
convergence on absolute differences with strict `<`.
"""
import numpy as np
from scipy.spatial import cKDTree

P2P, PT2PL, GICP = 1, 2, 5
F32 = np.float32
FLT_MAX = float(np.finfo(np.float32).max)


class Result:
    def __init__(self, T):
        self.transformation = T.astype(F32)
        self.correspondence_set = np.zeros((0, 2), np.int32)
        self.fitness = 0.0
        self.inlier_rmse = 0.0
        self.iterations = 0


def _transform_points(T, p):
    """geometry_utils.cu:34-52 TransformPoints: p <- R p + t in fp32"""
    R, t = T[:3, :3].astype(F32), T[:3, 3].astype(F32)
    return (p @ R.T + t).astype(F32)


def _rotate(T, v):
    return (v @ T[:3, :3].astype(F32).T).astype(F32)


def _rotate_covs(T, C):
    """geometry_utils.cu:257-265: C <- R C R^T"""
    R = T[:3, :3].astype(F32)
    return np.einsum("ij,njk,lk->nil", R, C, R).astype(F32)


def _correspondences(tree, tgt, pcd, max_dist, T):
    """registration.cu:33-80 over kdtree_flann.inl:96-122 / result_set.h:372-474: nearest target point with
    d2 < r*r (strict, both sides fp32), fitness = C / N, rmse = sqrt(sum d2 / C)"""
    res = Result(T)
    if max_dist <= 0.0 or len(pcd) == 0:
        return res
    r = F32(max_dist)
    r2 = F32(r * r)
    # the two nearest in fp64, then re-decided on the fp32 squared distance (lower index on a tie)
    _, j = tree.query(pcd.astype(np.float64), k=min(2, len(tgt)), distance_upper_bound=float(r) * 1.001, workers=-1)
    j = j.reshape(len(pcd), -1)
    best_d2 = np.full(len(pcd), np.inf, F32)
    best_j = np.full(len(pcd), -1, np.int64)
    for col in range(j.shape[1]):
        jj = j[:, col]
        ok = jj < len(tgt)
        d = pcd[ok] - tgt[jj[ok]]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F32)
        cur_d2, cur_j = best_d2[ok], best_j[ok]
        better = (d2 < cur_d2) | ((d2 == cur_d2) & (jj[ok] < cur_j))
        cur_d2[better], cur_j[better] = d2[better], jj[ok][better]
        best_d2[ok], best_j[ok] = cur_d2, cur_j
    hit = best_d2 < r2
    i = np.nonzero(hit)[0]
    res.correspondence_set = np.stack([i, best_j[hit]], 1).astype(np.int32)
    if len(i):
        res.fitness = float(F32(len(i)) / F32(len(pcd)))
        res.inlier_rmse = float(np.sqrt(F32(best_d2[hit].astype(np.float64).sum()) / F32(len(i))))
    return res


def _rodrigues(x):
    """utility/eigen.cu:28-50 TransformVector6fToMatrix4f"""
    x = x.astype(F32)
    T = np.eye(4, dtype=F32)
    T[:3, 3] = x[3:]
    th = F32(np.sqrt(F32(x[0] * x[0] + x[1] * x[1] + x[2] * x[2])))
    if th == 0:
        return T
    w = x[:3] / th
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)
    c, s = np.cos(np.float64(th)), np.sin(np.float64(th))
    T[:3, :3] = (c * np.eye(3) + s * K + (1 - c) * np.outer(w, w).astype(np.float64)).astype(F32)
    return T


def _solve(JTJ, JTr, det_thresh):
    """utility/eigen.cu:76-122: optional |det| test in fp32 range, JTJ x = -JTr, Rodrigues; failure -> identity"""
    A = JTJ.astype(F32).astype(np.float64)
    b = (-JTr).astype(F32).astype(np.float64)
    if det_thresh > 0:
        with np.errstate(all="ignore"):
            det = np.linalg.det(A)
        if not np.isfinite(det) or abs(det) > FLT_MAX or abs(det) < det_thresh:
            return np.eye(4, dtype=F32)
    try:
        x = np.linalg.solve(A, b)
    except np.linalg.LinAlgError:
        x = np.linalg.lstsq(A, b, rcond=None)[0]
    return _rodrigues(x)


def _update_pt2pl(pcd, tgt, tgt_nrm, cor, det_thresh):
    """transformation_estimation.cu:34-56,195-222"""
    if len(cor) == 0 or tgt_nrm is None:
        return np.eye(4, dtype=F32)
    vs, vt, nt = pcd[cor[:, 0]], tgt[cor[:, 1]], tgt_nrm[cor[:, 1]]
    r = np.einsum("ij,ij->i", (vs - vt).astype(F32), nt).astype(F32)
    J = np.concatenate([np.cross(vs, nt).astype(F32), nt], 1).astype(np.float64)
    return _solve(J.T @ J, J.T @ r.astype(np.float64), det_thresh)


def _update_p2p(pcd, tgt, cor):
    """kabsch.cu:42-120: sums over the correspondences DIVIDED BY model.size() (all source points), JacobiSVD,
    R = V diag(1, 1, det(U V)) U^T, t = ct - R cs"""
    if len(cor) == 0:
        return np.eye(4, dtype=F32)
    n_model = float(len(pcd))
    S, G = pcd[cor[:, 0]].astype(np.float64), tgt[cor[:, 1]].astype(np.float64)
    cs, ct = S.sum(0) / n_model, G.sum(0) / n_model
    H = (S - cs).T @ (G - ct) / n_model
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    R = V @ np.diag([1.0, 1.0, np.linalg.det(U @ V)]) @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = ct - R @ cs
    return T.astype(F32)


def gicp_weights(Cs, Ct):
    """generalized_icp.cu:91-92 W = SqrtMatrix3x3((Ct + Cs)^-1) with the reference's rule (eigenvalue.inl:100-154):
    the root is taken of A / A.maxCoeff() (signed maximum; never scaled back) unless A has no off-diagonal entries
    (then of A's diagonal itself); maxCoeff == 0 -> W = 0.  numpy.linalg.inv / eigh in fp64."""
    M = (Ct + Cs).astype(F32).astype(np.float64)
    Mi = np.linalg.inv(M).astype(F32).astype(np.float64)
    mx = Mi.reshape(-1, 9).max(1)
    off = (Mi[:, 0, 1] ** 2 + Mi[:, 0, 2] ** 2 + Mi[:, 1, 2] ** 2) > 0
    scale = np.where(off & (mx != 0), mx, 1.0)
    w, V = np.linalg.eigh(Mi / scale[:, None, None])
    W = np.einsum("nij,nj,nkj->nik", V, np.sqrt(np.maximum(w, 0.0)), V)
    diag = ~off
    if diag.any():       # the diagonal branch: eval = diagonal, evec = identity (sqrt of a negative entry is NaN there)
        with np.errstate(invalid="ignore"):
            d = np.sqrt(Mi[diag][:, [0, 1, 2], [0, 1, 2]])
        Wd = np.zeros((len(d), 3, 3))
        Wd[:, [0, 1, 2], [0, 1, 2]] = d
        W[diag] = Wd
    W[mx == 0] = 0.0
    return W


def _update_gicp(pcd, pcd_cov, tgt, tgt_cov, cor):
    """generalized_icp.cu:63-105,152-183: three rows per correspondence, J = W [-skew(vs) | I], r = W d"""
    if len(cor) == 0 or pcd_cov is None or tgt_cov is None:
        return np.eye(4, dtype=F32)
    vs, vt = pcd[cor[:, 0]].astype(np.float64), tgt[cor[:, 1]].astype(np.float64)
    W = gicp_weights(pcd_cov[cor[:, 0]], tgt_cov[cor[:, 1]])
    d = (pcd[cor[:, 0]] - tgt[cor[:, 1]]).astype(F32).astype(np.float64)
    x, y, z = vs.T
    zero = np.zeros_like(x)
    negskew = np.stack([zero, z, -y, -z, zero, x, y, -x, zero], 1).reshape(-1, 3, 3)
    Jfull = np.concatenate([negskew, np.broadcast_to(np.eye(3), negskew.shape)], 2)      # (n, 3, 6)
    J = np.einsum("nij,njk->nik", W, Jfull).reshape(-1, 6)
    r = np.einsum("nij,nj->ni", W, d).reshape(-1)
    return _solve(J.T @ J, J.T @ r, -1.0)          # (the reference's default det_thresh there: eigen.h:84)


def system_of_rows(J, r):
    """ComputeJTJandJTr (utility/eigen.inl:34-70): sum J J^T, sum J r, sum r^2 in the 32-double layout of the engine's
    and the oracle's compute_system ([0..20] upper triangle row-major, [21..26] J^T r, [27] sum r^2)"""
    J, r = J.astype(np.float64), r.astype(np.float64)
    A = J.T @ J
    out = np.zeros(32)
    out[:21] = A[np.triu_indices(6)]
    out[21:27] = J.T @ r
    out[27] = r @ r
    return out


def rows_symmetric(vs, ns, vt, nt):
    """transformation_estimation.cu:58-90: n = ns + nt, r = (vs - vt) . n, J = [(vs + vt) x n ; n]"""
    n = (ns + nt).astype(F32)
    r = np.einsum("ij,ij->i", (vs - vt).astype(F32), n).astype(F32)
    return np.concatenate([np.cross((vs + vt).astype(F32), n).astype(F32), n], 1), r


def rows_colored(vs, vt, nt, i_s, i_t, dit, lambda_geometric=0.968):
    """colored_icp.cu:150-216: two rows per correspondence (geometric, photometric), interleaved"""
    slg, slp = F32(np.sqrt(F32(lambda_geometric))), F32(np.sqrt(F32(1.0 - lambda_geometric)))
    d = (vs - vt).astype(np.float64)
    nt64, dit64 = nt.astype(np.float64), dit.astype(np.float64)
    dn = np.einsum("ij,ij->i", d, nt64)
    J0 = np.concatenate([np.cross(vs.astype(np.float64), nt64), nt64], 1) * float(slg)
    r0 = float(slg) * dn
    vs_proj = vs.astype(np.float64) - dn[:, None] * nt64
    is0 = np.einsum("ij,ij->i", dit64, vs_proj - vt.astype(np.float64)) + i_t.astype(np.float64)
    M = np.eye(3)[None, :, :] - np.einsum("ni,nj->nij", nt64, nt64)
    ditM = -np.einsum("ni,nij->nj", dit64, M)
    J1 = np.concatenate([np.cross(vs.astype(np.float64), ditM), ditM], 1) * float(slp)
    r1 = float(slp) * (i_s.astype(np.float64) - is0)
    J = np.stack([J0, J1], 1).reshape(-1, 6)
    r = np.stack([r0, r1], 1).reshape(-1)
    return J, r


def covariances_from_normals(nrm, eps=1e-3):
    """generalized_icp.cu:18-61: C = Rx diag(eps, 1, 1) Rx^T, Rx = GetRotationFromE1ToX(n) (c < -0.99 -> identity)"""
    n = nrm.astype(F32).astype(np.float64)
    v = np.cross(np.array([1.0, 0, 0]), n)
    c = n[:, 0]
    sv = np.zeros((len(n), 3, 3))
    sv[:, 0, 1], sv[:, 0, 2], sv[:, 1, 0] = -v[:, 2], v[:, 1], v[:, 2]
    sv[:, 1, 2], sv[:, 2, 0], sv[:, 2, 1] = -v[:, 0], -v[:, 1], v[:, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        Rx = np.eye(3) + sv + np.einsum("nij,njk->nik", sv, sv) * (1.0 / (1.0 + c))[:, None, None]
    Rx[c < -0.99] = np.eye(3)
    return np.einsum("nij,j,nkj->nik", Rx, np.array([eps, 1.0, 1.0]), Rx).astype(F32)


def registration_icp(src, tgt, max_dist, init=None, est=P2P, det_thresh=None, relative_fitness=1e-6,
                     relative_rmse=1e-6, max_iteration=30, tgt_nrm=None, src_cov=None, tgt_cov=None):
    """registration.cu:121-172"""
    src, tgt = np.ascontiguousarray(src, F32).reshape(-1, 3), np.ascontiguousarray(tgt, F32).reshape(-1, 3)
    if det_thresh is None:
        det_thresh = 1e-6 if est == PT2PL else -1.0
    T = np.eye(4, dtype=F32) if init is None else np.asarray(init, F32).reshape(4, 4).copy()
    tree = cKDTree(tgt.astype(np.float64))
    pcd = src.copy()
    pcd_cov = None if src_cov is None else np.asarray(src_cov, F32).reshape(-1, 3, 3).copy()
    tgt_cov = None if tgt_cov is None else np.asarray(tgt_cov, F32).reshape(-1, 3, 3)
    if not np.allclose(T, np.eye(4), rtol=0, atol=1e-5):        # Eigen isIdentity(), fp32 dummy precision
        pcd = _transform_points(T, pcd)
        if pcd_cov is not None:
            pcd_cov = _rotate_covs(T, pcd_cov)
    res = _correspondences(tree, tgt, pcd, max_dist, T)
    it = 0
    for it in range(1, max_iteration + 1):
        cor = res.correspondence_set
        if est == PT2PL:
            upd = _update_pt2pl(pcd, tgt, tgt_nrm, cor, det_thresh)
        elif est == GICP:
            upd = _update_gicp(pcd, pcd_cov, tgt, tgt_cov, cor)
        else:
            upd = _update_p2p(pcd, tgt, cor)
        T = (upd @ T).astype(F32)
        pcd = _transform_points(upd, pcd)
        if pcd_cov is not None:
            pcd_cov = _rotate_covs(upd, pcd_cov)
        prev = res
        res = _correspondences(tree, tgt, pcd, max_dist, T)
        if abs(F32(prev.fitness) - F32(res.fitness)) < F32(relative_fitness) and \
                abs(F32(prev.inlier_rmse) - F32(res.inlier_rmse)) < F32(relative_rmse):
            break
    res.iterations = it if max_iteration > 0 else 0
    return res
