"""The loop step's wave-wide 6x6 solve (csrc/wave_solver.h: a matrix row per lane) against the serial
routines it replaces (csrc/host_solver.h: determinant6, ldlt_solve6 -- utility/eigen.cu:107-122), both on
the device, on the same systems: bit for bit, pivoting / degenerate / failing cases included."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tri(A):
    return np.array([A[i, j] for i in range(6) for j in range(i, 6)], np.float64)


def _systems(rng):
    out = []

    def add(A, b):
        s = np.zeros(32, np.float64)
        s[:21] = _tri(A)
        s[21:27] = b
        s[27], s[28], s[29] = 1.0, 1.0, 100.0
        out.append(s)

    for k in range(400):  # J^T J of random rows: what the reduction produces
        J = rng.standard_normal((50, 6)) * rng.choice([1e-3, 1.0, 30.0], size=6)
        r = rng.standard_normal(50) * 0.01
        add(J.T @ J, J.T @ r)
    for k in range(200):  # symmetric indefinite / badly scaled: every pivot order gets exercised
        M = rng.standard_normal((6, 6)) * (10.0 ** rng.integers(-3, 4))
        add((M + M.T) * 0.5, rng.standard_normal(6))
    for k in range(100):  # rank deficient: a planar scene constrains 3 of 6
        J = np.zeros((40, 6))
        J[:, rng.choice(6, 3, replace=False)] = rng.standard_normal((40, 3))
        add(J.T @ J, J.T @ rng.standard_normal(40))
    for k in range(60):   # permuted diagonals (ties and exact zeros on the diagonal)
        d = rng.permutation([0.0, 1.0, 1.0, 2.0, 0.5, 3.0])
        add(np.diag(d), rng.standard_normal(6))
    add(np.zeros((6, 6)), np.zeros(6))
    add(np.eye(6), np.zeros(6))
    add(np.full((6, 6), 1.0), np.ones(6))
    add(np.eye(6) * 1e30, np.ones(6))          # determinant overflows fp32
    add(np.eye(6) * 1e-30, np.ones(6) * 1e-30)  # ... underflows
    A = np.eye(6)
    A[2, 2] = np.nan
    add(A, np.ones(6))
    return np.ascontiguousarray(np.stack(out))


@pytest.mark.parametrize("det_thresh", [-1.0, 1e-6, 1e6])
def test_wave_solve_equals_serial_bitwise(det_thresh):
    from cupoch_amd import _lib
    L = _lib.load()
    S = _systems(np.random.default_rng(5))
    n = S.shape[0]
    o_s = np.zeros((n, 16), np.float32)
    o_w = np.full((n, 16), 7.0, np.float32)
    k_s = np.zeros(n, np.int32)
    k_w = np.full(n, 7, np.int32)
    rc = L.mi_icp_debug_solve_both(0, S.ctypes.data_as(C.c_void_p), n, C.c_float(det_thresh),
                                   o_s.ctypes.data_as(C.c_void_p), o_w.ctypes.data_as(C.c_void_p),
                                   k_s.ctypes.data_as(C.c_void_p), k_w.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert (k_s == k_w).all()
    assert (o_s.view(np.uint32) == o_w.view(np.uint32)).all(), np.nonzero((o_s.view(np.uint32) != o_w.view(np.uint32)).any(1))[0][:10]
    if det_thresh > 0:
        assert 0 < k_s.sum() < n      # both outcomes of the determinant check occur
    else:
        assert k_s.all()
    # and the serial routine is the oracle's solver: checked against it in test_gpu_parity / test_host_solver
