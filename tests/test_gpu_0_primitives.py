"""GPU: the hand-written device primitives behind the engine (radix sort, scan,
Morton order) against numpy -- bit-exact integer work."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _sort(eng, keys, vals, bits):
    k = np.ascontiguousarray(keys, np.uint64).copy()
    v = np.ascontiguousarray(vals, np.uint32).copy()
    eng._chk(eng._L.mi_icp_debug_sort_pairs(eng._ctx, k.ctypes.data_as(C.c_void_p),
                                            v.ctypes.data_as(C.c_void_p), len(k), bits))
    return k, v


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1024, 1025, 4097, 100003, 3000000])
@pytest.mark.parametrize("bits", [8, 24, 39, 64])
def test_radix_sort_is_a_stable_sort(eng, n, bits):
    rng = np.random.default_rng(n * 131 + bits)
    hi = (1 << bits) - 1
    keys = rng.integers(0, 1 << 62, n, dtype=np.uint64) & np.uint64(hi)
    if n > 1000:                       # plenty of duplicates: stability must show
        keys[: n // 2] = keys[: n // 2] % np.uint64(97)
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort(eng, keys, vals, bits)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(k, keys[order])
    np.testing.assert_array_equal(v, vals[order])


@pytest.mark.parametrize("n", [1, 7, 255, 256, 2047, 2048, 2049, 5000, 8191, 8192, 8193, 40000, 65535, 65536, 65537, 70000, 4194304 + 17])
def test_exclusive_scan(eng, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 5, n, dtype=np.uint32)
    out = np.empty_like(a)
    tot = C.c_uint64(0)
    eng._chk(eng._L.mi_icp_debug_exclusive_scan(eng._ctx, a.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p), n, C.byref(tot)))
    ref = np.concatenate([[0], np.cumsum(a.astype(np.uint64))[:-1]]).astype(np.uint32)
    np.testing.assert_array_equal(out, ref)
    assert tot.value == int(a.astype(np.uint64).sum())


def test_morton_order_is_a_permutation_and_coherent(eng):
    rng = np.random.default_rng(3)
    n = 200000
    pts = rng.random((n, 3), dtype=np.float32)
    order = np.empty(n, np.uint32)
    eng._chk(eng._L.mi_icp_debug_morton_order(eng._ctx, pts.ctypes.data_as(C.c_void_p), n,
                                              order.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    sp = pts[order]
    # 64 consecutive points of the order form a compact blob: extent far below the cloud's
    ext = (sp[: n // 64 * 64].reshape(-1, 64, 3).max(1) - sp[: n // 64 * 64].reshape(-1, 64, 3).min(1)).max(1)
    assert np.median(ext) < 0.15
