"""GPU: invariants of the target tree as built (mi_icp_debug_get_tree) -- the facts the search's
exactness argument rests on (cupoch_amd/csrc/traverse.h):
  * the leaf lines hold every point exactly once, padding elsewhere;
  * a record's child boxes are the exact bounding boxes of the children's points;
  * a node's REGION (record floats 48..53, flag 54) contains no point of any other node in its
    interior -- what lets a lane whose search cube lies inside it stop looking elsewhere."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from cupoch_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def get_tree(eng):
    info = (C.c_int64 * 5)()
    eng._chk(eng._L.mi_icp_debug_get_tree(eng._ctx, info, None, None))
    nts, nleaf, leaf_first, nrec, nt = [int(v) for v in info]
    rec = np.empty((nrec, 64), np.float32)
    lines = np.empty((nleaf, 32), np.float32)
    eng._chk(eng._L.mi_icp_debug_get_tree(eng._ctx, info, rec.ctypes.data_as(C.c_void_p), lines.ctypes.data_as(C.c_void_p)))
    return nts, nleaf, leaf_first, rec, lines, nt


def record_index(node):
    h = 1
    while h * 8 <= node:
        h *= 8
    return node - h + (h - 1) // 7


def check_tree(eng, pts, expect_all_flagged):
    eng.set_target(pts)
    nts, nleaf, leaf_first, rec, lines, nt = get_tree(eng)
    assert nt == len(pts) and nleaf == (nts + 7) // 8      # (the Morton-run fallback tree does not pad to groups)
    nts = nleaf * 8
    xyz = np.stack([lines[:, 0:8], lines[:, 8:16], lines[:, 16:24]], -1).reshape(-1, 3)      # slot -> coordinates
    orig = lines[:, 24:32].copy().view(np.int32).reshape(-1)
    real = orig >= 0
    assert np.array_equal(np.sort(orig[real]), np.arange(len(pts)))                           # a permutation
    assert np.array_equal(xyz[real], pts[orig[real]])
    assert np.isinf(xyz[~real]).all()

    # leaf-level records: child c of node leaf_first + a is leaf 8a + c
    n_ll = (nleaf + 7) // 8
    for a in np.random.default_rng(0).permutation(n_ll)[:300]:
        r = rec[record_index(leaf_first + a)]
        for c in range(8):
            L = 8 * a + c
            p = xyz[L * 8:(L + 1) * 8][real[L * 8:(L + 1) * 8]] if L < nleaf else np.zeros((0, 3), np.float32)
            pair, ab = c >> 1, c & 1
            mn = r[pair * 12 + np.array([0, 2, 4]) + ab]
            mx = r[pair * 12 + np.array([6, 8, 10]) + ab]
            if len(p):
                assert np.array_equal(mn, p.min(0)) and np.array_equal(mx, p.max(0))
            else:
                assert (mn > mx).all()                                                        # inverted: never hit

    # regions, level by level: node `first + t` covers slots [t * span, (t + 1) * span)
    finite = np.isfinite(xyz).all(1) & real
    flagged = total = 0
    first, span = leaf_first, 64
    while first >= 1:
        count = (nts + span - 1) // span
        for t in range(count):
            r = rec[record_index(first + t)]
            lo, hi, flag = r[48:51], r[51:54], r[54:55].view(np.uint32)[0]
            mine = np.zeros(len(xyz), bool)
            mine[t * span:(t + 1) * span] = True
            if not (mine & real).any() or first == 1:      # (the root's own record is never consulted)
                continue
            total += 1
            if flag == 0:
                continue
            flagged += 1
            others = xyz[finite & ~mine]
            strictly_inside = ((others > lo) & (others < hi)).all(1)
            assert not strictly_inside.any(), (first, t, int(strictly_inside.sum()))
            own = xyz[finite & mine]
            inside = ((own >= lo) & (own <= hi)).all(1)
            assert inside.mean() > 0.99                                                       # (near-ties at a median may sit outside)
        if first == 1:
            break
        first //= 8
        span *= 8
    if expect_all_flagged:
        assert flagged == total
    check_leaf_regions(eng, nleaf, xyz, finite)
    check_leaf_halos(eng, nleaf, xyz, finite)
    return flagged, total


def halo_lines():
    """(line, faces that a member lies beyond, exact?) for the 18 halo lines of a leaf (leaf_halo.h)."""
    out = [(f, (f,), True) for f in range(6)]
    for f in range(6):
        for g in range((f | 1) + 1, 6):
            out.append((6 + ((f >> 1) + (g >> 1) - 1) * 4 + (f & 1) * 2 + (g & 1), (f, g), False))
    return out


def check_leaf_halos(eng, nleaf, xyz, finite):
    """The halo lines (leaf_halo.h) of a leaf with region R hold points of OTHER leaves with their true
    coordinates and slots, ascending in slot: face line f the points on or beyond face f of R and no other
    face, edge line (f, g) those on or beyond both; EVERY such point that is nearer to R (L-infinity distance
    to the box) than the line's reach (x[7]) is in the line -- or, where the line names an extension line
    (slot[7] >= 18), nearer than the extension's reach in one of the two; the region record's float 3 is the
    smallest of these reaches.  Lines 26..28 are the NEAR lines: the 7 / 14 / 21 nearest points of other
    leaves whatever they lie beyond, each line's reach covering the lines before it; the region record's float
    3 packs their reaches as three 10-bit fractions of float 7, rounded down."""
    NL = 29
    halos = np.empty((nleaf, NL, 32), np.float32)
    eng._chk(eng._L.mi_icp_debug_get_leaf_halos(eng._ctx, halos.ctypes.data_as(C.c_void_p)))
    reg = np.empty((nleaf, 8), np.float32)
    eng._chk(eng._L.mi_icp_debug_get_leaf_regions(eng._ctx, reg.ctypes.data_as(C.c_void_p)))
    lo, hi, reach_min, packed = reg[:, 0:3], reg[:, 4:7], reg[:, 7], reg[:, 3].copy().view(np.uint32)
    have = np.flatnonzero(reach_min > 0)
    if os.environ.get("MI_ICP_NO_CELLS") is not None or os.environ.get("MI_ICP_NO_LINKS") is not None:
        assert len(have) == 0
        return 0
    leaf_of = np.arange(len(xyz)) // 8
    pts = xyz.astype(np.float64)
    lines = halo_lines()
    assert sorted(l for l, _, _ in lines) == list(range(18))
    with_points = extended = 0

    def read_line(L, line_no, member):
        line = halos[L, line_no]
        slots = line[24:31].copy().view(np.int32)
        used = slots >= 0
        assert (np.diff(slots[used]) > 0).all(), (int(L), line_no, slots)
        assert used.all() or not used[np.argmin(used):].any()                         # unused entries at the end
        got = np.stack([line[0:7], line[8:15], line[16:23]], -1)
        assert np.array_equal(got[used], xyz[slots[used]]), (int(L), line_no)
        assert np.isinf(got[~used]).all()
        assert np.isinf(line[15]) and np.isinf(line[23])                              # the reach is no point
        assert member[slots[used]].all(), (int(L), line_no, "not a member: the leaf's own, or beyond other faces")
        return set(slots[used].tolist()), float(line[7]), int(line[31:32].copy().view(np.int32)[0])

    for L in np.random.default_rng(6).permutation(have)[:300]:
        foreign = finite & (leaf_of != L)
        up = pts - hi[L].astype(np.float64)                   # >= 0: on or beyond the upper face
        dn = lo[L].astype(np.float64) - pts                   # >= 0: on or beyond the lower face
        with np.errstate(invalid="ignore"):
            dist = np.maximum(np.maximum(up, dn).max(1), 0.0)
            beyond = np.stack([up[:, 0] >= 0, dn[:, 0] >= 0, up[:, 1] >= 0, dn[:, 1] >= 0, up[:, 2] >= 0, dn[:, 2] >= 0], 1)
        final_reaches, named = [], set()
        for line_no, faces, exact in lines:
            member = foreign & beyond[:, list(faces)].all(1)
            if exact:
                member &= beyond.sum(1) == 1
            seen, reach, ext = read_line(L, line_no, member)
            must = np.flatnonzero(member & (dist < reach))
            assert set(must.tolist()) <= seen, (int(L), line_no, reach, must[:5], sorted(seen))
            if ext >= 0:
                assert 18 <= ext < NL and ext not in named and len(seen) == 7, (int(L), line_no, ext)
                named.add(ext)
                more, reach2, none = read_line(L, ext, member)
                assert none == -1 and reach2 >= reach and not (seen & more)
                must = np.flatnonzero(member & (dist < reach2))
                assert set(must.tolist()) <= (seen | more), (int(L), line_no, ext, reach2, must[:5])
                reach = reach2
                extended += 1
            final_reaches.append(reach)
            with_points += int(bool(seen))
        assert np.isclose(float(reach_min[L]), min(final_reaches), rtol=1e-6), (int(L), reach_min[L], final_reaches)
        seen = set()
        for k in range(3):
            more, reach, none = read_line(L, 26 + k, foreign)
            assert none == -1 and not (seen & more)
            seen |= more
            must = np.flatnonzero(foreign & (dist < reach))
            assert set(must.tolist()) <= seen, (int(L), "near", k, reach, must[:5])
            q = (int(packed[L]) >> (10 * k)) & 1023
            promised = np.float32(reach_min[L]) * np.float32(0.0009765625) * np.float32(q)
            assert promised <= np.float32(reach), (int(L), "near", k, promised, reach)
            assert promised >= 0.99 * min(float(reach), float(reach_min[L])) - 2e-3 * float(reach_min[L])
    assert len(have) == 0 or with_points > 0
    return len(have)


def check_leaf_regions(eng, nleaf, xyz, finite):
    """A LEAF's region (mi_icp_debug_get_leaf_regions: what finishes a seeded query without a tree
    walk, nn_search.h) holds no point of any other leaf strictly inside; invalid ones are empty."""
    reg = np.empty((nleaf, 8), np.float32)
    eng._chk(eng._L.mi_icp_debug_get_leaf_regions(eng._ctx, reg.ctypes.data_as(C.c_void_p)))
    lo, hi = reg[:, 0:3], reg[:, 4:7]
    valid = (lo <= hi).all(1)
    leaf_of = np.arange(len(xyz)) // 8
    pts, owner = xyz[finite], leaf_of[finite]
    order = np.random.default_rng(5).permutation(np.flatnonzero(valid))[:400]
    for L in order:
        inside = ((pts > lo[L]) & (pts < hi[L])).all(1)
        assert not (inside & (owner != L)).any(), (int(L), int((inside & (owner != L)).sum()))
    # own points: inside the closed region except near-ties at a median
    own_ok = own_n = 0
    for L in order[:200]:
        mine = pts[owner == L]
        own_n += len(mine)
        own_ok += int(((mine >= lo[L]) & (mine <= hi[L])).all(1).sum())
    assert own_n == 0 or own_ok >= 0.98 * own_n
    return int(valid.sum())


def test_regions_uniform_cloud(eng):
    rng = np.random.default_rng(1)
    f, t = check_tree(eng, rng.random((60000, 3), dtype=np.float32), os.environ.get("MI_ICP_NO_CELLS") is None)
    assert t > 900


def test_regions_small_and_tiny_clouds(eng):
    rng = np.random.default_rng(2)
    for n in (1, 9, 100, 2731, 2732, 4097, 9000):
        check_tree(eng, rng.random((n, 3), dtype=np.float32), os.environ.get("MI_ICP_NO_CELLS") is None)


def test_regions_clustered_planar_and_quantised(eng):
    rng = np.random.default_rng(3)
    clustered = np.concatenate([c + rng.normal(0, 0.003, (s, 3)).astype(np.float32)
                                for c, s in zip(rng.random((40, 3)).astype(np.float32), rng.integers(5, 3000, 40))])
    check_tree(eng, clustered, False)
    planar = rng.random((30000, 3), dtype=np.float32)
    planar[:, 2] = 0.25                                                                       # one axis constant
    check_tree(eng, planar, False)
    grid = np.round(rng.random((40000, 3)) * 32).astype(np.float32) / 32                      # many equal coordinates
    check_tree(eng, grid, False)


def test_overflowing_cell_has_no_regions_but_the_rest_does(eng):
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.random((30000, 3), dtype=np.float32),
                          np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (20000, 1))])    # 20k copies: several groups
    flagged, total = check_tree(eng, pts, False)
    if os.environ.get("MI_ICP_NO_CELLS") is None:
        assert 0 < flagged < total
    else:
        assert flagged == 0                      # Morton runs are not kd cells: no regions at all
