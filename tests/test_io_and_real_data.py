"""PCD reader + the real-scan fixture on the CPU side (oracle only)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as orc


def fragment():
    d = np.load(os.path.join(ROOT, "tests", "golden", "fragment_every3rd.npz"))
    nrm = d["normals"].astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.ascontiguousarray(d["points"]), nrm


def test_pcd_reader_roundtrip(tmp_path):
    from cupoch_amd.io import read_pcd_arrays
    rng = np.random.default_rng(0)
    pts = rng.random((50, 3), dtype=np.float32)
    nrm = rng.random((50, 3), dtype=np.float32)
    rgb = rng.integers(0, 256, (50, 3), dtype=np.uint32)
    packed = ((rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]).astype(np.uint32).view(np.float32)
    rec = np.zeros(50, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<f4"),
                              ("normal_x", "<f4"), ("normal_y", "<f4"), ("normal_z", "<f4")])
    for i, k in enumerate("xyz"):
        rec[k] = pts[:, i]
        rec["normal_" + k] = nrm[:, i]
    rec["rgb"] = packed
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z\n"
           "SIZE 4 4 4 4 4 4 4\nTYPE F F F F F F F\nCOUNT 1 1 1 1 1 1 1\nWIDTH 50\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 50\nDATA binary\n")
    p = tmp_path / "t.pcd"
    p.write_bytes(hdr.encode() + rec.tobytes())
    a = read_pcd_arrays(str(p))
    np.testing.assert_array_equal(a["points"], pts)
    np.testing.assert_array_equal(a["normals"], nrm)
    np.testing.assert_allclose(a["colors"], rgb / 255.0, atol=1e-6)
    asc = tmp_path / "a.pcd"
    asc.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n"
                   "DATA ascii\n1 2 3\n4 5 6\n")
    a = read_pcd_arrays(str(asc))
    np.testing.assert_array_equal(a["points"], [[1, 2, 3], [4, 5, 6]])
    assert a["normals"] is None and a["colors"] is None


def test_pcd_reader_skips_auxiliary_fields_of_any_size(tmp_path):
    """a lidar file with `timestamp U 8` next to x y z must load (the reference skips what it does not
    decode: file_pcd.cu UnpackBinaryPCDElement / CheckHeader); the decoded fields themselves stay strict."""
    from cupoch_amd.io import read_pcd_arrays
    rec = np.zeros(5, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("timestamp", "<u8"), ("ring", "V3")])
    rec["x"], rec["y"], rec["z"] = np.arange(5), 2.0, 3.0
    rec["timestamp"] = 0x1122334455667788 + np.arange(5, dtype=np.uint64)
    hdr = ("VERSION 0.7\nFIELDS x y z timestamp ring\nSIZE 4 4 4 8 3\nTYPE F F F U I\nCOUNT 1 1 1 1 1\nWIDTH 5\nHEIGHT 1\n"
           "POINTS 5\nDATA binary\n")
    p = tmp_path / "aux.pcd"
    p.write_bytes(hdr.encode() + rec.tobytes())
    a = read_pcd_arrays(str(p))
    np.testing.assert_array_equal(a["points"], np.stack([np.arange(5), np.full(5, 2.0), np.full(5, 3.0)], 1).astype(np.float32))
    asc = tmp_path / "aux_ascii.pcd"
    asc.write_text("VERSION 0.7\nFIELDS x y z timestamp\nSIZE 4 4 4 8\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n"
                   "DATA ascii\n1 2 3 1234567890123\n4 5 6 1234567890124\n")
    np.testing.assert_array_equal(read_pcd_arrays(str(asc))["points"], [[1, 2, 3], [4, 5, 6]])
    # ... and a text column that is signed / real-valued whatever its declared type says: skipped all the same
    neg = tmp_path / "aux_ascii_signed.pcd"
    neg.write_text("VERSION 0.7\nFIELDS x y z intensity gain\nSIZE 4 4 4 8 2\nTYPE F F F I U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n"
                   "DATA ascii\n1 2 3 -17 0.5\n4 5 6 -1e3 -2.25\n")
    np.testing.assert_array_equal(read_pcd_arrays(str(neg))["points"], [[1, 2, 3], [4, 5, 6]])
    bad = tmp_path / "x8.pcd"
    bad.write_bytes(b"VERSION 0.7\nFIELDS x y z\nSIZE 8 4 4\nTYPE I F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n" + bytes(16))
    with pytest.raises(ValueError):
        read_pcd_arrays(str(bad))
    big = tmp_path / "aux9.pcd"
    big.write_bytes(b"VERSION 0.7\nFIELDS x y z blob\nSIZE 4 4 4 9\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n" + bytes(21))
    with pytest.raises(ValueError):
        read_pcd_arrays(str(big))


def test_lzf_format_vectors_and_roundtrip():
    """PCD's binary_compressed carries an LZF stream (mi_icp_lzf_*, host helpers of the C ABI).  The decoder
    on streams written out by hand from the published format (the reference's own liblzf: the next test):
    literal runs (ctrl < 32: ctrl + 1 bytes follow) and back
    references (len = (ctrl >> 5) + 2, 7 -> + next byte; distance = ((ctrl & 31) << 8 | next) + 1)."""
    from cupoch_amd.io import lzf_compress, lzf_decompress
    # "abc" literal, then 6 bytes from 3 back (overlapping copy): abcabcabc
    assert lzf_decompress(bytes([2]) + b"abc" + bytes([(6 - 2) << 5, 3 - 1]), 9) == b"abcabcabc"
    # long match: 1 literal 'x', then 40 bytes from 1 back: ctrl = 7 << 5, extension byte = 40 - 2 - 7
    assert lzf_decompress(bytes([0]) + b"x" + bytes([7 << 5, 40 - 2 - 7, 0]), 41) == b"x" * 41
    # distance with a high part: 300 literals (10 runs of <= 32), then 5 bytes from 300 back
    lit = bytes(range(256)) + bytes(range(44))
    runs = b"".join(bytes([len(lit[i:i + 32]) - 1]) + lit[i:i + 32] for i in range(0, 300, 32))
    assert lzf_decompress(runs + bytes([((5 - 2) << 5) | ((300 - 1) >> 8), (300 - 1) & 255]), 305) == lit + lit[:5]
    with pytest.raises(ValueError):
        lzf_decompress(bytes([(3 << 5), 9]), 5)                       # reference before the start of the output
    rng = np.random.default_rng(0)
    for data in (b"", b"a", bytes(rng.integers(0, 256, 10000, dtype=np.uint8)),          # incompressible
                 np.repeat(rng.random(500, dtype=np.float32), 7).tobytes(),              # repetitive floats
                 bytes(100000), (b"0123456789" * 3000)):
        comp = lzf_compress(data)
        assert lzf_decompress(comp, len(data)) == data
        if len(data) > 20000:
            assert len(comp) < len(data) // 4


def test_lzf_against_the_references_vendored_liblzf():
    """csrc/lzf.h restates the LZF format; the reference's PCD reader / writer call the liblzf vendored under
    third_party/liblzf (file_pcd.cu:218,461,690).  Both ways: streams THAT compressor writes decode here to the
    input, streams this compressor writes decode THERE -- what a file exchanged with the reference needs.  (The two
    compressors' byte streams need not be equal: the format leaves the match search to the writer.)"""
    import ctypes as C
    from cupoch_amd.io import lzf_compress, lzf_decompress
    L = orc.ref_lzf()
    if L is None:
        pytest.skip("oracle/_ref/libref_lzf.so not built (needs /root/reference)")
    rng = np.random.default_rng(3)
    pts = rng.random((4000, 3), dtype=np.float32)
    cases = [b"a" * 5, bytes(rng.integers(0, 256, 20000, dtype=np.uint8)), np.repeat(rng.random(700, dtype=np.float32), 5).tobytes(),
             bytes(70000), b"0123456789" * 2500, np.ascontiguousarray(pts.T).tobytes(),          # field-major floats: a PCD payload
             np.round(pts * 64).astype(np.float32).T.tobytes()]
    for data in cases:
        # the reference compresses, the engine decodes
        cap = len(data) * 2 + 64
        buf = C.create_string_buffer(cap)
        n = L.lzf_compress(data, len(data), buf, cap)
        assert n > 0
        assert lzf_decompress(buf.raw[:n], len(data)) == data
        # the engine compresses, the reference decodes
        comp = lzf_compress(data)
        out = C.create_string_buffer(len(data) + 16)
        m = L.lzf_decompress(comp, len(comp), out, len(data) + 16)
        assert m == len(data) and out.raw[:m] == data
        # and neither writer is much worse than the other on compressible data
        if n < len(data) // 2:
            assert len(comp) <= 1.25 * n + 64


@pytest.mark.parametrize("mode", ["binary", "ascii", "binary_compressed"])
def test_pcd_writer_reader_roundtrip_all_data_modes(tmp_path, mode):
    from cupoch_amd.io import read_pcd_arrays, write_pcd_arrays
    rng = np.random.default_rng(1)
    pts = (rng.random((777, 3), dtype=np.float32) * 10 - 5).astype(np.float32)
    nrm = rng.standard_normal((777, 3)).astype(np.float32)
    col = (rng.integers(0, 256, (777, 3)) / 255.0).astype(np.float32)
    p = str(tmp_path / "c.pcd")
    write_pcd_arrays(p, pts, nrm, col, ascii=(mode == "ascii"), compressed=(mode == "binary_compressed"))
    assert ("DATA " + mode).encode() in open(p, "rb").read(400)
    a = read_pcd_arrays(p)
    tol = 0 if mode != "ascii" else 1e-6           # %.10g
    np.testing.assert_allclose(a["points"], pts, rtol=tol, atol=0)
    np.testing.assert_allclose(a["normals"], nrm, rtol=tol, atol=0)
    np.testing.assert_allclose(a["colors"], col, atol=1e-6)


def test_oracle_icp_on_real_scan():
    """The example flow of examples/python/basic/icp_registration.py (threshold 0.02,
    point-to-plane) on the reference's sample scan against a moved copy of itself."""
    pts, nrm = fragment()
    ang = 0.03
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    T[:3, 3] = [0.004, -0.003, 0.002]
    src = orc.transform_points(np.linalg.inv(T).astype(np.float32), pts)
    res = orc.registration_icp(src, pts, 0.02, est=orc.EST_PT2PL, tgt_nrm=nrm, det_thresh=-1.0)
    assert res.fitness > 0.99
    assert np.linalg.norm(res.transformation - T) < 2e-4


def test_ply_and_pcd_writers_roundtrip(tmp_path):
    from cupoch_amd import io
    rng = np.random.default_rng(1)
    pts = rng.random((77, 3), dtype=np.float32)
    nrm = rng.random((77, 3), dtype=np.float32)
    c8 = rng.integers(0, 256, (77, 3)).astype(np.float32) / np.float32(255.0)
    for name, kw in (("b.ply", {}), ("a.ply", {"ascii": True})):
        io.write_ply_arrays(str(tmp_path / name), pts, nrm, c8, **kw)
        a = io.read_point_cloud_arrays(str(tmp_path / name))
        np.testing.assert_array_equal(a["points"], pts)
        np.testing.assert_array_equal(a["normals"], nrm)
        np.testing.assert_allclose(a["colors"], c8, atol=1e-6)
    io.write_pcd_arrays(str(tmp_path / "c.pcd"), pts, nrm, c8)
    a = io.read_point_cloud_arrays(str(tmp_path / "c.pcd"))
    np.testing.assert_array_equal(a["points"], pts)
    np.testing.assert_array_equal(a["normals"], nrm)
    np.testing.assert_allclose(a["colors"], c8, atol=1e-6)
    io.write_ply_arrays(str(tmp_path / "p.ply"), pts)                 # points only
    a = io.read_ply_arrays(str(tmp_path / "p.ply"))
    assert a["normals"] is None and a["colors"] is None and len(a["points"]) == 77
    with pytest.raises(ValueError):
        io.read_point_cloud_arrays(str(tmp_path / "x.xyz"))


def test_ply_reader_skips_faces_and_leading_elements(tmp_path):
    from cupoch_amd.io import read_ply_arrays
    # ascii, a camera element before the vertices and faces (a list property) after
    (tmp_path / "m.ply").write_text(
        "ply\nformat ascii 1.0\ncomment hand made\nelement camera 1\nproperty float fx\nproperty float fy\n"
        "element vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
        "property uchar green\nproperty uchar blue\nelement face 1\nproperty list uchar int vertex_indices\n"
        "end_header\n500 500\n0 0 0 255 0 0\n1 0 0 0 255 0\n0 1 0 0 0 255\n3 0 1 2\n")
    a = read_ply_arrays(str(tmp_path / "m.ply"))
    np.testing.assert_array_equal(a["points"], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    np.testing.assert_array_equal(a["colors"], np.eye(3, dtype=np.float32))
    # binary big endian with doubles
    rec = np.zeros(2, dtype=[("x", ">f8"), ("y", ">f8"), ("z", ">f8")])
    rec["x"], rec["y"], rec["z"] = [1.5, -2.0], [0.25, 4.0], [8.0, 16.0]
    (tmp_path / "be.ply").write_bytes(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty double x\n"
                                      b"property double y\nproperty double z\nend_header\n" + rec.tobytes())
    a = read_ply_arrays(str(tmp_path / "be.ply"))
    np.testing.assert_array_equal(a["points"], [[1.5, 0.25, 8.0], [-2.0, 4.0, 16.0]])


def colored_example_flow_oracle(src, scol, tgt, tcol, scales):
    """examples/python/advanced/colored_pointcloud_registration.py with the oracle"""
    cur = np.eye(4, dtype=np.float32)
    res = None
    for radius, iters in scales:
        sp, _, sc = orc.voxel_downsample(src, radius, colors=scol)
        tp, _, tc = orc.voxel_downsample(tgt, radius, colors=tcol)
        tn = orc.estimate_normals_radius(tp, radius * 2, 30)
        res = orc.registration_colored_icp(sp, tp, radius, sc, tc, tn, init=cur, max_iteration=iters)
        cur = res.transformation
    return res


def test_oracle_colored_icp_example_flow_on_the_real_coloured_fragment():
    from conftest import colored_fragment_pair
    src, scol, tgt, tcol, T = colored_fragment_pair()
    res = colored_example_flow_oracle(src, scol, tgt, tcol, [(0.04, 50), (0.02, 30)])
    assert res.fitness > 0.95
    assert np.linalg.norm(res.transformation - T) < 0.15 * np.linalg.norm(T - np.eye(4))
