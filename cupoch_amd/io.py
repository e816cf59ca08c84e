"""Minimal point-cloud readers for the formats on either side of the ICP path
(SURVEY.md section 8f-3): PCD (ascii / binary, fields x y z [normal_x normal_y
normal_z] [rgb]) as written by PCL / Open3D / cupoch
(reference: src/cupoch/io/class_io/pointcloud_io.cpp, io/file_format/file_pcd.cu).
Host-side parsing with numpy; the arrays go to the GPU through
utility.Vector3fVector / Engine.set_*."""
import numpy as np


def read_pcd_arrays(path):
    """Returns dict(points (n,3) f32, normals (n,3) f32 or None, colors (n,3) f32 in [0,1] or None)."""
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PCD: no DATA line")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, val = s.partition(" ")
            header[key.upper()] = val.split()
            if key.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(x) for x in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(x) for x in header.get("COUNT", ["1"] * len(fields))]
        n = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        mode = header["DATA"][0].lower()
        dt = []
        for name, sz, ty, cnt in zip(fields, sizes, types, counts):
            base = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
                    ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}[(ty.upper(), sz)]
            dt.append((name, base) if cnt == 1 else (name, base, (cnt,)))
        dt = np.dtype(dt)
        if mode == "binary":
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        elif mode == "ascii":
            rec = np.loadtxt(f, dtype=dt, max_rows=n, ndmin=1)
        else:
            raise ValueError("PCD: DATA %s is not supported (binary_compressed needs lzf)" % mode)

    def cols(names):
        if not all(k in rec.dtype.names for k in names):
            return None
        return np.ascontiguousarray(np.stack([rec[k].astype(np.float32) for k in names], 1))

    out = dict(points=cols(["x", "y", "z"]), normals=cols(["normal_x", "normal_y", "normal_z"]), colors=None)
    if out["points"] is None:
        raise ValueError("PCD: x y z fields missing")
    for key in ("rgb", "rgba"):
        if key in rec.dtype.names:
            raw = rec[key]
            u = raw.view(np.uint32) if raw.dtype.kind == "f" else raw.astype(np.uint32)
            out["colors"] = np.stack([(u >> 16) & 255, (u >> 8) & 255, u & 255], 1).astype(np.float32) / 255.0
            break
    return out


def read_point_cloud(path):
    """cupoch.io.read_point_cloud for .pcd files -> geometry.PointCloud on the GPU."""
    from . import geometry
    a = read_pcd_arrays(path)
    pc = geometry.PointCloud(a["points"])
    if a["normals"] is not None:
        pc.normals = a["normals"]
    if a["colors"] is not None:
        pc.colors = a["colors"]
    return pc
