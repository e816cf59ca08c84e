"""Point-cloud readers / writers for the formats on either side of the ICP path
(SURVEY.md section 8f-3): PCD (ascii / binary, fields x y z [normal_x normal_y
normal_z] [rgb]) and PLY (ascii / binary, vertex x y z [nx ny nz] [red green blue])
as written by PCL / Open3D / cupoch (reference: src/cupoch/io/class_io/pointcloud_io.cpp,
io/file_format/file_pcd.cu, file_ply.cu).
Host-side parsing with numpy; the arrays go to the GPU through
utility.Vector3fVector / Engine.set_*."""
import numpy as np


def lzf_decompress(data, out_size):
    """The LZF stream of PCD's binary_compressed (host helper of the C ABI: mi_icp_lzf_decompress)."""
    import ctypes as C
    from . import _lib
    src = bytes(data)
    out = C.create_string_buffer(max(int(out_size), 1))
    got = _lib.load().mi_icp_lzf_decompress(src, len(src), out, int(out_size))
    if got != out_size:
        raise ValueError("PCD: corrupt LZF stream (%d of %d bytes)" % (got, out_size))
    return out.raw[:out_size]


def lzf_compress(data):
    import ctypes as C
    from . import _lib
    src = bytes(data)
    out = C.create_string_buffer(2 * len(src) + 16)
    got = _lib.load().mi_icp_lzf_compress(src, len(src), out, len(out))
    if got <= 0 and len(src):
        raise ValueError("LZF compression failed")
    return out.raw[:got]


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
        decoded = ("x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb", "rgba")
        known = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
                 ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
        for name, sz, ty, cnt in zip(fields, sizes, types, counts):
            base = known.get((ty.upper(), sz))
            if base is None:
                # a field the reader does not decode (a lidar's `timestamp U 8`) is skipped by its size alone, as the
                # reference skips it (file_pcd.cu: UnpackBinaryPCDElement knows no such size, CheckHeader asks for x y z)
                if name in decoded or not 1 <= sz <= 8:
                    raise ValueError("PCD: field %s has no decodable type/size (%s %d)" % (name, ty, sz))
                base = {1: "u1", 2: "<u2", 4: "<u4", 8: "<u8"}.get(sz, "V%d" % sz)
                if mode == "ascii":
                    # text: whatever the column holds (a signed or real-valued `intensity I 8`) parses as a double
                    # and is dropped; an unsigned dtype made np.loadtxt raise where the C++ reader skips
                    base = "<f8"
            dt.append((name, base) if cnt == 1 else (name, base, (cnt,)))
        dt = np.dtype(dt)
        if mode == "binary":
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        elif mode == "ascii":
            rec = np.loadtxt(f, dtype=dt, max_rows=n, ndmin=1)
        elif mode == "binary_compressed":
            # uint32 compressed size, uint32 uncompressed size, an LZF stream whose payload is
            # FIELD-major: all x, then all y, ... (file_pcd.cu:455-520)
            csize, usize = np.frombuffer(f.read(8), "<u4")
            if int(usize) != n * dt.itemsize:
                raise ValueError("PCD: binary_compressed payload of %d bytes for %d records of %d" % (usize, n, dt.itemsize))
            raw = lzf_decompress(f.read(int(csize)), int(usize))
            rec = np.zeros(n, dt)
            pos = 0
            for name in dt.names:
                sub = dt.fields[name][0]
                nbytes = sub.itemsize * n
                rec[name] = np.frombuffer(raw, dtype=sub, count=n, offset=pos)
                pos += nbytes
        else:
            raise ValueError("PCD: unknown DATA mode %r" % mode)

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
    """cupoch.io.read_point_cloud for .pcd / .ply files -> geometry.PointCloud on the GPU."""
    from . import geometry
    a = read_point_cloud_arrays(path)
    pc = geometry.PointCloud(a["points"])
    if a["normals"] is not None:
        pc.normals = a["normals"]
    if a["colors"] is not None:
        pc.colors = a["colors"]
    return pc


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_arrays(path):
    """PLY (ascii / binary_little_endian / binary_big_endian), vertex element with x y z
    [nx ny nz] [red green blue]; other elements (faces) are ignored
    (reference: src/cupoch/io/file_format/file_ply.cu ReadPointCloudFromPLY via rply).
    Returns the same dict as read_pcd_arrays."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("PLY: missing magic")
        fmt = None
        elements = []   # (name, count, [(prop, type) | (prop, ("list", count_type, item_type))])
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY: no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[4], ("list", tok[2], tok[3])))
                else:
                    elements[-1][2].append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("PLY: unknown format %r" % fmt)
        rec = None
        for name, count, props in elements:
            has_list = any(isinstance(t, tuple) for _, t in props)
            if name != "vertex":
                if rec is not None:
                    break                     # everything we need has been read
                if has_list or fmt == "ascii":
                    if fmt != "ascii":
                        raise ValueError("PLY: list element %r before the vertex element" % name)
                    for _ in range(count):
                        f.readline()
                else:
                    size = sum(np.dtype(_PLY_TYPES[t]).itemsize for _, t in props)
                    f.seek(size * count, 1)
                continue
            if has_list:
                raise ValueError("PLY: list properties on vertices are not supported")
            if fmt == "ascii":
                dt = np.dtype([(p, _PLY_TYPES[t]) for p, t in props])
                rec = np.loadtxt(f, dtype=dt, max_rows=count, ndmin=1) if count else np.zeros(0, dt)
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(p, order + _PLY_TYPES[t]) for p, t in props])
                rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        if rec is None:
            raise ValueError("PLY: no vertex element")

    def cols(names, scale=None):
        if not all(k in rec.dtype.names for k in names):
            return None
        a = np.stack([rec[k].astype(np.float32) for k in names], 1)
        if scale is not None and rec[names[0]].dtype.kind in "ui":
            a = a / np.float32(scale)
        return np.ascontiguousarray(a, dtype=np.float32)

    out = dict(points=cols(["x", "y", "z"]), normals=cols(["nx", "ny", "nz"]),
               colors=cols(["red", "green", "blue"], 255.0))
    if out["points"] is None:
        raise ValueError("PLY: x y z properties missing")
    return out


def write_ply_arrays(path, points, normals=None, colors=None, ascii=False):
    """WritePointCloudToPLY's layout: float x y z [nx ny nz] uchar [red green blue]."""
    pts = np.asarray(points, np.float32).reshape(-1, 3)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.zeros(len(pts), np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    if normals is not None:
        nr = np.asarray(normals, np.float32).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = nr[:, 0], nr[:, 1], nr[:, 2]
    if colors is not None:
        c8 = np.clip(np.asarray(colors, np.float32).reshape(-1, 3) * 255.0, 0, 255).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    names = {"<f4": "float", "u1": "uchar"}
    head = ["ply", "format %s 1.0" % ("ascii" if ascii else "binary_little_endian"),
            "comment Created by cupoch_amd", "element vertex %d" % len(pts)]
    head += ["property %s %s" % (names[t], n) for n, t in fields] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        if ascii:
            for r in rec:
                f.write((" ".join(repr(float(v)) if isinstance(v, np.floating) else str(int(v)) for v in r)
                         + "\n").encode("ascii"))
        else:
            f.write(rec.tobytes())


def write_pcd_arrays(path, points, normals=None, colors=None, ascii=False, compressed=False):
    """PCD v0.7 with the fields PCL / cupoch write: x y z [normal_x normal_y normal_z] [rgb];
    DATA binary, ascii or binary_compressed (file_pcd.cu:628-700)."""
    pts = np.asarray(points, np.float32).reshape(-1, 3)
    fields = ["x", "y", "z"]
    cols = [pts]
    if normals is not None:
        fields += ["normal_x", "normal_y", "normal_z"]
        cols.append(np.asarray(normals, np.float32).reshape(-1, 3))
    if colors is not None:
        c8 = np.clip(np.asarray(colors, np.float32).reshape(-1, 3) * 255.0, 0, 255).astype(np.uint32)
        packed = ((c8[:, 0] << 16) | (c8[:, 1] << 8) | c8[:, 2]).astype(np.uint32).view(np.float32)
        fields.append("rgb")
        cols.append(packed.reshape(-1, 1))
    data = np.ascontiguousarray(np.concatenate(cols, 1), dtype=np.float32)
    n, k = data.shape
    head = ["# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7", "FIELDS " + " ".join(fields),
            "SIZE " + " ".join(["4"] * k), "TYPE " + " ".join(["F"] * k), "COUNT " + " ".join(["1"] * k),
            "WIDTH %d" % n, "HEIGHT 1", "VIEWPOINT 0 0 0 1 0 0 0", "POINTS %d" % n,
            "DATA " + ("ascii" if ascii else ("binary_compressed" if compressed else "binary"))]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        if ascii:
            for row in data:
                f.write((" ".join("%.10g" % v for v in row) + "\n").encode("ascii"))
        elif compressed:
            payload = np.ascontiguousarray(data.T).tobytes()          # field-major
            comp = lzf_compress(payload)
            f.write(np.array([len(comp), len(payload)], "<u4").tobytes())
            f.write(comp)
        else:
            f.write(data.tobytes())


def read_point_cloud_arrays(path):
    ext = str(path).rsplit(".", 1)[-1].lower()
    if ext == "pcd":
        return read_pcd_arrays(path)
    if ext == "ply":
        return read_ply_arrays(path)
    raise ValueError("read_point_cloud: unsupported extension .%s (pcd and ply are)" % ext)


def write_point_cloud(path, pointcloud, write_ascii=False, compressed=False):
    """cupoch.io.write_point_cloud for .pcd / .ply (pointcloud_io.h:70-75: write_ascii, compressed)"""
    def host(v):
        return None if v is None or len(v) == 0 else np.asarray(v.cpu())
    pts = host(pointcloud.points)
    nrm = host(pointcloud.normals) if pointcloud.has_normals() else None
    col = host(pointcloud.colors) if pointcloud.has_colors() else None
    ext = str(path).rsplit(".", 1)[-1].lower()
    if ext == "ply":
        write_ply_arrays(path, pts, nrm, col, ascii=write_ascii)
    elif ext == "pcd":
        write_pcd_arrays(path, pts, nrm, col, ascii=write_ascii, compressed=compressed)
    else:
        raise ValueError("write_point_cloud: unsupported extension .%s" % ext)
    return True
