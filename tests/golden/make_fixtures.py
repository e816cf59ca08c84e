#!/usr/bin/env python3
"""Regenerates the real-data fixtures from the reference's sample files (needs
/root/reference; run from the repo root).  Only sub-sampled DATA is committed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cupoch_amd.io import read_pcd_arrays, read_ply_arrays   # noqa: E402

REF = "/root/reference/examples/testdata"
OUT = os.path.dirname(os.path.abspath(__file__))

a = read_pcd_arrays(os.path.join(REF, "fragment.pcd"))
sel = np.flatnonzero(np.isfinite(a["points"]).all(1))[::3]
np.savez_compressed(os.path.join(OUT, "fragment_every3rd.npz"), points=a["points"][sel],
                    normals=a["normals"][sel].astype(np.float16))

# the WHOLE scan, points only (113,662 x 3 float32): the input of the reference's one published benchmark
# (examples/python/basic/benchmarks.py; its README chart) -- scripts/measure_reference_benchmark.py times the same
# four calls on it on the GPU box, where /root/reference does not exist
full = a["points"][np.isfinite(a["points"]).all(1)]
np.savez_compressed(os.path.join(OUT, "fragment_points.npz"), points=full)

# coloured RGB-D fragment of the reference's colored-ICP example
# (examples/python/advanced/colored_pointcloud_registration.py): every 2nd vertex
b = read_ply_arrays(os.path.join(REF, "colored_icp", "frag_115.ply"))
np.savez_compressed(os.path.join(OUT, "frag115_every2nd.npz"), points=b["points"][::2],
                    colors=np.round(b["colors"][::2] * 255.0).astype(np.uint8))
print("ok")
