#!/usr/bin/env python3
"""Where kd_build_groups' time goes: set_target of a 10M cloud with normals, five times (run under rocprofv3 --kernel-trace
--stats with MI_ICP_LIB_PATH pointing at a variant library built with -DMI_AB_NO_SORT / -DMI_AB_NO_WRITE / -DMI_AB_COHERENT;
the variants' trees are WRONG -- timing only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cupoch_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
tgt = torch.rand((n, 3), generator=g, device="cuda")
nrm = torch.randn((n, 3), generator=g, device="cuda"); nrm /= torch.linalg.norm(nrm, dim=1, keepdim=True)
e = Engine(0)
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.set_target(tgt, nrm); e.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("set_target ms:", " ".join("%.3f" % t for t in ts[1:]), "lib", os.environ.get("MI_ICP_LIB_PATH", "default"))
