"""Does a source with NaN / inf / far-away points survive set_source and the search?  argv: kinds (far,nan,inf) radius"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cupoch_amd.engine import Engine
kinds = sys.argv[1].split(",")
radius = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n_src = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
rng = np.random.default_rng(10)
tgt = rng.random((300_000, 3), dtype=np.float32)
src = rng.random((n_src, 3), dtype=np.float32)
if "far" in kinds: src[:500] = src[:500] * 50 - 25
if "nan" in kinds: src[500:520] = np.nan
if "inf" in kinds: src[520:540] = np.inf
eng = Engine(0)
eng.set_target(tgt); eng.synchronize(); print("target ok", flush=True)
eng.set_source(src); eng.synchronize(); print("source ok", flush=True)
idx, d2, st = eng.search_radius_1nn(radius)
print(kinds, radius, "search ok kind", eng.last_search_kind(), st, flush=True)
