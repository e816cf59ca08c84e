#!/bin/bash
# Round 5, run k: the sort ahead of the first search taken out again -- shard steps, the headline, the transient, the seeded tests.
O=gpurun_out/r05k
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05k/shard.jsonl"):
    r = json.loads(l)
    print({k: r[k] for k in r if k in ("ranks", "step_ms", "nn_ms", "reduce_ms", "speedup_vs_1")})
PY
timeout 600 python -m pytest tests/test_gpu_seeded.py tests/test_gpu_parity.py -m gpu -q --timeout=600 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric' | tee $O/bench.json | python scripts/benchline.py
timeout 300 python scripts/dev/transient_trace.py 2>&1 | tail -12
