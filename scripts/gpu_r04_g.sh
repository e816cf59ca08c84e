#!/bin/bash
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|error" $O/t_gpu.log | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric' | tee $O/bench.json | python scripts/benchline.py
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04g/bench.json"))["config"]["secondary"]
print({k:v for k,v in d.items() if k!="strong_100M" and k!="transient"})
PY
timeout 600 python scripts/dev/transient_census.py 2>&1 | grep -E "after" | cut -c1-330 | tee $O/transient_census.txt
timeout 600 python scripts/measure_noisy.py 2>&1 | grep '^{' | tee $O/noisy.jsonl | cut -c60-330
