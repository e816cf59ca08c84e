#!/bin/bash
# Round 6: the group-stationary search -- its parity tests, the seeded / fuzz / baseline-config suites it must not
# disturb, and the bench's secondary figures (first pass, first pass with halos, transient).
O=gpurun_out/r06c
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_first_pass.py -x -q --timeout=600 > $O/t_gs.log 2>&1; echo "gs tests rc=$?"; tail -15 $O/t_gs.log | grep -v "^E  "
timeout 900 python bench.py --no-cpu-baseline --big-points 0 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06c/bench_10m.json"))
s = d["config"]["secondary"]
for k in ("first_pass_ms", "first_pass_kind", "first_pass_with_halos_ms", "first_pass_with_halos_kind", "transient_30_iteration_loop_ms",
          "noisy_sigma_0.15_nn_ms", "noisy_sigma_0.15_it_per_s", "cold_30_iteration_call_ms"):
    print(k, s.get(k))
PY
