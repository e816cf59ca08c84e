#!/bin/bash
# Round 5, last: after the re-location gate moved to a leaf's width -- all GPU tests, smoke, both headline lines, the call
# latencies, the reference's benchmark, KinFu (the rows that gate can move), on the code as committed; same layout as
# scripts/gpu_final_r05.sh (gpurun_out/r05/ -> scripts/collect_profiles_r05.sh).
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
cp gpurun_out/fuzz_registration_rules.json $O/ 2>/dev/null
timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric' | tee $O/bench_10m_steps20_warmup5.json | python scripts/benchline.py
timeout 600 python scripts/measure_latency.py 2>&1 | grep '^{' > $O/call_latency.jsonl; cut -c60-200 $O/call_latency.jsonl
timeout 600 python scripts/measure_kinfu.py 2>&1 | grep '^{' > $O/kinfu.jsonl
timeout 900 python scripts/measure_reference_benchmark.py > $O/reference_benchmark_fragment.jsonl 2> $O/reference_benchmark.err; echo "refbench rc=$?"; cut -c1-60 $O/reference_benchmark_fragment.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05/reference_benchmark_fragment.jsonl"):
    d = json.loads(l); print(d["call"], d["gpu_ms"], d["speedup_vs_1_thread"], d["speedup_vs_best_cpu"], d["parity_ok"])
PY
