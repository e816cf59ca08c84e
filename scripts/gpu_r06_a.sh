#!/bin/bash
# Round 6, first call: the tree as round 5 left it -- all GPU tests, smoke, the headline line (same-round baseline for
# every A/B that follows).
O=gpurun_out/r06a
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
