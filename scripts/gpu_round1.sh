#!/bin/bash
# first GPU session: build check, primitives, parity, smoke, bench, rocprof
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt
timeout 600 python -m pytest tests/test_gpu_0_primitives.py -m gpu -q --timeout=300 > gpurun_out/t_prim.log 2>&1; echo "prim rc=$?"
tail -5 gpurun_out/t_prim.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=400 > gpurun_out/t_parity.log 2>&1; echo "parity rc=$?"
tail -25 gpurun_out/t_parity.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 300 python bench.py --points 1000000 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1m.log 2>&1; echo "bench1m rc=$?"; tail -2 gpurun_out/bench_1m.log
timeout 600 python bench.py > gpurun_out/bench_10m.log 2>&1; echo "bench10m rc=$?"; tail -2 gpurun_out/bench_10m.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; 
