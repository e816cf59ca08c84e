#!/bin/bash
# Round 5, late: the rows of profiles/r05_* that the round's last changes move (the halo_want counter in 1024 words, split
# planes halfway between the halves, od_accumulate's rows), re-measured in ONE pass on the code as committed; same layout
# as scripts/gpu_final_r05.sh (gpurun_out/r05/ -> scripts/collect_profiles_r05.sh).  Everything else under profiles/r05_*
# stays from that script's pass: its kernels did not change.
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
cp gpurun_out/fuzz_registration_rules.json $O/ 2>/dev/null
timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric' | tee $O/bench_10m_steps20_warmup5.json | python scripts/benchline.py
timeout 900 python scripts/measure_configs.py 2>&1 | grep '^{' > $O/configs.jsonl; tail -2 $O/configs.jsonl | cut -c1-200
timeout 600 python scripts/measure_latency.py 2>&1 | grep '^{' > $O/call_latency.jsonl; cut -c1-200 $O/call_latency.jsonl
MI_ICP_LATENCY_HOST=1 timeout 600 python scripts/measure_latency.py 100000 1000000 10000000 2>&1 | grep '^{' > $O/call_latency_host_inputs.jsonl
timeout 600 python scripts/measure_odometry.py 2>&1 | grep '^{' > $O/odometry.jsonl; cut -c1-140 $O/odometry.jsonl
timeout 600 python scripts/measure_kinfu.py 2>&1 | grep '^{' > $O/kinfu.jsonl
timeout 900 python scripts/measure_reference_benchmark.py > $O/reference_benchmark_fragment.jsonl 2> $O/reference_benchmark.err; echo "refbench rc=$?"
{ timeout 300 python scripts/dev/transient_trace.py; MI_ICP_NO_LOCATE_PLANES=1 timeout 300 python scripts/dev/transient_trace.py; } > $O/transient_trace.txt 2>&1; echo "trace rc=$?"
prof --kernel-trace --stats --output-format csv -d $R/$O/st_cold -o s -- python $R/scripts/measure_latency.py 10000000 > $O/st_cold.log 2>&1; echo "stats cold rc=$?"
prof --kernel-trace --stats --output-format csv -d $R/$O/st_transient -o s -- python $R/scripts/dev/transient_one.py > $O/st_transient.log 2>&1; echo "stats transient rc=$?"
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*trace.csv" -size +8M -delete 2>/dev/null
du -sh $O | tail -1
