#!/bin/bash
# Round 5, first pass on the GPU: the split library + this round's test changes; the reference's own benchmark; a bench
# line on this round's box; where a shard's step goes.
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
cp gpurun_out/fuzz_registration_rules.json $O/ 2>/dev/null
timeout 900 python scripts/measure_reference_benchmark.py > $O/reference_benchmark_fragment.jsonl 2> $O/reference_benchmark.err; echo "refbench rc=$?"; cut -c1-400 $O/reference_benchmark_fragment.jsonl; tail -3 $O/reference_benchmark.err
timeout 900 python bench.py 2> $O/bench.err | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
timeout 600 python scripts/measure_step_breakdown.py > $O/shard_step_breakdown.txt 2> $O/breakdown.err; echo "breakdown rc=$?"; cat $O/shard_step_breakdown.txt; tail -3 $O/breakdown.err
timeout 600 python scripts/measure_shard.py 2>$O/shard.err | grep '^{' > $O/shard_emulation.jsonl; cat $O/shard_emulation.jsonl
du -sh $O | tail -1
