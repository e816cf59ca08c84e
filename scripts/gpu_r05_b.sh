#!/bin/bash
# Round 5, second pass: locate by planes + sort + re-location (A/B against MI_ICP_NO_LOCATE_PLANES), the k-NN rows on
# resident-wave index rows, the step breakdown with sampled search stamps.
O=gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error|error" $O/t_gpu.log | tail -5
timeout 600 python bench.py --no-cpu-baseline --big-points 0 2> $O/bench.err | grep '^{"metric' | tee $O/bench_new.json | python scripts/benchline.py
python -c "
import json;d=json.load(open('$O/bench_new.json'))['config']['secondary'];print({k:v for k,v in d.items() if not isinstance(v,(dict,str)) or 'kind' in k})"
MI_ICP_NO_LOCATE_PLANES=1 timeout 600 python bench.py --no-cpu-baseline --big-points 0 2> $O/bench_old.err | grep '^{"metric' | tee $O/bench_old.json | python scripts/benchline.py
python -c "
import json;d=json.load(open('$O/bench_old.json'))['config']['secondary'];print({k:v for k,v in d.items() if not isinstance(v,(dict,str)) or 'kind' in k})"
timeout 300 python scripts/dev/transient_trace.py > $O/transient_trace_new.txt 2>&1; cut -c1-1500 $O/transient_trace_new.txt | tail -4
MI_ICP_NO_LOCATE_PLANES=1 timeout 300 python scripts/dev/transient_trace.py > $O/transient_trace_old.txt 2>&1; cut -c1-1500 $O/transient_trace_old.txt | tail -4
timeout 600 python scripts/measure_knn.py 1,0.0 8,0.0 30,0.0 30,0.01 64,0.0 100,0.0 2>$O/knn.err | grep '^{' > $O/knn_search.jsonl; cut -c1-300 $O/knn_search.jsonl
timeout 600 python scripts/measure_normals_10m.py 2>&1 | grep normals > $O/normals_10m.txt; cat $O/normals_10m.txt
timeout 600 python scripts/measure_step_breakdown.py > $O/shard_step_breakdown.txt 2> $O/breakdown.err; cat $O/shard_step_breakdown.txt | grep -v "^#"
timeout 300 python scripts/measure_noisy.py 2>/dev/null | grep '^{' > $O/noisy.jsonl; cut -c60-260 $O/noisy.jsonl
du -sh $O | tail -1
