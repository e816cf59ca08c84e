#!/bin/bash
# full validation + profiles for the round: all GPU tests, smoke, bench, rocprof stats, PMC traffic, configs
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_10m.json
MI_ICP_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --steps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_10m_rccl1.json
timeout 600 python scripts/measure_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/configs.jsonl; tail -3 gpurun_out/configs.jsonl
timeout 300 python scripts/nn_census.py 2>&1 | grep -v amdgpu.ids > gpurun_out/census.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1; echo "stats rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc*/p_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_packet_kernel<true' in k or 'reduce_kernel' in k or 'rs_scatter' in k:
            print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'launches', max(len(x) for x in v.values()))
PY
