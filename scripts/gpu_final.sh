#!/bin/bash
# The round's full validation + every measurement that goes into profiles/ (run on the MI355X box:
# gpurun -- scripts/gpu_final.sh).  Everything lands under gpurun_out/final/.
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric' | tee $O/bench_10m_driver_flags.json | python scripts/benchline.py
MI_ICP_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_rccl_1rank.json | python scripts/benchline.py
for n in 100000 1000000 5000000 20000000; do timeout 600 python bench.py --points $n --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric'; done > $O/bench_by_size.jsonl
timeout 900 python scripts/measure_configs.py 2>&1 | grep '^{' > $O/configs.jsonl; tail -2 $O/configs.jsonl | cut -c1-200
timeout 600 python scripts/measure_noisy.py 2>&1 | grep '^{' > $O/noisy.jsonl
timeout 600 python scripts/measure_latency.py 2>&1 | grep '^{' > $O/call_latency.jsonl
MI_ICP_LATENCY_HOST=1 timeout 600 python scripts/measure_latency.py 100000 1000000 10000000 2>&1 | grep '^{' > $O/call_latency_host_inputs.jsonl
timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_emulation.jsonl
MI_ICP_SHARD_MAILBOX=1 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' >> $O/shard_emulation.jsonl
MI_ICP_FORCE_COMM=2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_mailbox_1rank.json | python scripts/benchline.py
timeout 600 python scripts/measure_colored.py 2>&1 | grep '^{' > $O/colored.jsonl
timeout 600 python scripts/measure_kinfu.py 2>&1 | grep '^{' > $O/kinfu.jsonl
timeout 600 python scripts/measure_odometry.py 2>&1 | grep '^{' > $O/odometry.jsonl
timeout 600 python scripts/measure_links_payoff.py 2>&1 | grep '^{' > $O/links_payoff.jsonl; MI_ICP_NO_LINKS=1 timeout 600 python scripts/measure_links_payoff.py 2>&1 | grep '^{' >> $O/links_payoff.jsonl
timeout 600 python scripts/measure_knn.py 1,0.0 8,0.0 30,0.0 30,0.01 64,0.0 100,0.0 100,0.02 2>&1 | grep '^{' > $O/knn_search.jsonl
timeout 600 scripts/gpu_reduce_sweep.sh > $O/reduce_variants.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -o r02 -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/rocprof_stats.log 2>&1; echo "stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cold -o r02cold -- python $R/scripts/measure_latency.py 10000000 > $R/$O/rocprof_cold.log 2>&1; echo "cold stats rc=$?"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc$i -o p -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $R/$O/pmc$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
python - <<'PY' | tee gpurun_out/final/pmc_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes of: python bench.py --steps 6 --warmup 2 --repeats 1 (10M-vs-10M point-to-plane); averages per launch")
for d in sorted(glob.glob('gpurun_out/final/pmc*/p_counter_collection.csv')):
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'nn_packet_kernel<true' in k or 'reduce_pt2pl' in k:
            print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, 'launches', max(len(x) for x in v.values()))
PY
scripts/gpu_traffic.sh > $O/traffic.log 2>&1; tail -6 $O/traffic.log
