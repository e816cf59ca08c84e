#!/bin/bash
# instruction / stall counters of the search kernel on the bench workload
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcn$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmcn$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_nn_summary.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcn*/p_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_packet_kernel<true' in k or 'reduce_kernel' in k:
            print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'launches', max(len(x) for x in v.values()))
PY
