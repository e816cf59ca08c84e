#!/bin/bash
# counters of one kernel (name substring K) while running command CMD: instruction mix, stalls, traffic
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${K:-leaf_halo_build}
CMD=${CMD:-"python $R/scripts/dev/halo_build_time.py"}
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck$i -o p -- $CMD > $R/gpurun_out/pmck$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
K=$K python - <<'PY'
import csv, glob, collections, os
K = os.environ["K"]
for d in sorted(glob.glob('gpurun_out/pmck*/p_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if K in k:
            print(k[:40], {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'launches', max(len(x) for x in v.values()))
PY
