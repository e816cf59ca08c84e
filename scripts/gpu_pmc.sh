#!/bin/bash
# PMC passes over the bench workload (each counter set in its own run, kernel-trace only)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_MISSES_DUPLICATE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc*/p_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_packet_kernel<true' in k or 'reduce_kernel' in k:
            print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
