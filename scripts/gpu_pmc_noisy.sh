#!/bin/bash
# instruction / traffic counters of the seeded search kernel on the noisy workload: SIGMA (default 0.15)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${SIGMA:-0.15}
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcz$i -o p -- python $R/scripts/dev/noisy_one.py $S > $R/gpurun_out/pmcz$i.log 2>&1; echo "pmc$i rc=$? : $set"
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_noisy_summary_${S}.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcz*/p_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_packet_kernel<true' in k:
            print(k, {c: round(sum(x[2:])/max(len(x[2:]),1),1) for c,x in v.items()}, 'launches', max(len(x) for x in v.values()))
PY
