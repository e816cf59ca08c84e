#!/bin/bash
# The PMC passes of gpu_final.sh on their own (instruction mix, waits, L2 hits, busy cycles of the search and
# reduction kernels, per launch) -> gpurun_out/final/pmc_summary.txt
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc$i -o p -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $R/$O/pmc$i.log 2>&1; echo "pmc$i rc=$? : $set"
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
