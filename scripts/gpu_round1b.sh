#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=400 > gpurun_out/t_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/t_parity.log
timeout 300 python scripts/nn_census.py > gpurun_out/census.log 2>&1; cat gpurun_out/census.log | grep -v amdgpu.ids
rocprofv3 -L > gpurun_out/counters.txt 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_csv -o r01 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_csv.log 2>&1; echo "stats rc=$?"
cd $R; find gpurun_out -name "*.csv" | head -30
