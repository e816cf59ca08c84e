#!/bin/bash
# Round 6: the validation + every measurement that goes into profiles/r06_* -- ONE pass on the code as committed
# (gpurun -- scripts/gpu_final_r06.sh [sections]).  Everything lands under gpurun_out/r06/; scripts/collect_profiles_r06.sh
# copies it into profiles/ under the names profiles/README.md lists.
# sections (default: all): tests traffic bench rows refbench rehearsal stats pmc pmcrows
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${*:-tests traffic bench rows refbench rehearsal stats pmc pmcrows}
has() { [[ " $S " == *" $1 "* ]]; }
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
if has tests; then
  python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
  cp gpurun_out/fuzz_registration_rules.json $O/ 2>/dev/null
  python scripts/dev/occupancy.py 2>&1 | grep occupancy > $O/occupancy.txt
fi
# (the traffic section runs ahead of the bench: bench.py quotes profiles/nn_traffic.json, which this section writes)
if has traffic; then
  scripts/gpu_traffic.sh > $O/traffic.log 2>&1; tail -6 $O/traffic.log
  [ -s gpurun_out/nn_traffic.json ] && cp gpurun_out/nn_traffic.json profiles/nn_traffic.json   # (this box's copy: what the bench section quotes)
  for c in FETCH_SIZE WRITE_SIZE; do
    prof --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_noisy_$c -o p -- python $R/scripts/dev/noisy_one.py 0.15 > $O/pmc_noisy_$c.log 2>&1; echo "pmc noisy $c rc=$?"
  done
  { echo "# FETCH_SIZE / WRITE_SIZE (KB per launch as reported: vector loads count at half their bytes, profiles/r06_fetch_calibration.txt) on the noisy workload"
    python scripts/pmc_kernels.py "$O/pmc_noisy_*SIZE/p_counter_collection.csv" "nn_packet_kernel<true" leaf_halo; } | tee $O/pmc_noisy_traffic.txt
fi
if has bench; then
  timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
  timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric' | tee $O/bench_10m_steps20_warmup5.json | python scripts/benchline.py
  for n in 100000 1000000 5000000 20000000 50000000 100000000; do timeout 600 python bench.py --points $n --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric'; done > $O/bench_by_size.jsonl
  MI_ICP_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_rccl_1rank.json | python scripts/benchline.py
  MI_ICP_FORCE_COMM=2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_mailbox_1rank.json | python scripts/benchline.py
fi
if has rows; then
  timeout 900 python scripts/measure_configs.py 2>&1 | grep '^{' > $O/configs.jsonl; tail -2 $O/configs.jsonl | cut -c1-200
  { timeout 300 python scripts/dev/voxel_one.py; timeout 300 python scripts/dev/voxel_one.py 1000000 0.02; } 2>&1 | grep voxel > $O/voxel.txt; cat $O/voxel.txt
  timeout 600 python scripts/measure_noisy.py 2>&1 | grep '^{' > $O/noisy.jsonl; cut -c60-180 $O/noisy.jsonl
  timeout 600 python scripts/measure_latency.py 2>&1 | grep '^{' > $O/call_latency.jsonl
  MI_ICP_LATENCY_HOST=1 timeout 600 python scripts/measure_latency.py 100000 1000000 10000000 2>&1 | grep '^{' > $O/call_latency_host_inputs.jsonl
  timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_emulation.jsonl
  MI_ICP_MAILBOX=host MI_ICP_SHARD_MAILBOX=1 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' >> $O/shard_emulation.jsonl
  MI_ICP_MAILBOX=device MI_ICP_SHARD_MAILBOX=1 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' >> $O/shard_emulation.jsonl
  MI_ICP_FUSED_MAX=3000000 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' | sed 's/"exchange": "none"/"exchange": "none", "iteration": "ONE launch with per-packet totals (icp_small_iteration_kernel, MI_ICP_FUSED_MAX=3000000)"/' >> $O/shard_emulation.jsonl
  timeout 600 python scripts/measure_colored.py 2>&1 | grep '^{' > $O/colored.jsonl
  timeout 600 python scripts/measure_kinfu.py 2>&1 | grep '^{' > $O/kinfu.jsonl
  timeout 600 python scripts/measure_odometry.py 2>&1 | grep '^{' > $O/odometry.jsonl
  timeout 600 python scripts/measure_knn.py 1,0.0 8,0.0 30,0.0 30,0.01 64,0.0 100,0.0 2>&1 | grep '^{' > $O/knn_search.jsonl
  timeout 600 python scripts/measure_normals_10m.py 2>&1 | grep normals > $O/normals_10m.txt
  timeout 900 python scripts/measure_config1.py > $O/config1_cpu_p2p_100k.json 2>/dev/null; cut -c1-300 $O/config1_cpu_p2p_100k.json
  { timeout 300 python scripts/dev/normals_clustered.py; timeout 300 python scripts/dev/icp_outliers.py; timeout 300 python scripts/dev/first_pass_stats.py; } 2>&1 | grep -v amdgpu.ids > $O/non_uniform_clouds.txt
  timeout 600 python scripts/dev/transient_census.py 2>&1 | grep -E "after" > $O/transient_census.txt; cat $O/transient_census.txt | cut -c1-200
fi
if has refbench; then
  timeout 900 python scripts/measure_reference_benchmark.py > $O/reference_benchmark_fragment.jsonl 2> $O/reference_benchmark.err; echo "refbench rc=$?"
  timeout 600 python scripts/measure_step_breakdown.py > $O/shard_step_breakdown.txt 2> $O/breakdown.err; echo "breakdown rc=$?"
  { timeout 300 python scripts/dev/transient_trace.py; MI_ICP_NO_LOCATE_PLANES=1 timeout 300 python scripts/dev/transient_trace.py; } > $O/transient_trace.txt 2>&1; echo "trace rc=$?"
fi
if has rehearsal; then
  for w in 2 8; do
    MI_ICP_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2951$w bench.py --gpus $w --steps 10 --warmup 3 --repeats 2 > $O/rehearsal$w.log 2>&1; echo "rehearsal $w rc=$?"
    grep '^{"metric' $O/rehearsal$w.log
  done | grep "^{" > $O/bench_rehearsal_one_device.jsonl
  # ... and with the in-library RCCL measured beside the mailbox: RCCL refuses a second rank on one device -- the error path
  MI_ICP_BENCH_ONE_DEVICE=1 MI_ICP_BENCH_RCCL_BESIDE=1 MI_ICP_COMM_INIT_MS=20000 MI_ICP_BENCH_RCCL_TIMEOUT_S=40 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 2 > $O/rehearsal2b.log 2>&1; echo "rehearsal 2 + rccl beside rc=$?"
  grep '^{"metric' $O/rehearsal2b.log >> $O/bench_rehearsal_one_device.jsonl
  cut -c1-400 $O/bench_rehearsal_one_device.jsonl
fi
if has stats; then
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_head -o s -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/st_head.log 2>&1; echo "stats head rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_cold -o s -- python $R/scripts/measure_latency.py 10000000 > $O/st_cold.log 2>&1; echo "stats cold rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_noisy -o s -- python $R/scripts/dev/noisy_one.py 0.15 > $O/st_noisy.log 2>&1; echo "stats noisy rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_configs -o s -- python $R/scripts/measure_configs.py > $O/st_configs.log 2>&1; echo "stats configs rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_knn -o s -- python $R/scripts/measure_knn.py 30,0.0 100,0.0 > $O/st_knn.log 2>&1; echo "stats knn rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_transient -o s -- python $R/scripts/dev/transient_one.py > $O/st_transient.log 2>&1; echo "stats transient rc=$?"
  prof --kernel-trace --stats --output-format csv -d $R/$O/st_voxel -o s -- python $R/scripts/dev/voxel_one.py > $O/st_voxel.log 2>&1; echo "stats voxel rc=$?"
fi
if has pmc; then
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    prof --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_head$i -o p -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $O/pmc_head$i.log 2>&1; echo "pmc head $i rc=$?"
    [ $i -le 3 ] && { prof --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_noisy$i -o p -- python $R/scripts/dev/noisy_one.py 0.15 > $O/pmc_noisy$i.log 2>&1; echo "pmc noisy $i rc=$?"; }
  done
  { echo "# rocprofv3 --pmc passes of: python bench.py --steps 6 --warmup 2 --repeats 1 (10M-vs-10M point-to-plane, exact correspondences); averages per launch"
    python scripts/pmc_kernels.py "$O/pmc_head*/p_counter_collection.csv" "nn_packet_kernel<true" reduce_pt2pl
    echo "# the same counters on the noisy workload (scripts/dev/noisy_one.py 0.15: 10M target, 6M noisy source, sigma = 0.15 spacings, converged iterations with halos)"
    python scripts/pmc_kernels.py "$O/pmc_noisy*/p_counter_collection.csv" "nn_packet_kernel<true" leaf_halo; } | tee $O/pmc_summary.txt
fi
if has pmcrows; then
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    prof --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_rows$i -o p -- python $R/scripts/measure_configs.py > $O/pmc_rows$i.log 2>&1; echo "pmc rows $i rc=$?"
    [ $i -le 2 ] && { prof --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_knn$i -o p -- python $R/scripts/measure_knn.py 30,0.0 100,0.0 > $O/pmc_knn$i.log 2>&1; echo "pmc knn $i rc=$?"; }
  done
  { echo "# rocprofv3 --pmc passes of scripts/measure_configs.py (config 2, config 5 = GICP 5M, builds and one-time costs at 10M, VoxelDownSample, EstimateNormals 2M) and scripts/measure_knn.py 30,0.0 100,0.0; averages per launch"
    python scripts/pmc_kernels.py "$O/pmc_rows*/p_counter_collection.csv" reduce_kernel kd_build_groups cells_ voxel vox_ vx_ rs_scatter knn_normals transform_cloud cov_from tree_scale
    python scripts/pmc_kernels.py "$O/pmc_knn*/p_counter_collection.csv" knn_search; } | tee $O/pmc_rows_summary.txt
fi
# keep the merge small: rocprofv3's databases are not needed
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
du -sh $O | tail -1
