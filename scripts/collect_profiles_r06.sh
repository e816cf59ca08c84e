#!/bin/bash
# gpurun_out/r06/* (scripts/gpu_final_r06.sh) -> profiles/r06_* (the names profiles/README.md lists)
S=gpurun_out/r06; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2"; }
cpif $S/bench_10m.json $P/r06_bench_10m.json
cpif $S/bench_10m_steps20_warmup5.json $P/r06_bench_10m_steps20_warmup5.json
cpif $S/bench_by_size.jsonl $P/r06_bench_by_size.jsonl
cpif $S/bench_10m_rccl_1rank.json $P/r06_bench_10m_rccl_1rank.json
cpif $S/bench_10m_mailbox_1rank.json $P/r06_bench_10m_mailbox_1rank.json
cpif $S/bench_rehearsal_one_device.jsonl $P/r06_bench_rehearsal_one_device.jsonl
cpif $S/configs.jsonl $P/r06_configs_measured.jsonl
cpif $S/voxel.txt $P/r06_voxel.txt
cpif $S/noisy.jsonl $P/r06_noisy_workload_measured.jsonl
cpif $S/call_latency.jsonl $P/r06_call_latency.jsonl
cpif $S/call_latency_host_inputs.jsonl $P/r06_call_latency_host_inputs.jsonl
cpif $S/shard_emulation.jsonl $P/r06_shard_emulation.jsonl
cpif $S/colored.jsonl $P/r06_colored_icp_measured.jsonl
cpif $S/kinfu.jsonl $P/r06_kinfu_measured.jsonl
cpif $S/odometry.jsonl $P/r06_odometry_measured.jsonl
cpif $S/knn_search.jsonl $P/r06_knn_search_measured.jsonl
cpif $S/normals_10m.txt $P/r06_normals_10m.txt
cpif $S/config1_cpu_p2p_100k.json $P/r06_config1_cpu_p2p_100k.json
cpif $S/non_uniform_clouds.txt $P/r06_non_uniform_clouds.txt
cpif $S/transient_census.txt $P/r06_transient_census.txt
cpif $S/reference_benchmark_fragment.jsonl $P/r06_reference_benchmark_fragment.jsonl
cpif $S/shard_step_breakdown.txt $P/r06_shard_step_breakdown.txt
cpif $S/transient_trace.txt $P/r06_transient_trace.txt
cpif $S/occupancy.txt $P/r06_occupancy.txt
cpif $S/fuzz_registration_rules.json $P/r06_fuzz_registration_rules.json
cpif $S/pmc_summary.txt $P/r06_pmc_summary.txt
cpif $S/pmc_noisy_traffic.txt $P/r06_pmc_noisy_traffic.txt
cpif $S/pmc_rows_summary.txt $P/r06_pmc_rows_summary.txt
cpif gpurun_out/fetch_calibration.txt $P/r06_fetch_calibration.txt
cpif gpurun_out/nn_traffic.json $P/nn_traffic.json
for k in head cold noisy configs knn transient voxel; do
  f=$(find $S/st_$k -name 's_kernel_stats.csv' | head -1)
  case $k in head) n=r06_rocprofv3_kernel_stats.csv;; cold) n=r06_cold_call_rocprofv3_kernel_stats.csv;; *) n=r06_${k}_rocprofv3_kernel_stats.csv;; esac
  [ -n "$f" ] && cp "$f" $P/$n
done
ls -la $P | grep r06_ | wc -l
