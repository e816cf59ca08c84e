#!/bin/bash
# gpurun_out/r05/* (scripts/gpu_final_r05.sh) -> profiles/r05_* (the names profiles/README.md lists)
S=gpurun_out/r05; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2"; }
cpif $S/bench_10m.json $P/r05_bench_10m.json
cpif $S/bench_10m_steps20_warmup5.json $P/r05_bench_10m_steps20_warmup5.json
cpif $S/bench_by_size.jsonl $P/r05_bench_by_size.jsonl
cpif $S/bench_10m_rccl_1rank.json $P/r05_bench_10m_rccl_1rank.json
cpif $S/bench_10m_mailbox_1rank.json $P/r05_bench_10m_mailbox_1rank.json
cpif $S/bench_rehearsal_one_device.jsonl $P/r05_bench_rehearsal_one_device.jsonl
cpif $S/configs.jsonl $P/r05_configs_measured.jsonl
cpif $S/voxel.txt $P/r05_voxel.txt
cpif $S/noisy.jsonl $P/r05_noisy_workload_measured.jsonl
cpif $S/call_latency.jsonl $P/r05_call_latency.jsonl
cpif $S/call_latency_host_inputs.jsonl $P/r05_call_latency_host_inputs.jsonl
cpif $S/shard_emulation.jsonl $P/r05_shard_emulation.jsonl
cpif $S/colored.jsonl $P/r05_colored_icp_measured.jsonl
cpif $S/kinfu.jsonl $P/r05_kinfu_measured.jsonl
cpif $S/odometry.jsonl $P/r05_odometry_measured.jsonl
cpif $S/knn_search.jsonl $P/r05_knn_search_measured.jsonl
cpif $S/normals_10m.txt $P/r05_normals_10m.txt
cpif $S/config1_cpu_p2p_100k.json $P/r05_config1_cpu_p2p_100k.json
cpif $S/non_uniform_clouds.txt $P/r05_non_uniform_clouds.txt
cpif $S/transient_census.txt $P/r05_transient_census.txt
cpif $S/reference_benchmark_fragment.jsonl $P/r05_reference_benchmark_fragment.jsonl
cpif $S/shard_step_breakdown.txt $P/r05_shard_step_breakdown.txt
cpif $S/transient_trace.txt $P/r05_transient_trace.txt
cpif $S/occupancy.txt $P/r05_occupancy.txt
cpif $S/fuzz_registration_rules.json $P/r05_fuzz_registration_rules.json
cpif $S/pmc_summary.txt $P/r05_pmc_summary.txt
cpif $S/pmc_noisy_traffic.txt $P/r05_pmc_noisy_traffic.txt
cpif $S/pmc_rows_summary.txt $P/r05_pmc_rows_summary.txt
cpif gpurun_out/fetch_calibration.txt $P/r05_fetch_calibration.txt
cpif gpurun_out/nn_traffic.json $P/nn_traffic.json
for k in head cold noisy configs knn transient voxel; do
  f=$(find $S/st_$k -name 's_kernel_stats.csv' | head -1)
  case $k in head) n=r05_rocprofv3_kernel_stats.csv;; cold) n=r05_cold_call_rocprofv3_kernel_stats.csv;; *) n=r05_${k}_rocprofv3_kernel_stats.csv;; esac
  [ -n "$f" ] && cp "$f" $P/$n
done
ls -la $P | grep r05_ | wc -l
