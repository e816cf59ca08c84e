#!/bin/bash
# gpurun_out/r03/* (scripts/gpu_final_r03.sh) -> profiles/r03_* (the names profiles/README.md lists)
S=gpurun_out/r03; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2"; }
cpif $S/bench_10m.json $P/r03_bench_10m.json
cpif $S/bench_10m_steps20_warmup5.json $P/r03_bench_10m_steps20_warmup5.json
cpif $S/bench_by_size.jsonl $P/r03_bench_by_size.jsonl
cpif $S/bench_10m_rccl_1rank.json $P/r03_bench_10m_rccl_1rank.json
cpif $S/bench_10m_mailbox_1rank.json $P/r03_bench_10m_mailbox_1rank.json
cpif $S/configs.jsonl $P/r03_configs_measured.jsonl
cpif $S/noisy.jsonl $P/r03_noisy_workload_measured.jsonl
cpif $S/call_latency.jsonl $P/r03_call_latency.jsonl
cpif $S/call_latency_host_inputs.jsonl $P/r03_call_latency_host_inputs.jsonl
cpif $S/shard_emulation.jsonl $P/r03_shard_emulation.jsonl
cpif $S/colored.jsonl $P/r03_colored_icp_measured.jsonl
cpif $S/kinfu.jsonl $P/r03_kinfu_measured.jsonl
cpif $S/odometry.jsonl $P/r03_odometry_measured.jsonl
cpif $S/knn_search.jsonl $P/r03_knn_search_measured.jsonl
cpif $S/normals_10m.txt $P/r03_normals_10m.txt
cpif $S/config1_cpu_p2p_100k.json $P/r03_config1_cpu_p2p_100k.json
cpif $S/non_uniform_clouds.txt $P/r03_non_uniform_clouds.txt
for k in head cold noisy configs knn transient; do
  f=$(find $S/st_$k -name 's_kernel_stats.csv' | head -1)
  case $k in head) n=r03_rocprofv3_kernel_stats.csv;; cold) n=r03_cold_call_rocprofv3_kernel_stats.csv;; *) n=r03_${k}_rocprofv3_kernel_stats.csv;; esac
  [ -n "$f" ] && cp "$f" $P/$n
done
ls -la $P | grep r03_ | wc -l
