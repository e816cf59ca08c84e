#!/bin/bash
# gpu_final.sh without the PMC / traffic / sweep passes (those kernels did not change): full validation and
# the measurements a change to the loop's step or the reduction's finish moves.  Lands in gpurun_out/final/.
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/t_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric' | tee $O/bench_10m_driver_flags.json | python scripts/benchline.py
for n in 100000 1000000 5000000 20000000; do timeout 600 python bench.py --points $n --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric'; done > $O/bench_by_size.jsonl
timeout 600 python scripts/measure_latency.py 2>&1 | grep '^{' > $O/call_latency.jsonl
MI_ICP_LATENCY_HOST=1 timeout 600 python scripts/measure_latency.py 100000 1000000 10000000 2>&1 | grep '^{' > $O/call_latency_host_inputs.jsonl
timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_emulation.jsonl
MI_ICP_SHARD_MAILBOX=1 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' >> $O/shard_emulation.jsonl
MI_ICP_FORCE_COMM=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_rccl_1rank.json | python scripts/benchline.py
MI_ICP_FORCE_COMM=2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | grep '^{"metric' | tee $O/bench_10m_mailbox_1rank.json | python scripts/benchline.py
timeout 600 python scripts/measure_colored.py 2>&1 | grep '^{' > $O/colored.jsonl
timeout 600 python scripts/measure_kinfu.py 2>&1 | grep '^{' > $O/kinfu.jsonl
timeout 600 python scripts/measure_odometry.py 2>&1 | grep '^{' > $O/odometry.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -o r02 -- python $R/bench.py --no-cpu-baseline --no-secondary > $R/$O/rocprof_stats.log 2>&1; echo "stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cold -o r02cold -- python $R/scripts/measure_latency.py 10000000 > $R/$O/rocprof_cold.log 2>&1; echo "cold stats rc=$?"
cd $R
# rehearsal of the N > 1 bench path on this one GPU (every rank on cuda:0; gloo for torch's own collectives)
for n in 2 4; do MI_ICP_BENCH_ONE_DEVICE=1 timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 20 --warmup 5 2>&1 | grep '^{"metric'; done > $O/bench_rehearsal_one_device.jsonl
python scripts/benchline.py < $O/bench_rehearsal_one_device.jsonl
