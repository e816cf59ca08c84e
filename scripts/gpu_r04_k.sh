#!/bin/bash
# bench.py --gpus 2 rehearsed on one GPU with the in-library RCCL measured beside the mailbox (RCCL refuses two ranks on
# one device: the error / watchdog path of rccl_beside and of mi_icp_comm_init), then the plain rehearsal
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/k_build.log 2>&1; echo "build rc=$?"
MI_ICP_BENCH_ONE_DEVICE=1 MI_ICP_BENCH_RCCL_BESIDE=1 MI_ICP_COMM_INIT_MS=20000 MI_ICP_BENCH_RCCL_TIMEOUT_S=40 timeout 600 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 2 > gpurun_out/k_beside.log 2>&1; echo "beside rc=$?"
grep '^{"metric' gpurun_out/k_beside.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['config']['exchange'])[:1200])"
tail -5 gpurun_out/k_beside.log | cut -c1-300
MI_ICP_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 8 --steps 10 --warmup 3 --repeats 2 > gpurun_out/k_plain8.log 2>&1; echo "plain8 rc=$?"
grep '^{"metric' gpurun_out/k_plain8.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -3
