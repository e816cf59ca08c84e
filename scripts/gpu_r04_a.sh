#!/bin/bash
# Round 4, first GPU pass: the whole -m gpu suite + the measurements the round's first changes need
# (voxel path, mid-size fused iteration, halo build ahead of the first pass, bench incl. strong_100M, N > 1 rehearsal).
O=gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
S=${*:-tests configs shard transient bench rehearsal}
has() { [[ " $S " == *" $1 "* ]]; }
if has tests; then
  python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/t_gpu.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
  cp gpurun_out/fuzz_registration_rules.json $O/ 2>/dev/null
fi
if has configs; then
  timeout 900 python scripts/measure_configs.py 2>&1 | grep '^{' > $O/configs.jsonl; grep -i "voxel\|tree build\|staging" $O/configs.jsonl | cut -c1-220
  MI_ICP_VOXEL_OLD=1 timeout 900 python scripts/measure_configs.py 2>&1 | grep '^{' | grep -i voxel > $O/configs_voxel_old.jsonl; cut -c1-200 $O/configs_voxel_old.jsonl
fi
if has shard; then
  timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_two_kernels.jsonl; cut -c1-260 $O/shard_two_kernels.jsonl
  for w in 16 12; do
    MI_ICP_MID_WAVES=$w MI_ICP_MID_MAX=6000000 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_mid_w$w.jsonl; echo "mid waves/CU $w"; cut -c1-260 $O/shard_mid_w$w.jsonl
  done
  MI_ICP_FUSED_MAX=3000000 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_fused_small.jsonl; echo "per-packet totals (fused_small)"; cut -c1-260 $O/shard_fused_small.jsonl
  MI_ICP_SHARD_MAILBOX=1 MI_ICP_MID_MAX=6000000 timeout 600 python scripts/measure_shard.py 2>&1 | grep '^{' > $O/shard_mid_mailbox.jsonl; echo "mid + mailbox against itself"; cut -c1-260 $O/shard_mid_mailbox.jsonl
fi
if has transient; then
  timeout 300 python scripts/dev/transient_one.py 2>&1 | grep loop | tee $O/transient_new.txt
  MI_ICP_LINKS_ASYNC=1 timeout 300 python scripts/dev/transient_one.py 2>&1 | grep loop | tee $O/transient_async.txt
fi
if has bench; then
  timeout 900 python bench.py 2>&1 | grep '^{"metric' | tee $O/bench_10m.json | python scripts/benchline.py
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r04a/bench_10m.json"))
print(json.dumps(d["config"].get("secondary", {}), indent=1)[:3000])
PY
fi
if has rehearsal; then
  MI_ICP_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --repeats 2 --points 2000000 > $O/rehearsal2.log 2>&1; echo "rehearsal rc=$?"; grep '^{"metric' $O/rehearsal2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['parallelism']); print(json.dumps(d['config']['exchange']))"; tail -3 $O/rehearsal2.log | cut -c1-300
fi
