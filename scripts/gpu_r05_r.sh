#!/bin/bash
# Round 5, run r: the reduction's two-level ticket against the flat one (-DMI_AB_FLAT_TICKET variant library), same box,
# alternating: headline bench and the shard emulation; the odometry with rows instead of atomics + the wave solve; all GPU tests.
O=gpurun_out/r05r
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -5 | tee $O/tests.txt
{
for v in default flat default flat; do
  L=$R/cupoch_amd/lib/libmi_icp_$v.so; [ $v = default ] && L=$R/cupoch_amd/lib/libmi_icp.so
  echo "== $v: bench"
  MI_ICP_LIB_PATH=$L timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | python scripts/benchline.py 2>/dev/null || true
  echo "== $v: shards"
  MI_ICP_LIB_PATH=$L timeout 200 python scripts/measure_shard.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
done
} 2>&1 | tee $O/ticket_ab.txt
timeout 300 python scripts/measure_odometry.py 2>&1 | grep -v amdgpu.ids | tee $O/odometry.txt
