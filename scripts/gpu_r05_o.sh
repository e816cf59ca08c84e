#!/bin/bash
# Round 5, run o: re-location ahead of the first seeded searches WITHOUT halos (until the loop declines them) against
# round 5's first rule (halos only), same box, alternating: the bench's cold call and its first searches one by one.
O=gpurun_out/r05o
mkdir -p $O
export TMPDIR=/tmp
{
for v in new old new old; do
  if [ $v = old ]; then export MI_ICP_RELOCATE_HALOS_ONLY=1; else unset MI_ICP_RELOCATE_HALOS_ONLY; fi
  timeout 200 python scripts/dev/cold_trace.py 10000000
done
unset MI_ICP_RELOCATE_HALOS_ONLY
echo "== 1M"; timeout 100 python scripts/dev/cold_trace.py 1000000
echo "== 1M old"; MI_ICP_RELOCATE_HALOS_ONLY=1 timeout 100 python scripts/dev/cold_trace.py 1000000
} 2>&1 | tee $O/cold_trace.txt
timeout 600 python -m pytest tests/test_gpu_seeded.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
