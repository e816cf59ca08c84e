#!/bin/bash
# Round 5, run q: the halo_want counter in 1024 words + split planes halfway between the halves.  Tests of the tree and
# the seeded search; the cold call with re-location until the halos are declined / with halos only; the census of its
# second search; the transient and the noisy loops.
O=gpurun_out/r05q
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
{
for v in new old new old; do
  if [ $v = old ]; then export MI_ICP_RELOCATE_HALOS_ONLY=1; else unset MI_ICP_RELOCATE_HALOS_ONLY; fi
  timeout 200 python scripts/dev/cold_trace.py 10000000
done
unset MI_ICP_RELOCATE_HALOS_ONLY
} 2>&1 | grep -v amdgpu.ids | tee $O/cold_trace.txt
MI_ICP_NO_LINKS=1 timeout 300 python scripts/dev/cold_census.py 10000000 2>&1 | grep -v amdgpu.ids | tee $O/cold_census.txt
timeout 300 python scripts/dev/transient_trace.py 10000000 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee $O/transient_trace.txt
timeout 300 python scripts/measure_noisy.py 2>&1 | grep -v amdgpu.ids | tee $O/noisy.txt
