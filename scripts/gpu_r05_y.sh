#!/bin/bash
# Round 5, run y: the fuzz soak (MI_ICP_FUZZ_CASES=300: 300 search + 100 registration + 120 voxel cases) with the tree-invariant
# and seeded-search tests, on the round's final tree (the halo_want counter in 1024 words, split planes between the halves).
O=gpurun_out/r05y
mkdir -p $O
export TMPDIR=/tmp
MI_ICP_FUZZ_CASES=300 timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_tree_invariants.py tests/test_gpu_seeded.py -m gpu -q --timeout=600 > $O/soak.log 2>&1
grep -E "passed|failed|error" $O/soak.log | tail -3 | tee $O/soak.txt
cp gpurun_out/fuzz_registration_rules.json $O/fuzz_registration_rules_soak100.json 2>/dev/null
