#!/bin/bash
# the point-to-plane reduction on the 10M bench: the memory-parallel kernel against the generic one
# (the grid / unroll sweep that picked 512 blocks x 4 elements in flight needs the knobs of commit history)
echo -n "reduce_pt2pl_kernel<4>: "; python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>&1 | python scripts/benchline.py
echo -n "generic reduce_kernel:  "; MI_ICP_NO_FAST_REDUCE=1 python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>&1 | python scripts/benchline.py
echo -n "... without 24-B records: "; MI_ICP_NO_TREC=1 python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>&1 | python scripts/benchline.py
