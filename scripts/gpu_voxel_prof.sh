#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/vox.py <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, sys.argv[1])
from bench import synth
from cupoch_amd.engine import Engine
eng = Engine(0)
src, tgt, nrm, T, md = synth(10_000_000)
d, dn = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
for _ in range(4):
    eng.voxel_downsample(d, 0.01, dn)
torch.cuda.synchronize()
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vox -o v -- python /tmp/vox.py $R > $R/gpurun_out/prof_vox.log 2>&1
cd $R
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_vox/v_kernel_stats.csv')):
    if float(r['Percentage'])>0.3: print(r['Name'][:60].ljust(62), r['Calls'].rjust(4), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
PY
