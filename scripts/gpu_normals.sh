#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "normal or gradient or colored or golden" 2>&1 | tail -3
python - <<'PY'
import sys, time, json, numpy as np, torch
sys.path.insert(0,'.')
from bench import synth
from cupoch_amd.engine import Engine
eng=Engine(0)
src,tgt,nrm,T,md=synth(10_000_000)
d=torch.from_numpy(tgt).cuda()
for n,k in ((2_000_000,30),(2_000_000,20),(10_000_000,30)):
    eng.estimate_normals_knn(d[:n],k); torch.cuda.synchronize()
    t0=time.perf_counter(); eng.estimate_normals_knn(d[:n],k); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(json.dumps({"row":"EstimateNormals(KNN %d)"%k,"n":n,"ms":dt*1e3,"Mpts_per_s":n/dt/1e6}))
t0=time.perf_counter(); eng.estimate_normals_radius(d[:2_000_000], 0.02, 30); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print(json.dumps({"row":"EstimateNormals(radius 0.02, max_nn 30)","n":2_000_000,"ms":dt*1e3}))
PY
