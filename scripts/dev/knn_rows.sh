O=gpurun_out/r03b; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q --timeout=600 -k "knn or normal or colored or gicp or kdtree" 2>&1 | tail -2
timeout 600 python scripts/measure_configs.py 2>&1 | grep '^{' > $O/configs.jsonl; grep -i "normals\|config5\|GICP" $O/configs.jsonl | cut -c1-160
timeout 600 python scripts/measure_knn.py 1,0.0 8,0.0 30,0.0 30,0.01 64,0.0 100,0.0 2>&1 | grep '^{' > $O/knn_search.jsonl; cut -c50-130 $O/knn_search.jsonl
timeout 600 python scripts/measure_normals_10m.py 2>&1 | grep normals > $O/normals_10m.txt; cat $O/normals_10m.txt
timeout 600 python scripts/measure_colored.py 2>&1 | grep '^{' > $O/colored.jsonl; cut -c1-200 $O/colored.jsonl
K=knn_normals CMD="python $GRAFT_REPO_ROOT/scripts/measure_normals_10m.py" timeout 500 scripts/dev/pmc_kernel.sh 2>&1 | grep -v amdgpu.ids | grep -v "^pmc" > $O/pmc_normals_last.txt; cat $O/pmc_normals_last.txt
