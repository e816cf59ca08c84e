#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/measure_colored.py 2000000 > gpurun_out/colored_measured.jsonl 2> gpurun_out/colored_measured.err
cat gpurun_out/colored_measured.jsonl; tail -5 gpurun_out/colored_measured.err
