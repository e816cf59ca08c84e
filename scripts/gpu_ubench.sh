#!/bin/bash
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate | tee gpurun_out/valu_rate.log
