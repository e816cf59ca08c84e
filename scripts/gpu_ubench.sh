#!/bin/bash
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate | tee gpurun_out/valu_rate.log
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/salu_rate.hip -o /tmp/salu_rate && timeout 120 /tmp/salu_rate | tee gpurun_out/salu_rate.log
