#!/bin/bash
# Round 4, second GPU pass: kernel timelines of the voxel path (new / old) and of the transient
O=gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp && timeout 300 rocprofv3 "$@"); }
python scripts/dev/voxel_one.py 2>&1 | grep voxel | tee $O/voxel_new.txt
MI_ICP_VOXEL_OLD=1 python scripts/dev/voxel_one.py 2>&1 | grep voxel | tee $O/voxel_old.txt
prof --kernel-trace --stats --output-format csv -d $R/$O/st_voxel_new -o s -- python $R/scripts/dev/voxel_one.py > $O/st_voxel_new.log 2>&1; echo "stats voxel new rc=$?"
MI_ICP_VOXEL_OLD=1 prof --kernel-trace --stats --output-format csv -d $R/$O/st_voxel_old -o s -- python $R/scripts/dev/voxel_one.py > $O/st_voxel_old.log 2>&1; echo "stats voxel old rc=$?"
prof --kernel-trace --output-format csv -d $R/$O/tr_transient -o s -- python $R/scripts/dev/transient_one.py > $O/tr_transient.log 2>&1; echo "trace transient rc=$?"
find $O -name "*kernel_stats.csv" | while read f; do echo "== $f"; cut -d, -f1-4 "$f" | sed 's/(.*)"/"/' | head -16; done
# keep only what is small
find $O -name "*.db" -delete 2>/dev/null
ls -la $O/tr_transient/* | head
