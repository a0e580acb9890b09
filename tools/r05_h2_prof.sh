#!/bin/bash
# round 5: where a ViT-B pass under f16x2 spends its time: kernel stats (rocprofv3), per-tile stamps, 3 vs 4 LDS stages
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out
tools/build_ablations.sh h2stamps > /dev/null 2>&1
RELAX_HIP_LIB=tools/abl/librelax_h2stamps.so timeout 300 python tools/vit_step.py f16x2 1024 1 2>&1 | grep "^h2 " | python3 tools/stamp_lines.py > gpurun_out/r05_h2_stamps.txt
cat gpurun_out/r05_h2_stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h2vit -- python3 $R/tools/vit_step.py f16x2 1024 3 > $R/gpurun_out/prof_h2vit.log 2>&1
f=$(ls $R/gpurun_out/prof_h2vit/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r05_h2_vit_kernel_stats.csv
head -14 $f | cut -c1-180
