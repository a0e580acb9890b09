#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/build_ablations.sh h2stamps > /dev/null 2>&1
RELAX_HIP_LIB=tools/abl/librelax_h2stamps.so timeout 300 python tools/vit_step.py f16x2 1024 1 2>&1 | grep "^h2 " | python3 tools/stamp_lines.py | tee gpurun_out/r05_h2_stamps.txt
for i in 1 2; do timeout 300 python tools/vit_step.py f16x2 1024 5; done 2>&1 | grep -v amdgpu
