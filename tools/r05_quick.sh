#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_h2.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/vit_step.py f16x2 1024 5 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python tools/resnet_step.py 1024 5 2>&1 | grep -v amdgpu.ids | tail -1
tools/build_ablations.sh h2stamps > /dev/null 2>&1
RELAX_HIP_LIB=tools/abl/librelax_h2stamps.so timeout 300 python tools/vit_step.py f16x2 1024 1 2>&1 | grep "^h2 " | python3 tools/stamp_lines.py
