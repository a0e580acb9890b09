#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_h2.py -q -x -k "vit" 2>&1 | tail -3
for kb in 1 0 1 0; do RELAX_OPTS=h2_kb=$kb timeout 300 python tools/vit_step.py f16x2 1024 5 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/kb=$kb: /"; done
tools/build_ablations.sh h2stamps > /dev/null 2>&1
RELAX_HIP_LIB=tools/abl/librelax_h2stamps.so timeout 300 python tools/vit_step.py f16x2 1024 1 2>&1 | grep "^h2 " | python3 tools/stamp_lines.py
