#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do
RELAX_HIP_LIB=tools/librelax_base.so timeout 300 python tools/vit_step.py f16x2 1024 5 2>&1 | grep -v amdgpu.ids | sed 's/^/base: /'
timeout 300 python tools/vit_step.py f16x2 1024 5 2>&1 | grep -v amdgpu.ids | sed 's/^/new:  /'
done
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q 2>&1 | tail -3
