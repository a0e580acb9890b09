#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_h2.py tests/test_gpu_x6.py -q -x 2>&1 | tail -4
for e in 1 0 1 0; do RELAX_OPTS=rn_h2_early=$e timeout 300 python tools/resnet_step.py 1024 5 2>&1 | grep -v amdgpu.ids | tail -1; done
tools/resnet_layers.sh early 1024 2>&1 | head -42 | tail -36
