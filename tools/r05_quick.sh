#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(for o in rn_h2=0 rn_h2=1 rn_h2=0 rn_h2=1; do RELAX_OPTS=$o timeout 300 python tools/resnet_step.py 1024 5 both; done) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_resnet_step.txt
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_x6.py tests/test_gpu_backbones.py -x -q 2>&1 | tail -3
