#!/bin/bash
# round 5: first GPU run of the f16x2 kernel: its parity tests, then one ViT-B pass per arithmetic (timing)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q -s 2>&1 | tail -60 > gpurun_out/r05_h2_tests.txt
tail -40 gpurun_out/r05_h2_tests.txt
for p in bf16x6 f16x2; do timeout 300 python tools/vit_step.py $p 1024 5; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_h2_vit_step.txt
