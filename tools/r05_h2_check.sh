#!/bin/bash
# round 5: the f16x2 parity tests, one ViT-B pass per arithmetic / loop form (timing), per-tile stamps and kernel stats of the pass
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q -s 2>&1 | tail -80 > gpurun_out/r05_h2_tests.txt
tail -5 gpurun_out/r05_h2_tests.txt
(for p in bf16x6 f16x2; do timeout 300 python tools/vit_step.py $p 1024 5; done
 RELAX_H2_FORM=0 timeout 300 python tools/vit_step.py f16x2 1024 5
 RELAX_H2_FORM=2 timeout 300 python tools/vit_step.py f16x2 1024 5) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_h2_vit_step.txt
tools/r05_h2_prof.sh
