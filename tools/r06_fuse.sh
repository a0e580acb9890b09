#!/bin/bash
# round 6: the back-to-back conv2 -> conv3 blocks and the f16x2 conv1: parity tests, then per-launch times of one ResNet-50 pass with and without them
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_h2.py tests/test_gpu_backbones.py -m gpu -q -s -x -k "resnet50 or back_to_back" > gpurun_out/r06_fuse_tests.log 2>&1
tail -3 gpurun_out/r06_fuse_tests.log
for o in rn_fuse=1 rn_fuse=1,rn_c1_h2=0 rn_fuse=0,rn_c1_h2=0; do
  RELAX_OPTS=$o python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1
done
RELAX_OPTS=rn_fuse=1 bash tools/resnet_layers.sh r06_fuse 1024 > /dev/null 2>&1
head -45 gpurun_out/resnet_layers_r06_fuse.txt | grep "true, true\|true, true, false"; tail -1 gpurun_out/resnet_layers_r06_fuse.txt
