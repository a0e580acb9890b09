#!/bin/bash
# round 6: the fetch-ahead form of gemm_h3's convolution epilogue (-DH3_EP_FETCH_AHEAD=1, tools/build_ablations.sh h3ahead h3ahead_stamps) against the
# product library on the same box: one ResNet-50 pass each, twice, then the stamps of the variant
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1 | sed 's/^/product  /'
  RELAX_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_r06/librelax_h3ahead.so python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1 | sed 's/^/ahead    /'
done
RELAX_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_r06/librelax_h3ahead_stamps.so python3 tools/resnet_step.py 1024 1 both 2>&1 | grep "^h2 " | tail -40 | head -12
