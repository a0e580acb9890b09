#!/bin/bash
# round 6: per-tile cycle stamps (prologue / K loop / epilogue) of every gemm_h3 launch of one ResNet-50 pass (tools/abl_r06/librelax_h2stamps.so)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
RELAX_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_r06/librelax_h2stamps.so python3 tools/resnet_step.py ${1:-1024} 1 both 2>&1 | grep "^h2 " | tail -40
