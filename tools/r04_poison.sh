#!/bin/bash
# the GPU suite under the two poison modes (workspaces / output tensors start as 0xFF bytes): results must not depend on what a buffer held before
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
RELAX_DEBUG_POISON=1 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r04/poison_ws.txt
RELAX_TEST_POISON_OUT=1 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r04/poison_out.txt
