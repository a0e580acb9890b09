#!/bin/bash
# the GPU suite under the two poison modes (workspaces / output tensors start as 0xFF bytes): results must not depend on what a buffer held before
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
RELAX_DEBUG_POISON=1 python -m pytest tests -m gpu -q -x > /tmp/poison_ws.log 2>&1; echo "RELAX_DEBUG_POISON=1 rc=$? $(grep -E ' passed| failed| error' /tmp/poison_ws.log | tail -1)" | tee gpurun_out/r04/poison_ws.txt
RELAX_TEST_POISON_OUT=1 python -m pytest tests -m gpu -q -x > /tmp/poison_out.log 2>&1; echo "RELAX_TEST_POISON_OUT=1 rc=$? $(grep -E ' passed| failed| error' /tmp/poison_out.log | tail -1)" | tee gpurun_out/r04/poison_out.txt
