#!/bin/bash
# round 6: the whole GPU suite on the final code, plain and under the two poison modes (workspaces / output tensors start as 0xFF bytes), then smoke()
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06s
python -m pytest tests -m gpu -q -x > gpurun_out/r06s/suite.log 2>&1; echo "plain rc=$? $(grep -E ' passed| failed| error' gpurun_out/r06s/suite.log | tail -1)" | tee gpurun_out/r06s/summary.txt
RELAX_DEBUG_POISON=1 python -m pytest tests -m gpu -q -x > gpurun_out/r06s/poison_ws.log 2>&1; echo "RELAX_DEBUG_POISON=1 rc=$? $(grep -E ' passed| failed| error' gpurun_out/r06s/poison_ws.log | tail -1)" | tee -a gpurun_out/r06s/summary.txt
RELAX_TEST_POISON_OUT=1 python -m pytest tests -m gpu -q -x > gpurun_out/r06s/poison_out.log 2>&1; echo "RELAX_TEST_POISON_OUT=1 rc=$? $(grep -E ' passed| failed| error' gpurun_out/r06s/poison_out.log | tail -1)" | tee -a gpurun_out/r06s/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/r06s/summary.txt
