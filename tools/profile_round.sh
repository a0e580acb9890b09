#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
set -x
R=$GRAFT_REPO_ROOT
python -m pytest -m gpu -q --timeout=900 tests > $R/gpurun_out/t9.log 2>&1
python bench.py > $R/gpurun_out/bench7.json 2> $R/gpurun_out/bench7.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof7 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/prof7.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc7_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/pmc7_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc7_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/pmc7_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc7_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/pmc7_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof7x3 -- python3 $R/tools/profile_step.py bf16x3 > $R/gpurun_out/prof7x3.log 2>&1
tail -2 $R/gpurun_out/t9.log
