#!/bin/bash
# round 4: the optical-flow stage - parity tests, A/B timing against the two-kernel path, per-kernel times (rocprofv3 --kernel-trace --stats)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_flow.py tests/test_gpu_config5.py tests/test_gpu_reference_png_sets.py -m gpu -x -q 2>&1 | tail -12 > $O/flow_pytest_$TAG.txt
tail -6 $O/flow_pytest_$TAG.txt
python tools/flow_ab.py 2160 3840 32 2>&1 | tee $O/flow_ab_2160_$TAG.txt
python tools/flow_ab.py 1080 1920 32 2>&1 | tee $O/flow_ab_1080_$TAG.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python3 $R/tools/flow_step.py 2160 3840 32 1 4 > /dev/null 2>&1
f=$(find /tmp/fp -name "*kernel_stats.csv" | head -1)
cp $f $O/flow_kernel_stats_$TAG.csv
cut -d, -f1-4 $f | head -16
