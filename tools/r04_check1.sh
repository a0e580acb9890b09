#!/bin/bash
# round 4, first GPU pass: the new parity tests (reference PNG sets at 1080p / 2160p, host-fed dataset pass, backend-aware records)
# and the host-fed dataset bench.  Run through gpurun.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_reference_png_sets.py tests/test_gpu_dataset.py tests/test_gpu_errors_and_dist.py -m gpu -x -q 2>&1 | tail -25 > $O/check1_pytest.txt
tail -5 $O/check1_pytest.txt
python bench.py --workload config4 --dataset-clips 512 --host-clips > $O/ds_hostfed_512.json 2> $O/ds_hostfed_512.err
tail -c 1500 $O/ds_hostfed_512.json
