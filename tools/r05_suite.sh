#!/bin/bash
# round 5: the whole GPU suite (every fp32-grade arithmetic), output kept
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r05_suite.txt
tail -15 gpurun_out/r05_suite.txt
