#!/bin/bash
cd "$(dirname "$0")/.."
export RELAX_DIST_BACKEND=gloo
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -5
