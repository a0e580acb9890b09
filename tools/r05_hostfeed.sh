#!/bin/bash
# round 5: the 8-rank host-feed rehearsal on the one-GPU box (config 3 and config 4)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export RELAX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
# config 3: 32 clips per batch and rank at this round's rate (101.7 clips/s per GPU): 315 ms per batch
timeout 900 python bench.py --gpus 8 --workload config3 --dataset-clips 512 --clips-per-step 32 --stub-compute-ms 315 --loader-workers 8 \
    > gpurun_out/r05_rehearsal_config3.json 2> gpurun_out/r05_rehearsal_config3.err; grep -i "error\|Traceback" gpurun_out/r05_rehearsal_config3.err | head -5
# config 4: 64 clips per batch at 200 clips/s per GPU: 320 ms per batch
timeout 900 python bench.py --gpus 8 --workload config4 --dataset-clips 2048 --clips-per-step 64 --stub-compute-ms 320 --loader-workers 8 \
    > gpurun_out/r05_rehearsal_config4.json 2> gpurun_out/r05_rehearsal_config4.err; grep -i "error\|Traceback" gpurun_out/r05_rehearsal_config4.err | head -5
# the same with the NUMA binding off, for comparison
RELAX_NUMA_BIND=0 timeout 900 python bench.py --gpus 8 --workload config4 --dataset-clips 2048 --clips-per-step 64 --stub-compute-ms 320 --loader-workers 8 \
    > gpurun_out/r05_rehearsal_config4_nobind.json 2> gpurun_out/r05_rehearsal_config4_nobind.err
python3 - <<'PY'
import json
for f in ("r05_rehearsal_config3","r05_rehearsal_config4","r05_rehearsal_config4_nobind"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, {k:d[k] for k in d if k not in ("what","workload","note","cpu_model")})
    except Exception as e: print(f, "failed", e)
PY
