#!/bin/bash
# round 5: the GPU suite under both poison modes, and the self-arming real-checkpoint tests armed with checkpoint FILES (the synthetic state
# dicts saved the way the hub files are: plain for ResNet-50, DINO's {"teacher": {"backbone.…"}} wrapping for the ViT)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
import relax_vqa_amd
from relax_vqa_amd import synth
torch.save({k: torch.from_numpy(v) for k, v in synth.resnet50_state_dict().items()}, "/tmp/rn50.pth")
torch.save({"teacher": {"backbone." + k: torch.from_numpy(v) for k, v in synth.vit_state_dict("vit_base").items()}, "epoch": 3}, "/tmp/vit.pth")
PY
RELAX_RESNET50_WEIGHTS=/tmp/rn50.pth RELAX_VIT_WEIGHTS=/tmp/vit.pth timeout 900 python -m pytest tests/test_gpu_real_weights.py -q -s 2>&1 | grep "checkpoint\|passed\|failed" | tee gpurun_out/r05_real_weights_armed.txt
RELAX_DEBUG_POISON=1 timeout 1500 python -m pytest tests -m gpu -x -q > /tmp/poison_ws.log 2>&1; echo "RELAX_DEBUG_POISON=1 rc=$? $(grep -E ' passed| failed| error' /tmp/poison_ws.log | tail -1)" | tee gpurun_out/r05_poison_ws.txt
RELAX_TEST_POISON_OUT=1 timeout 1500 python -m pytest tests -m gpu -x -q > /tmp/poison_out.log 2>&1; echo "RELAX_TEST_POISON_OUT=1 rc=$? $(grep -E ' passed| failed| error' /tmp/poison_out.log | tail -1)" | tee gpurun_out/r05_poison_out.txt
