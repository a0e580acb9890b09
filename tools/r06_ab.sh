#!/bin/bash
# round 6: same-box A/B of the product library against a variant library (tools/abl_r06/librelax_<tag>.so): ViT-B pass and ResNet-50 pass over 1024
# fragments, alternating, three times.   tools/r06_ab.sh <tag>
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/abl_r06/librelax_$1.so
for r in 1 2 3; do
  echo "product: $(python3 tools/vit_step.py f16x2 1024 5 2>&1 | tail -1 | cut -c1-70) | $(python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1 | cut -c1-50)"
  echo "$1: $(RELAX_HIP_LIB=$V python3 tools/vit_step.py f16x2 1024 5 2>&1 | tail -1 | cut -c1-70) | $(RELAX_HIP_LIB=$V python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1 | cut -c1-50)"
done
