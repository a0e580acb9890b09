#!/bin/bash
# quick A/B of engine options on one ResNet-50 pass: tools/r06_quick.sh "opt=a,opt=b" "opt=c" ...
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd $GRAFT_REPO_ROOT
for o in "$@"; do RELAX_OPTS=$o python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1; done
for o in "$@"; do RELAX_OPTS=$o python3 tools/resnet_step.py 1024 5 both 2>&1 | tail -1; done
