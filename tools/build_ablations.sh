#!/bin/bash
# Builds timing-only ablation copies of the library (WRONG results) into tools/abl/ for tools/gemm_bench.py:
#   RELAX_HIP_LIB=tools/abl/librelax_abl4.so python tools/gemm_bench.py --precision bf16x3 ...
set -e
cd "$(dirname "$0")/../relax-vqa_amd/csrc"
make -s
mkdir -p ../../tools/abl
for n in "$@"; do
  if [ "$n" = stamps ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -DRELAX_GEMM_STAMPS -c gemm.hip -o /tmp/gemm_stamps.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o fragment.o flow.o resize.o /tmp/gemm_stamps.o layers.o resnet50.o vit.o head.o -o ../../tools/abl/librelax_stamps.so
    continue
  fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -DRELAX_X3_ABLATE=$n -c gemm.hip -o /tmp/gemm_abl$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o fragment.o flow.o resize.o /tmp/gemm_abl$n.o layers.o resnet50.o vit.o head.o -o ../../tools/abl/librelax_abl$n.so
done
