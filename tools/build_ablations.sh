#!/bin/bash
# Builds diagnostic copies of the library into tools/abl/ (never the product build):
#   tools/build_ablations.sh <n> ...        bf16x3 loop ablations (-DRELAX_X3_ABLATE=n, WRONG results)
#   tools/build_ablations.sh f32:<n>        fp32 loop ablations (-DRELAX_F32_ABLATE=n, WRONG results)
#   tools/build_ablations.sh stamps         fp32 / bf16x3 kernel with per-phase timestamps
#   tools/build_ablations.sh x6stamps       bf16x6 kernel with per-phase cycle stamps (prints per launch, syncs)
#   (the bf16x6 loop ablations of round 2 - RELAX_X6_ABLATE - were deleted from gemm_x6.hip in round 3 together with the other
#    experiment branches; their measurements are in DESIGN.md section 3.2 and the code is in the history: commit a51794b)
#   tools/build_ablations.sh att_stamps     fp32 attention kernel with per-phase cycle shares
#   tools/build_ablations.sh flowstamps     fused Farneback iteration with tick stamps per phase of a step (tools/flow_stamps.py prints them)
#   tools/build_ablations.sh flow:<mask>    fused Farneback iteration without parts of its work (-DRELAX_FLOW_ABLATE=mask, WRONG results, timing only)
#   tools/build_ablations.sh a6stamps       bf16x6 attention kernel with ticks per phase of an item (tools/attn_stamps.py prints them)
# use: RELAX_HIP_LIB=tools/abl/librelax_<name>.so python tools/gemm_bench.py ...
set -e
cd "$(dirname "$0")/../relax-vqa_amd/csrc"
make -s
mkdir -p ../../tools/abl
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function"
FLOWCC="$CC -fno-slp-vectorize -ffp-contract=off"   # as the Makefile builds flow.hip
OBJS="api.o fragment.o flow.o resize.o gemm.o gemm_x6.o conv1_x6.o attention_x6.o layers.o resnet50.o vit.o head.o host_logic.o"
link() {  # link <replaced object> <new object> <output name>
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ${OBJS/$1/$2} -o ../../tools/abl/librelax_$3.so
}
for n in "$@"; do
  case "$n" in
    stamps) $CC -DRELAX_GEMM_STAMPS -c gemm.hip -o /tmp/gemm_stamps.o; link gemm.o /tmp/gemm_stamps.o stamps ;;
    x6stamps) $CC -DRELAX_X6_STAMPS -c gemm_x6.hip -o /tmp/gemm_x6_stamps.o; link gemm_x6.o /tmp/gemm_x6_stamps.o x6stamps ;;
    f32:*) $CC -DRELAX_F32_ABLATE=${n#f32:} -c gemm.hip -o /tmp/gemm_f32abl.o; link gemm.o /tmp/gemm_f32abl.o f32abl${n#f32:} ;;
    att_stamps) $CC -DRELAX_ATT_STAMPS=0 -c layers.hip -o /tmp/layers_stamps.o; link layers.o /tmp/layers_stamps.o att_stamps ;;
    flowstamps) $FLOWCC -DRELAX_FLOW_STAMPS -c flow.hip -o /tmp/flow_stamps.o; link flow.o /tmp/flow_stamps.o flowstamps ;;
    flow:*) $FLOWCC -DRELAX_FLOW_ABLATE=${n#flow:} -c flow.hip -o /tmp/flow_abl${n#flow:}.o; link flow.o /tmp/flow_abl${n#flow:}.o flowabl${n#flow:} ;;
    a6stamps) $CC -DRELAX_A6_STAMPS -c attention_x6.hip -o /tmp/attention_x6_stamps.o; link attention_x6.o /tmp/attention_x6_stamps.o a6stamps ;;
    *) $CC -DRELAX_X3_ABLATE=$n -c gemm.hip -o /tmp/gemm_abl$n.o; link gemm.o /tmp/gemm_abl$n.o abl$n ;;
  esac
done
