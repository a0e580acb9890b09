#!/bin/bash
# Builds diagnostic copies of the library into tools/abl/ (never the product build).  The product sources carry only empty stamp hooks
# and no ablation switches: a diagnostic kernel exists in the variant translation units tools/abl/*_stamps.hip alone (each defines the
# hooks and #includes the product file), so no -D on the product build can make librelax_hip.so record stamps or return wrong numbers.
#   tools/build_ablations.sh x6stamps       bf16x6 kernel with per-phase cycle stamps (prints per launch, syncs)
#   tools/build_ablations.sh h2stamps       f16x2 kernel with per-phase cycle stamps (prints per launch, syncs)
#   tools/build_ablations.sh h2l2hit        the same with every K step of a plain gemm_h3 re-reading the first four (WRONG results: the K loop with
#                                           every DMA piece served by L2)
#   tools/build_ablations.sh b2babl:<mask>  the back-to-back kernel of gemm_x6.hip without parts of its tail (1 no fp32 stores, 2 no residual loads, 4 no conv3 phase;
#                                           WRONG results, timing only); b2bpf:<0|1|2> / x6stg2: same-bits variants (B-fragment prefetch mode; two LDS stages)
#   tools/build_ablations.sh h3b2b          the f16x2 kernel with the conv3 launches of ResNet-50's layer3 / layer4 never fetching their A operand and the 3x3
#                                           launches never storing their planes (WRONG results: the most a conv2 -> conv3 fusion there could return)
#   tools/build_ablations.sh flowstamps     fused Farneback iteration with tick stamps per phase of a step (tools/flow_stamps.py prints them)
#   tools/build_ablations.sh a6stamps       bf16x6 attention kernel with ticks per phase of an item (tools/attn_stamps.py prints them;
#                                           overwrites the first floats of the fp32 output: timing only)
# The loop ablations of rounds 1 - 4 (RELAX_X3_ABLATE, RELAX_F32_ABLATE, RELAX_X6_ABLATE, RELAX_FLOW_ABLATE: WRONG results, timing only)
# and the stamps of the fp32-option kernels were deleted from the sources in rounds 3 and 5; their measurements are in LAB_NOTES.md and
# profiles/, the code in the history (commits a51794b, d58970c).
# use: RELAX_HIP_LIB=tools/abl/librelax_<name>.so python tools/gemm_bench.py ...   (tools/abl/*.so is not shipped to the GPU box by
# gpurun - .gpurunignore - build it there: tools/build_ablations.sh <name> as the first step of the command)
set -e
cd "$(dirname "$0")/../relax-vqa_amd/csrc"
make -s
mkdir -p ../../tools/abl
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function"
FLOWCC="$CC -fno-slp-vectorize -ffp-contract=off"   # as the Makefile builds flow.hip
OBJS="api.o fragment.o flow.o resize.o gemm.o gemm_x6.o gemm_h2.o conv1_x6.o attention_x6.o attention_h2.o layers.o resnet50.o vit.o head.o host_logic.o"
link() {  # link <replaced object> <new object> <output name>
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ${OBJS/$1/$2} -o ../../tools/abl/librelax_$3.so
}
for n in "$@"; do
  case "$n" in
    x6stamps) $CC -I. -c ../../tools/abl/gemm_x6_stamps.hip -o /tmp/gemm_x6_stamps.o; link gemm_x6.o /tmp/gemm_x6_stamps.o x6stamps ;;
    h2stamps) $CC -I. -c ../../tools/abl/gemm_h2_stamps.hip -o /tmp/gemm_h2_stamps.o; link gemm_h2.o /tmp/gemm_h2_stamps.o h2stamps ;;
    b2babl:*) $CC -I. -DB2B_ABL_MASK=${n#b2babl:} -c ../../tools/abl/gemm_x6_b2b_abl.hip -o /tmp/gemm_x6_b2babl.o; link gemm_x6.o /tmp/gemm_x6_b2babl.o b2babl${n#b2babl:} ;;
    b2bpf:*) $CC -I. -DB2B_PF_MODE=${n#b2bpf:} -c ../../tools/abl/gemm_x6_b2b_abl.hip -o /tmp/gemm_x6_b2bpf.o; link gemm_x6.o /tmp/gemm_x6_b2bpf.o b2bpf${n#b2bpf:} ;;
    x6stg2) $CC -I. -DX6_STG2 -c ../../tools/abl/gemm_x6_b2b_abl.hip -o /tmp/gemm_x6_stg2.o; link gemm_x6.o /tmp/gemm_x6_stg2.o x6stg2 ;;
    h3b2b) $CC -I. -DH3_B2B_BOUND -c ../../tools/abl/gemm_h2_stamps.hip -o /tmp/gemm_h2_b2b.o; link gemm_h2.o /tmp/gemm_h2_b2b.o h3b2b ;;
    h3ahead) $CC -I. -DH3_EP_FETCH_AHEAD=1 -c gemm_h2.hip -o /tmp/gemm_h2_ahead.o; link gemm_h2.o /tmp/gemm_h2_ahead.o h3ahead ;;
    h3ahead_stamps) $CC -I. -DH3_EP_FETCH_AHEAD=1 -c ../../tools/abl/gemm_h2_stamps.hip -o /tmp/gemm_h2_ahead_st.o; link gemm_h2.o /tmp/gemm_h2_ahead_st.o h3ahead_stamps ;;
    h3stagger:*) $CC -I. -DH3_STAGGER=${n#h3stagger:} -c gemm_h2.hip -o /tmp/gemm_h2_stg.o; link gemm_h2.o /tmp/gemm_h2_stg.o h3stagger${n#h3stagger:} ;;
    h2l2hit) $CC -I. -DH3_L2HIT -c ../../tools/abl/gemm_h2_stamps.hip -o /tmp/gemm_h2_l2hit.o; link gemm_h2.o /tmp/gemm_h2_l2hit.o h2l2hit ;;
    flowstamps) $FLOWCC -I. -c ../../tools/abl/flow_stamps.hip -o /tmp/flow_stamps.o; link flow.o /tmp/flow_stamps.o flowstamps ;;
    a6stamps) $CC -I. -c ../../tools/abl/attention_x6_stamps.hip -o /tmp/attention_x6_stamps.o; link attention_x6.o /tmp/attention_x6_stamps.o a6stamps ;;
    *) echo "unknown diagnostic build: $n" >&2; exit 2 ;;
  esac
done
