#!/bin/bash
# Same-box A/B of library builds (boxes differ by +-1.5 %, so two builds are only comparable inside ONE gpurun call):
#   tools/ab_bench.sh tools/abl/librelax_head.so relax-vqa_amd/csrc/librelax_hip.so [-- bench args]
# runs the headline bench twice per library, alternating, and prints clips/s, ms per step and the roofline fraction.
libs=(); args=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args=("$@"); break; fi
  libs+=("$1"); shift
done
for i in 1 2; do
  for L in "${libs[@]}"; do
    RELAX_HIP_LIB=$L python bench.py --no-cpu-baseline --no-h2d --no-fast-mode --steps 12 "${args[@]}" 2>&1 | tail -1 |
      python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['frac'],4))"
  done
done
