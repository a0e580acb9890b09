#!/bin/bash
# round 6: attention_h2 per-launch time, product library against tools/abl_r06/librelax_prev.so on the same box (rocprofv3 --stats of one ViT-B pass, twice each)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for t in product prev product2 prev2; do
  if [[ $t == prev* ]]; then export RELAX_HIP_LIB=$R/tools/abl_r06/librelax_prev.so; else unset RELAX_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/attn_$t -- python3 $R/tools/vit_step.py f16x2 1024 3 > /dev/null 2>&1
  f=$(ls -t $R/gpurun_out/attn_$t/*/*kernel_stats.csv | head -1)
  echo "$t: $(python3 -c "import csv,sys; [print(r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us') for r in csv.DictReader(open('$f')) if 'attention_h2' in r['Name']]")"
done
