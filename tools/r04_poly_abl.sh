#!/bin/bash
# round 4: poly_expansion without parts of its work (timing only, WRONG results): 1 no stores, 2 no loads, 4 one of five horizontal taps
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for l in "" polyabl1 polyabl2 polyabl3 polyabl4 polyabl7; do
  if [ -n "$l" ]; then export RELAX_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl/librelax_$l.so; else unset RELAX_HIP_LIB; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/flow_step.py 2160 3840 8 1 2 > /dev/null 2>&1
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  echo "== ${l:-product}"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('poly_expansion','pyramid_fused','flow_visualise')): print('%-28s calls %3s avg %8.1f us' % (r['Name'][:28], r['Calls'], float(r['AverageNs'])/1e3))
"
done
