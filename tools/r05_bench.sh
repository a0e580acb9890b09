#!/bin/bash
# round 5: the f16x2 parity tests, smoke(), then the driver's default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_x6.py -x -q 2>&1 | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1200 python bench.py > gpurun_out/r05_bench_config3.json 2> gpurun_out/r05_bench_config3.err
tail -3 gpurun_out/r05_bench_config3.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_config3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')})
r=d['roofline']; print({k:r[k] for k in r if k not in ('kernel','traffic_note','frac_note')})
print('bf16x6_mode', d.get('bf16x6_mode',{}).get('value')); print('h2d', d.get('with_pinned_host_to_device_copy',{}).get('value'))
for k,v in (d.get('other_workloads') or {}).items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), (v.get('host_fed') or {}).get('value'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_faithful'))
PY
