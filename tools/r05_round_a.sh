#!/bin/bash
# round 5 evidence, part A: the bench lines (run through gpurun)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
( time timeout 1200 python bench.py > $O/bench_config3.json 2> $O/bench_config3.err ) 2> $O/bench_config3.time; tail -3 $O/bench_config3.time
timeout 300 python bench.py --workload config2 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config2.json 2>> $O/bench.err
timeout 300 python bench.py --workload config4 --no-cpu-baseline --no-h2d --steps 6 --warmup 2 > $O/bench_config4.json 2>> $O/bench.err
timeout 600 python bench.py --workload config4 --dataset-clips 1200 --host-clips > $O/bench_config4_dataset.json 2>> $O/bench.err
timeout 400 python bench.py --workload full2160p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full2160p.json 2>> $O/bench.err
timeout 400 python bench.py --workload full1080p --no-cpu-baseline --no-h2d --no-fast-mode --steps 4 --warmup 1 --clips-per-step 8 > $O/bench_full1080p.json 2>> $O/bench.err
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r05p/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline', {})
        print(os.path.basename(f), round(d['value'], 2), d['unit'], 'ms/step', round(d['ms_per_step'], 1), 'frac', round(r.get('frac', 0), 3), 'alg', round(r.get('algorithmic_tflops', 0), 1))
        for k, v in (d.get('other_workloads') or {}).items(): print('   ', k, round(v.get('value', 0), 2), (v.get('host_fed') or {}).get('value'))
        if 'with_pinned_host_to_device_copy' in d: print('    h2d', d['with_pinned_host_to_device_copy']['value'])
        if 'bf16x6_mode' in d: print('    bf16x6', d['bf16x6_mode'].get('value'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
