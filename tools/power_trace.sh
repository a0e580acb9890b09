#!/bin/bash
# Samples rocm-smi (power, clocks, temperature) twice a second while a command runs; prints the samples' summary.
#   tools/power_trace.sh OUT.txt -- python bench.py --steps 60 --no-cpu-baseline --no-fast-mode --no-h2d
out=$1; shift; shift
( while true; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > "$out.raw" &
poll=$!
"$@"
rc=$?
kill $poll 2>/dev/null
python3 - "$out" <<'PY'
import json, sys
rows = []
for line in open(sys.argv[1] + ".raw"):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    c = d.get("card0", {})
    rows.append(c)
keys = sorted({k for r in rows for k in r})
with open(sys.argv[1], "w") as f:
    f.write(f"{len(rows)} samples of rocm-smi --showpower --showclocks --showtemp, 0.5 s apart\n")
    for k in keys:
        vals = []
        for r in rows:
            v = str(r.get(k, "")).strip("()").replace("Mhz", "").replace("MHz", "")
            try:
                vals.append(float(v))
            except ValueError:
                pass
        if vals:
            vals.sort()
            f.write(f"{k:60s} min {vals[0]:9.1f}  median {vals[len(vals) // 2]:9.1f}  max {vals[-1]:9.1f}\n")
print(open(sys.argv[1]).read())
PY
exit $rc
