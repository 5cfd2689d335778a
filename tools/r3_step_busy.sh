#!/bin/bash
# kernels-busy vs wall per model-shaped step: plain run (wall), then the same under rocprofv3
# (sum of kernel durations / steps).  Output: gpurun_out/<tag>/step_busy.txt
TAG=${1:-r03}
OUT=gpurun_out/$TAG/steps
mkdir -p "$OUT"
export TMPDIR=/tmp
: > gpurun_out/$TAG/step_busy.txt
for c in cfg1 cfg1c cfg2 cfg2c cfg4 cfg4_topk cfg5; do
    timeout -k 10 120 python tools/r3_step_busy.py $c > "$OUT/$c.wall" 2>&1 || { tail "$OUT/$c.wall"; exit 1; }
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$c" -- python3 tools/r3_step_busy.py $c > "$OUT/$c.prof" 2>&1 || { tail "$OUT/$c.prof"; exit 1; }
    python3 - "$c" "$OUT" >> gpurun_out/$TAG/step_busy.txt <<'PY'
import csv, glob, sys
c, out = sys.argv[1], sys.argv[2]
wall = [l for l in open(f"{out}/{c}.wall") if l.startswith("STEP")][-1].split()
steps = int(wall[3])
f = glob.glob(f"{out}/{c}/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:4]
print(f"{c}: wall {wall[5]} ms/step, kernels busy {tot / steps / 1e6:.3f} ms/step in {calls / steps:.0f} launches/step; top: " +
      "; ".join(f'{r["Name"][:40]} {float(r["TotalDurationNs"]) / steps / 1e6:.3f}' for r in top))
PY
done
cat gpurun_out/$TAG/step_busy.txt
