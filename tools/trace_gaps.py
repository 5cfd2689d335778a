"""Idle gaps on the GPU from a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows = rows[len(rows) // 2:]   # second half: steady state
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernels {len(rows)}  busy {busy/1e6:.2f} ms  span {span/1e6:.2f} ms  idle {100*(1-busy/span):.1f} %")
gaps = {}
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g >= thr:
        key = (a["Kernel_Name"][:60], b["Kernel_Name"][:60])
        n, t = gaps.get(key, (0, 0.0))
        gaps[key] = (n + 1, t + g)
for (ka, kb), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t/1e3:8.2f} ms in {n:3d} gaps   after {ka}  ->  before {kb}")
