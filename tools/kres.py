"""Compare `-Rpass-analysis=kernel-resource-usage` remarks of two builds of one .hip file:
    python tools/kres.py old_remarks.txt new_remarks.txt [filter]"""
import re
import subprocess
import sys


def parse(f):
    out, cur = {}, None
    for ln in open(f):
        m = re.search(r'Function Name: (\S+)', ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', ln)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def dem(x):
    return subprocess.run(['c++filt', x], capture_output=True, text=True).stdout.strip()


o, n = parse(sys.argv[1]), parse(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
keys = ("VGPRs", "ScratchSize", "Occupancy", "VGPRs Spill", "SGPRs Spill", "TotalSGPRs")
for k in n:
    name = dem(k)
    if flt and flt not in name:
        continue
    a, b = o.get(k, {}), n[k]
    fa = {x: a.get(x) for x in keys}
    fb = {x: b.get(x) for x in keys}
    if fa != fb:
        print(name[:150])
        print("   old", fa)
        print("   new", fb)
