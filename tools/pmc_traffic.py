#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes of tools/profile_round.sh into profiles/traffic.json.

    python tools/pmc_traffic.py gpurun_out/<tag> <rows> <K> [round-label]

HBM bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): rocprofv3 reports both counters in
KiB, and on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes
(MI355X_MICROARCH.md, section HBM) -- hence the factor 2 on the read side; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  bench.py reads the file to fill `roofline.traffic`.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(dirname, counter):
    out = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "smlvm::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("smlvm::")[1].split("(")[0]
            out.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    tag_dir, rows, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    label = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(tag_dir.rstrip("/"))
    fetch = per_kernel(os.path.join(tag_dir, "pmc_fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(tag_dir, "pmc_write"), "WRITE_SIZE")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    detail = {}
    for name in sorted(set(fetch) & set(write)):
        kind = "sparsemax_fwd" if "sparsemax_fwd" in name else ("sparsemax_bwd" if "sparsemax_bwd" in name else None)
        if kind is None:
            continue
        total = 1024.0 * (2.0 * fetch[name] + write[name])
        data[f"{kind}_K{K}_R{rows}"] = total
        detail[name] = {"FETCH_SIZE_KiB": fetch[name], "WRITE_SIZE_KiB": write[name],
                        "hbm_bytes_per_launch": total,
                        "algorithmic_bytes": (8.0 if kind.endswith("fwd") else 12.0) * K * rows}
    data.setdefault("_detail", {})[label] = detail
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(detail, indent=1))


if __name__ == "__main__":
    main()
