#!/usr/bin/env python3
"""Condenses the passes of tools/pmc_round.sh into profiles/:

    python tools/pmc_collect.py <tag>

  profiles/<tag>_pmc_kernels.json   per target: ms per call (plain run, HIP events), HBM bytes per
                                    call = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) (rocprofv3 reports KiB;
                                    gfx950 counts a 128-byte read request as 64: MI355X_MICROARCH.md,
                                    HBM section), the achieved rate on those bytes, SQ_* sums
  profiles/traffic.json             gains / refreshes the keys <what>_K<K>_R<R>[_s<sigma>]
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = {"fwd": "sparsemax_fwd", "bwd": "sparsemax_bwd", "count": "support_count_",
       "count_cols": "support_count_", "compact": "support_compact_",
       "compact_cols": "support_compact_", "offsets": "scan_", "wlr": "wlr_",
       "sparsemap": "sparsemap_fwd_kernel", "fused": "support_compact_fused", "fwdadv": "sparsemax_fwd",
       "efwd": "entmax15_fwd"}
REPS = 3


def counters(d, what):
    """{counter: total over the matching dispatches} and the kernel names seen."""
    tot, names = {}, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if PAT[what] in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names.add(r["Kernel_Name"].split("smlvm::")[-1].split("(")[0])
    return tot, sorted(names)


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", tag, "pmc2")
    out = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for d in sorted(glob.glob(os.path.join(base, "*"))):
        name = os.path.basename(d)
        sq = name.startswith("sq_")
        m = re.match(r"(?:sq_)?([a-z_]+?)_(?:K(\d+)_s([\d.]+)|B(\d+)_n(\d+))$", name)
        if not m:
            continue
        what = m.group(1)
        log = open(os.path.join(d, "time.log")).read()
        ms = float(re.search(r"([\d.]+) ms per call", log).group(1))
        R = int(re.search(r"R=(\d+)", log).group(1)) if "R=" in log else None
        calls = REPS + (1 if what == "fwd" else 0)   # (fwd's set-up launch is the same kernel)
        e = {"what": what, "ms_per_call": ms, "R": R}
        if m.group(2):
            e["K"], e["sigma"] = int(m.group(2)), float(m.group(3))
        nm = re.search(r"N = (\d+)", log)
        if nm:
            e["N"] = int(nm.group(1))
        it = re.search(r"iterations per call: (\d+)", log)
        if it:
            e["active_set_iterations"] = int(it.group(1))
        if not sq:
            f, names = counters(os.path.join(d, "fetch"), what)
            w, _ = counters(os.path.join(d, "write"), what)
            if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
                fk, wk = f["FETCH_SIZE"] / calls, w["WRITE_SIZE"] / calls
                e.update(kernels=names, FETCH_SIZE_KiB=fk, WRITE_SIZE_KiB=wk,
                         hbm_bytes_per_call=1024.0 * (2.0 * fk + wk))
                e["hbm_GBps"] = e["hbm_bytes_per_call"] / (ms * 1e-3) / 1e9
                e["frac_of_8TBps"] = e["hbm_GBps"] / 8000.0
                key = f"{ {'fwd': 'sparsemax_fwd', 'bwd': 'sparsemax_bwd'}.get(what, 'support_' + what if what in ('count', 'compact', 'count_cols', 'compact_cols', 'offsets') else what)}_K{e['K']}_R{R}"
                if e["sigma"] != 1.0:
                    key += f"_s{e['sigma']}"
                traffic[key] = e["hbm_bytes_per_call"]
        else:
            a, names = counters(os.path.join(d, "sqa"), what)
            b, _ = counters(os.path.join(d, "sqb"), what)
            e["kernels"] = names
            e["counters_per_call"] = {k: v / calls for k, v in {**a, **b}.items()}
            c = e["counters_per_call"]
            if c.get("SQ_WAVE_CYCLES"):
                e["wait_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                e["issue_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            if it and c.get("SQ_INSTS_VALU"):
                n_it = e["active_set_iterations"]
                e["per_iteration"] = {k: c[k] / n_it for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if k in c}
        out[name] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_kernels.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    for k, e in out.items():
        if "hbm_GBps" in e:
            print(f"{k:28s} {e['ms_per_call']:8.4f} ms  traffic {e['hbm_bytes_per_call'] / 1e9:8.3f} GB  {e['hbm_GBps']:7.0f} GB/s  {e['frac_of_8TBps']:.3f}")
        elif "wait_frac_of_wave_cycles" in e:
            print(f"{k:28s} {e['ms_per_call']:8.4f} ms  issue {e['issue_frac_of_wave_cycles']:.2f} wait {e['wait_frac_of_wave_cycles']:.2f}  {e.get('per_iteration', '')}")


if __name__ == "__main__":
    main()
