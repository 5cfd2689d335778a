"""Print the keys of a bench line that show the distributed code path ran (first argument: log file)."""
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{") and '"metric"' in ln:
        j = json.loads(ln)
        print({k: j[k] for k in ("value", "n_gpus", "ms_per_step", "ranks_seen", "rccl_version")})
        print("sparsemap", [(e["B"], e["sigma"], round(e["samples_per_s"])) for e in j.get("sparsemap", [])])
        print("train_steps", {k: (round(v["ms_per_step"], 2), v.get("grad_allreduce")) for k, v in j.get("train_steps", {}).items()})
        print("has sweep", "sweep" in j, "cpu_baseline" in j)
    elif ln.strip():
        print("LOG", ln.rstrip()[:300])
