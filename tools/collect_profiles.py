"""Copies the small summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/:
kernel stats of the full bench and of the headline-only run, the bench line, and the sparsemax rows
of the two PMC passes.  usage: python tools/collect_profiles.py <tag>"""
import csv, glob, json, shutil, sys

tag = sys.argv[1]
for kind, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/{tag}/{kind}/vm/*_counter_collection.csv")[0]
    with open(f"profiles/{tag}_{kind}.csv", "w") as out:
        out.write("Kernel_Name,Grid_Size,Workgroup_Size,VGPR_Count,Counter_Name,Counter_Value_KiB,duration_ns\n")
        for r in csv.DictReader(open(f)):
            if "sparsemax" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                out.write('"%s",%s,%s,%s,%s,%f,%d\n' % (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"],
                                                        r["VGPR_Count"], ctr, float(r["Counter_Value"]), dur))
shutil.copy(glob.glob(f"gpurun_out/{tag}/trace/vm/*_kernel_stats.csv")[0], f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(glob.glob(f"gpurun_out/{tag}/trace_main/vm/*_kernel_stats.csv")[0],
            f"profiles/{tag}_bench_main_kernel_stats.csv")
line = [l for l in open(f"gpurun_out/{tag}/bench.json") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench.json", "w").write(line)
d = json.loads(line)
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
print({k: (round(v["ms"], 3), round(v.get("ms_median", 0), 3), round(v.get("hbm_GBps", 0))) for k, v in d["kernels"].items()})
for k, v in d.get("sweep", {}).items():
    print(k, round(v["fwd_ms"], 3), round(v["bwd_ms"], 3), round(v["fwd_frac"], 3), round(v["bwd_frac"], 3),
          round(v["fwd_bwd_frac"], 3), round(v["mean_support"], 1))
for k, v in d.get("train_steps", {}).items():
    print(k, round(v["ms_per_step"], 3))
print(d.get("model_shaped_latency"))
