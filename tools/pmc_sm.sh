set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/smpmc
timeout -k 10 60 tools/sm_time 1024 128 0 1.0 || exit 1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/smpmc/p1 -- tools/sm_time 1024 128 0 1.0 > gpurun_out/smpmc/p1.log 2>&1 || { tail gpurun_out/smpmc/p1.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM --output-format csv -d gpurun_out/smpmc/p2 -- tools/sm_time 1024 128 0 1.0 > gpurun_out/smpmc/p2.log 2>&1 || { tail gpurun_out/smpmc/p2.log; exit 1; }
python3 - <<'P'
import csv,glob
for d in ("p1","p2"):
    for f in glob.glob(f"gpurun_out/smpmc/{d}/**/*counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            if "sparsemap_fwd" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,v)
P
