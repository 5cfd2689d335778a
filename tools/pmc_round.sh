#!/bin/bash
# GPU box (through gpurun): per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
# SQ_* instruction / wait counters for the kernels the bench line does not cover with counters:
# the K = 256 and K = 65536 sweep widths, the compaction / weighted-reduction kernels, the SparseMAP
# solver.  Every pass profiles tools/pmc_kern (one entry point, a few launches, no Python).
#   usage: tools/pmc_round.sh <tag> [quick]
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG/pmc2
LIB=sparse-marginalization-lvm_amd/csrc/libsparsemarg_hip.so
mkdir -p "$OUT"
export TMPDIR=/tmp
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_RD"
run() {  # run <name> <what> <R> <K> <sigma> <passes...>
    local name=$1 what=$2 R=$3 K=$4 sg=$5; shift 5
    local d="$OUT/$name"
    mkdir -p "$d"
    timeout -k 5 60 tools/pmc_kern $LIB $what $R $K $sg 10 > "$d/time.log" 2>&1 || { echo "FAILED plain $name"; tail -3 "$d/time.log"; exit 1; }
    for pass in "$@"; do
        local ctrs
        case $pass in
            fetch) ctrs="FETCH_SIZE";; write) ctrs="WRITE_SIZE";; sqa) ctrs="$SQA";; sqb) ctrs="$SQB";;
        esac
        timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d "$d/$pass" -- tools/pmc_kern $LIB $what $R $K $sg 3 > "$d/$pass.log" 2>&1 \
            || { echo "FAILED $name $pass"; tail -5 "$d/$pass.log"; exit 1; }
    done
    echo "[$(date +%T)] $name: $(grep 'ms per call' $d/time.log)"
}
M=1048576
for sg in 1.0 0.05; do
    for what in fwd bwd; do
        run ${what}_K256_s$sg $what $M 256 $sg fetch write
        run ${what}_K4096_s$sg $what $M 4096 $sg fetch write
        run ${what}_K65536_s$sg $what 32768 65536 $sg fetch write
    done
done
for K in 4096 256; do
    for what in count count_cols compact compact_cols wlr offsets; do
        run ${what}_K${K}_s1.0 $what $M $K 1.0 fetch write
    done
done
run wlr_K256_s0.05 wlr $M 256 0.05 fetch write
# instruction mix / wait cycles of the kernels furthest below their roofline, and of the solver
run sq_fwd_K256_s0.05 fwd $M 256 0.05 sqa sqb
run sq_fwd_K256_s1.0 fwd $M 256 1.0 sqa sqb
run sq_fwd_K65536_s1.0 fwd 32768 65536 1.0 sqa sqb
run sq_fwd_K65536_s0.05 fwd 32768 65536 0.05 sqa sqb
run sq_fwd_K4096_s1.0 fwd $M 4096 1.0 sqa sqb
run sq_bwd_K4096_s1.0 bwd $M 4096 1.0 sqa sqb
run sq_sparsemap_B1024_n128 sparsemap 1024 128 1.0 sqa sqb
run sq_sparsemap_B8192_n128 sparsemap 8192 128 1.0 sqa sqb
find "$OUT" -name '*.csv' -size +4M -delete
find "$OUT" -name '*.db' -delete
echo done
