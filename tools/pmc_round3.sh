#!/bin/bash
# GPU box (through gpurun): counters for what round 3 added or changed -- the fused single-pass
# compaction, the forward on the adversarial mix, the entmax15 forward, the solver after C9-C11.
# Same layout as tools/pmc_round.sh (tools/pmc_collect.py reads it):  tools/pmc_round3.sh <tag>
set -u
TAG=${1:-r03}
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
    timeout -k 5 60 tools/pmc_kern $LIB $what $R $K $sg 20 > "$d/time.log" 2>&1 || { echo "FAILED plain $name"; tail -3 "$d/time.log"; exit 1; }
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
run fused_K4096_s1.0 fused $M 4096 1.0 fetch write
run fused_K256_s1.0 fused $M 256 1.0 fetch write
run fwdadv_K256_s1.0 fwdadv $M 256 1.0 fetch write
run fwdadv_K65536_s1.0 fwdadv 32768 65536 1.0 fetch write
run efwd_K4096_s1.0 efwd $M 4096 1.0 fetch write
run efwd_K65536_s1.0 efwd 32768 65536 1.0 fetch write
run sq_fused_K256_s1.0 fused $M 256 1.0 sqa sqb
run sq_fwdadv_K256_s1.0 fwdadv $M 256 1.0 sqa sqb
run sq_fwdadv_K65536_s1.0 fwdadv 32768 65536 1.0 sqa sqb
run sq_sparsemap_B1024_n128 sparsemap 1024 128 1.0 sqa sqb
run sq_sparsemap_B8192_n128 sparsemap 8192 128 1.0 sqa sqb
find "$OUT" -name '*.csv' -size +4M -delete
find "$OUT" -name '*.db' -delete
echo done
