#!/bin/bash
# Runs on the GPU box (through gpurun): GPU tests, the bench line, a rocprofv3 kernel-trace summary
# of the same bench command, and two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass)
# over a short run of the headline workload.  Everything lands under gpurun_out/<tag>/.
#   usage: tools/profile_round.sh <tag> [skip-tests]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {  # step <seconds> <logfile> <cmd...>
    local t=$1 log=$2; shift 2
    timeout -k 10 "$t" "$@" > "$log" 2>&1
    local rc=$?
    echo "[$(date +%T)] rc=$rc  $*"
    if [ $rc -ne 0 ]; then tail -30 "$log"; exit $rc; fi
}
if [ "${2:-}" != "skip-tests" ]; then
    step 600 "$OUT/pytest_gpu.log" python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread
fi
timeout -k 10 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo bench failed; tail "$OUT/bench.err"; exit 1; }
step 300 "$OUT/rocprof_stats.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu
# the headline workload alone: the sweep launches the same kernel templates on other inputs, which
# would mix into the per-kernel averages above
step 200 "$OUT/rocprof_stats_main.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_main" -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-sweep
step 120 "$OUT/pmc_fetch.log" timeout -s KILL 110 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-sweep
step 120 "$OUT/pmc_write.log" timeout -s KILL 110 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-sweep
# keep only the small summaries (traces of 1M-row launches are small anyway; PMC csv can be big)
find "$OUT" -name '*.csv' -size +8M -delete
ls -R "$OUT" | head -50
tail -c 600 "$OUT/bench.json"
