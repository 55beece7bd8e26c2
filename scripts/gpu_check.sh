#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, a short bench.  Stops at the first step that
# was killed by its timeout (never start another GPU step after a hang).
set -u
mkdir -p gpurun_out
run_step() {  # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "=== $name ===" | tee -a gpurun_out/summary.log
    timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/summary.log
    tail -n 25 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "$name TIMED OUT -- stopping" | tee -a gpurun_out/summary.log
        exit 1
    fi
    return $rc
}
: > gpurun_out/summary.log
run_step build 300 python __graft_entry__.py
run_step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
SMOKE=$?
run_step pytest_gpu 1000 python -m pytest tests -m gpu -q --timeout 600
if [ "${SKIP_BENCH:-0}" != "1" ] && [ $SMOKE -eq 0 ]; then
    run_step bench 600 python bench.py ${BENCH_ARGS:-}
    grep -h '^{' gpurun_out/bench.log > gpurun_out/bench.json || true
fi
cat gpurun_out/summary.log
