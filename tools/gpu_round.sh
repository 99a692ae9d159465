#!/usr/bin/env bash
# One GPU-box session: coherence probe, smoke, GPU parity tests, bench.
# A step that is killed by its timeout (124/137) ends the session: no further
# GPU work after a hang.  Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
run() { # name timeout cmd...
    local name=$1 t=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/session.log
    timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/session.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "$name timed out: stopping the session" | tee -a gpurun_out/session.log
        tail -5 "gpurun_out/$name.log"
        exit 1
    fi
    return $rc
}
: > gpurun_out/session.log
for step in "$@"; do
    case $step in
        probe) run probe 120 tools/_build/coherence_probe ;;
        smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
        tests) run gpu_tests 900 python -m pytest tests -m gpu -q -x --durations=10 ;;
        tests_all) run gpu_tests 900 python -m pytest tests -m gpu -q --durations=10 ;;
        bench) run bench 600 python bench.py ;;
        bench_zipf) run bench_zipf 600 python bench.py --zipf --no-cpu-baseline ;;
        bench_ns) run bench_ns 900 python bench.py --workload ns_1Mx500K --no-cpu-baseline ;;
        bench_c3) run bench_c3 900 python bench.py --workload c3_amazon670k_150Kx670K --no-cpu-baseline ;;
        e2e) run e2e 600 python tools/e2e_api_timing.py ;;
        drift_ns) run drift_ns 900 python tests/studies/drift_study.py 400000 200000 ;;
        *) echo "unknown step $step" ;;
    esac
done
cat gpurun_out/session.log
for f in probe smoke gpu_tests bench bench_zipf bench_ns bench_c3 e2e drift_ns; do
    [ -f gpurun_out/$f.log ] && { echo "--- $f (tail)"; tail -25 gpurun_out/$f.log; }
done
exit 0
