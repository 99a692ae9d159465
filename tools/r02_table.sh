#!/usr/bin/env bash
# bench lines of the other workloads for DESIGN section 6's table
set -u
mkdir -p gpurun_out/r02
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" --no-extras --no-cpu-baseline --repeats 3 > gpurun_out/r02/t_$name.json 2> gpurun_out/r02/t_$name.err || { echo "$name failed"; tail -3 gpurun_out/r02/t_$name.err; return; }
python3 - $name <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/r02/t_{sys.argv[1]}.json"))
print("%-16s value %.4g ms/step %.4f frac %.4f kernel ms %s W %s" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"],
      [round(x, 3) for x in j["roofline"]["kernel_ms_by_sweep"]], j["config"].get("concurrent_wavefronts_per_sweep")))
PY
}
run c4 --workload c4_wiki500k_780Kx500K
run c2_zipf --workload c2_100Kx30K --zipf
run c2_zipf_final --workload c2_100Kx30K --zipf --parity final
run c3_final --workload c3_amazon670k_150Kx670K --parity final
run ns_final --parity final
