#!/usr/bin/env bash
# bench lines (main loop only) of the shipped library and of experiment builds, north-star and C2 workloads
set -u
mkdir -p gpurun_out/r02
out=gpurun_out/r02/variants.txt
: > $out
show() { python3 - "$1" "$2" <<'PY' >> gpurun_out/r02/variants.txt
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-22s value %.4g frac %.4f kernel ms by sweep %s W %s" % (sys.argv[1], j["value"], j["roofline"]["frac"],
          [round(x, 3) for x in j["roofline"]["kernel_ms_by_sweep"]], j["config"].get("concurrent_wavefronts_per_sweep", "")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for wl in ns_1Mx500K c2_100Kx30K; do
  timeout -k 10 300 python3 bench.py --workload $wl --no-extras --no-cpu-baseline --repeats 3 > gpurun_out/r02/v_main_$wl.json 2> gpurun_out/r02/v_main_$wl.err || exit 1
  show "main $wl" gpurun_out/r02/v_main_$wl.json
  for v in "$@"; do
    timeout -k 10 300 python3 tools/run_with_lib.py tools/_build/libxc_$v.so bench.py --workload $wl --no-extras --no-cpu-baseline --repeats 3 > gpurun_out/r02/v_${v}_$wl.json 2> gpurun_out/r02/v_${v}_$wl.err || exit 1
    show "$v $wl" gpurun_out/r02/v_${v}_$wl.json
  done
done
cat $out
