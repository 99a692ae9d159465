#!/bin/bash
# kernel time against a fixed number of concurrent wavefronts (occupancy / latency study)
for wl in ns_1Mx500K c2_100Kx30K; do
  for w in 1280 2560 5120 6144 8192 10240; do
    XCOLUMNS_BCA_WAVES=$w python bench.py --no-cpu-baseline --workload $wl --steps 6 --warmup 2 > gpurun_out/ws.json || exit 1
    python - "$wl" "$w" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ws.json"))
print(sys.argv[1], "W", sys.argv[2], "kernel_ms %.4f" % d["roofline"]["avg_kernel_ms"], "step_ms %.4f" % d["ms_per_step"], flush=True)
PY
  done
done
