#!/usr/bin/env bash
# Round-2 GPU session: rocprofv3 profiles of the default bench's main loop (north-star workload) and of
# configs[1] (C2), their summaries, and the full default bench line.
set -u
mkdir -p gpurun_out/r02
tools/profile_bench.sh ns || exit 1
python3 tools/summarize_profile.py gpurun_out ns 10 gpurun_out/r02/traffic_ns_1Mx500K.json 1000000 600000000 > gpurun_out/r02/r02_ns_rocprofv3_summary.txt 2>&1
tools/profile_bench.sh c2 --workload c2_100Kx30K || exit 1
python3 tools/summarize_profile.py gpurun_out c2 10 gpurun_out/r02/traffic_c2_100Kx30K.json 100000 60000000 > gpurun_out/r02/r02_c2_rocprofv3_summary.txt 2>&1
for f in gpurun_out/prof_ns_trace gpurun_out/prof_c2_trace; do
  find $f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/$(basename $f)_kernel_stats.csv
done
cp gpurun_out/prof_ns_trace.log gpurun_out/r02/bench_ns_profiled.json 2>/dev/null
cp gpurun_out/prof_c2_trace.log gpurun_out/r02/bench_c2_profiled.json 2>/dev/null
timeout -k 10 900 python3 bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err || { tail -5 gpurun_out/r02/bench_default.err; exit 1; }
timeout -k 10 600 python3 bench.py --workload c2_100Kx30K --no-zipf > gpurun_out/r02/bench_c2.json 2> gpurun_out/r02/bench_c2.err || { tail -5 gpurun_out/r02/bench_c2.err; exit 1; }
head -12 gpurun_out/r02/r02_ns_rocprofv3_summary.txt
grep -n "timed region\|bca_sweep_csr_kernel launches" gpurun_out/r02/r02_ns_rocprofv3_summary.txt gpurun_out/r02/r02_c2_rocprofv3_summary.txt | cut -c1-300
python3 - <<'PY'
import json
for f in ("bench_default", "bench_c2"):
    j = json.load(open(f"gpurun_out/r02/{f}.json"))
    print(f, "value %.4g ms/step %.4f frac %.4f by_sweep %s f64 %.4f api %s zipf %s" % (
        j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_by_sweep"], j["roofline"].get("frac_f64_records") or -1,
        {k: round(v["rows_per_s"] / 1e6, 1) for k, v in j.get("api_call", {}).items() if isinstance(v, dict)},
        (j.get("zipf") or {}).get("roofline_frac")))
PY
