#!/usr/bin/env bash
# GPU box, round 3: the default bench line + rocprofv3 summaries of the north-star loop on float32 and on float64 records
mkdir -p gpurun_out/r03
timeout -k 10 600 python bench.py > gpurun_out/r03/bench_ns.json 2> gpurun_out/r03/bench_ns.err || { tail -20 gpurun_out/r03/bench_ns.err; exit 1; }
tools/profile_bench.sh ns > gpurun_out/r03/prof_ns.log 2>&1 || { tail -5 gpurun_out/r03/prof_ns.log; exit 1; }
python3 tools/summarize_profile.py gpurun_out ns 10 gpurun_out/r03/traffic_ns_1Mx500K.json 1000000 600000000 > gpurun_out/r03/r03_ns_rocprofv3_summary.txt
XCOLUMNS_BCA_SHADOW=0 XCOLUMNS_BCA_PACKED=0 tools/profile_bench.sh nsf64 > gpurun_out/r03/prof_nsf64.log 2>&1 || { tail -5 gpurun_out/r03/prof_nsf64.log; exit 1; }
python3 tools/summarize_profile.py gpurun_out nsf64 10 gpurun_out/r03/traffic_ns_1Mx500K_f64.json 1000000 0 > gpurun_out/r03/r03_ns_f64_rocprofv3_summary.txt
cp gpurun_out/prof_ns_trace.log gpurun_out/r03/bench_ns_profiled_run.log 2>/dev/null
cp gpurun_out/prof_nsf64_trace.log gpurun_out/r03/bench_ns_f64_profiled_run.log 2>/dev/null
head -30 gpurun_out/r03/r03_ns_f64_rocprofv3_summary.txt
