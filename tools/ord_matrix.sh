#!/usr/bin/env bash
# GPU box: ordered-sweep timing over builds and rows-per-wave (tools/ord_timing.py); output under gpurun_out/r03
out=gpurun_out/r03/$1; shift
mkdir -p gpurun_out/r03; : > $out
for spec in "$@"; do
  lib=${spec%%:*}; R=${spec##*:}
  for w in "--workload c3_amazon670k_150Kx670K" "--workload ns_1Mx500K --sweeps 3" "--workload c2_100Kx30K --zipf" "--workload ns_1Mx500K --zipf --sweeps 3"; do
    echo "== build $lib rows_per_wave $R: $w" >> $out
    if [ "$lib" = default ]; then XCOLUMNS_BCA_ORD_ROWS=$R timeout -k 10 300 python tools/ord_timing.py $w 2>&1 | grep -v amdgpu.ids >> $out
    else XCOLUMNS_BCA_ORD_ROWS=$R timeout -k 10 300 python tools/run_with_lib.py tools/_build/libxc_$lib.so tools/ord_timing.py $w 2>&1 | grep -v amdgpu.ids >> $out; fi
  done
done
cat $out
