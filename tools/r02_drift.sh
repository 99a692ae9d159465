#!/usr/bin/env bash
# Round-2 study: per-sweep |utility - sequential oracle| with the round-1 validation (XC_VALIDATE=1)
# and the commit protocol (XC_VALIDATE=2), C2 uniform / Zipf and the north-star size.
set -u
mkdir -p gpurun_out/r02
for mode in 2 1; do
  XC_VALIDATE=$mode timeout -k 10 300 python tests/studies/drift_study.py 100000 30000 > gpurun_out/r02/drift_c2_v$mode.txt 2>&1 || exit 1
  XC_VALIDATE=$mode timeout -k 10 300 python tests/studies/drift_study.py 100000 30000 zipf > gpurun_out/r02/drift_c2_zipf_v$mode.txt 2>&1 || exit 1
done
XC_VALIDATE=2 timeout -k 10 500 python tests/studies/drift_study.py 1000000 500000 > gpurun_out/r02/drift_ns_v2.txt 2>&1 || exit 1
XC_VALIDATE=1 timeout -k 10 500 python tests/studies/drift_study.py 1000000 500000 > gpurun_out/r02/drift_ns_v1.txt 2>&1 || exit 1
grep -h "waves=\|oracle" gpurun_out/r02/drift_*.txt
