#!/usr/bin/env bash
# Round-2 study: the default concurrency policy (commit protocol) at several staleness budgets, 3 runs each.
set -u
mkdir -p gpurun_out/r02
export XC_STUDY_SWEEPS=6 XC_STUDY_REPEATS=3 XC_STUDY_BUDGETS=4e-3,8e-3,1.6e-2,3.2e-2,1.0
timeout -k 10 300 python tests/studies/policy_study.py 100000 30000 > gpurun_out/r02/policy_c2.txt 2>&1 || exit 1
timeout -k 10 300 python tests/studies/policy_study.py 100000 30000 zipf > gpurun_out/r02/policy_c2_zipf.txt 2>&1 || exit 1
timeout -k 10 300 python tests/studies/policy_study.py 20000 5000 > gpurun_out/r02/policy_20k.txt 2>&1 || exit 1
timeout -k 10 300 python tests/studies/policy_study.py 20000 5000 zipf > gpurun_out/r02/policy_20k_zipf.txt 2>&1 || exit 1
timeout -k 10 400 python tests/studies/policy_study.py 400000 200000 zipf > gpurun_out/r02/policy_400k_zipf.txt 2>&1 || exit 1
XC_STUDY_BUDGETS=4e-3,1.6e-2,1.0 XC_STUDY_REPEATS=2 timeout -k 10 500 python tests/studies/policy_study.py 1000000 500000 zipf > gpurun_out/r02/policy_ns_zipf.txt 2>&1 || exit 1
grep -h "budget=\|oracle" gpurun_out/r02/policy_*.txt | cut -c1-260
