#!/usr/bin/env bash
# GPU box: the whole -m gpu suite (log under gpurun_out/r03), then whatever command follows
mkdir -p gpurun_out/r03
tag=$1; shift
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 -s > gpurun_out/r03/gpu_tests_$tag.log 2>&1
rc=$?
tail -40 gpurun_out/r03/gpu_tests_$tag.log
[ $rc -ne 0 ] && exit $rc
"$@"
