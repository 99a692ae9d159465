#!/usr/bin/env bash
# rocprofv3 runs of bench.py on the GPU box: kernel trace + stats, then PMC passes of their own
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; never together with a trace domain).
# The profiled command is the default bench's main timed loop: `--no-extras --repeats 1` drops the
# other legs (f64-record variant, API call, Zipf, C2 parity) whose launches of the same kernel
# template would otherwise be averaged into its line.  Summaries land in gpurun_out/prof_*; copy
# what is judged into profiles/ (tools/summarize_profile.py).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
TAG=${1:-c2}
shift || true
ARGS="$*"
mkdir -p $OUT
cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/prof_${TAG}_trace.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_fetch.log 2>&1 || { echo "fetch run failed"; tail -5 $OUT/prof_${TAG}_fetch.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_write.log 2>&1 || { echo "write run failed"; tail -5 $OUT/prof_${TAG}_write.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/prof_${TAG}_sq -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_sq.log 2>&1 || { echo "sq run failed"; tail -5 $OUT/prof_${TAG}_sq.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/prof_${TAG}_tcc -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_tcc.log 2>&1 || { echo "tcc run failed"; tail -5 $OUT/prof_${TAG}_tcc.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/prof_${TAG}_ea -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 $ARGS > $OUT/prof_${TAG}_ea.log 2>&1 || { echo "ea run failed"; tail -5 $OUT/prof_${TAG}_ea.log; exit 1; }
find $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write -type f | head -30
