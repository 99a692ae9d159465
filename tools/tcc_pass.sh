#!/usr/bin/env bash
# L2 (TCC) hit / miss counters of the sweep kernel, one PMC pass each
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
WL=${1:-ns_1Mx500K}
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_tcc.txt
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/tcc_$tag -- python3 bench.py --no-cpu-baseline --workload $WL --steps 4 --warmup 6 > $OUT/tcc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/tcc_$tag.log; continue; }
  python3 - $OUT/tcc_$tag <<'PY'
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "bca_sweep_csr_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "launches", len(v), "last4 avg %.4g" % (sum(v[-4:]) / 4))
PY
done
