#!/bin/bash
# The sharded call with the real GPU engine, two ranks on the one GPU (gloo), against the sequential oracle at a size
# where the checker-engine study (tests/studies/shard_exchange_study.py, XC_STUDY_ROWS_GEN=1) runs too: the exchanges
# per sweep and the overlapped / blocking form one by one.   tools/shard_gpu_study.sh [n,m,sweeps] > gpurun_out/r03/shard_gpu_study.txt
shape=${1:-200000,100000,6}
for ov in 0 1; do
  for s in 1 2 4 8 16 auto; do
    echo "== exchanges=$s overlap=$ov shape=$shape"
    XC_BCA_REHEARSAL_SHAPE=$shape XC_BCA_REHEARSAL_ORACLE=1 XCOLUMNS_BCA_EXCHANGES=$s XCOLUMNS_BCA_EXCHANGE_OVERLAP=$ov \
      timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
      tests/studies/bca_sharded_rehearsal.py 2>&1 | grep -E "^utilities|^oracle|^exchanges" | python -c "
import sys, re
L = sys.stdin.read().splitlines()
g = lambda p: [float(x) for x in re.search(p + r'\s*\[(.*?)\]', '\n'.join(L)).group(1).split(',')]
u, o = g('utilities'), g('oracle')
print('  diff per sweep', ' '.join('%.1e' % abs(a - b) for a, b in zip(u, o)), ' '.join(l for l in L if l.startswith('exchanges')))
"
  done
done
