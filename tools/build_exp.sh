#!/usr/bin/env bash
# diagnostic builds of the library with one kind of sweep traffic compiled out
# (tools/zipf_sweep1.py XC_LIB=...): never shipped, results are wrong by construction.
set -e
cd "$(dirname "$0")/../xcolumns_amd/csrc"
for exp in SKIP_ACC SKIP_DELTA SKIP_HOT; do
  d=../../tools/_build/exp_$exp; mkdir -p $d
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DXC_EXP_$exp \
      -I../../include -I. -c xc_bca.hip -o $d/xc_bca.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/libxc_exp_$exp.so $d/xc_bca.o \
      _build/xc_lib.o _build/xc_topk.o _build/xc_confusion.o _build/xc_dense.o _build/xc_fw.o _build/xc_coverage.o ) &
done
# row top-k without its selection rounds: the memory-only floor of topk_csr_q4_kernel (tools/topk_timing.py XC_LIB=...)
d=../../tools/_build/exp_TOPK_NOSELECT; mkdir -p $d
( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DXC_EXP_TOPK_NOSELECT \
    -I../../include -I. -c xc_topk.hip -o $d/xc_topk.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/libxc_exp_TOPK_NOSELECT.so $d/xc_topk.o \
    _build/xc_lib.o _build/xc_bca.o _build/xc_confusion.o _build/xc_dense.o _build/xc_fw.o _build/xc_coverage.o ) &
wait
ls -la ../../tools/_build/*.so
