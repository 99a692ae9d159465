#!/usr/bin/env bash
# diagnostic build of the library with in-kernel s_memtime stamps (tools/stamp_study.py)
set -e
cd "$(dirname "$0")/../xcolumns_amd/csrc"
mkdir -p ../../tools/_build/stamps_obj
for f in xc_lib xc_topk xc_confusion xc_bca xc_dense xc_fw xc_coverage xc_order xc_bca_det xc_scatter; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DXC_STAMPS \
      -I../../include -I. -c $f.hip -o ../../tools/_build/stamps_obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/libxcolumns_amd_stamps.so ../../tools/_build/stamps_obj/*.o
echo built tools/_build/libxcolumns_amd_stamps.so
