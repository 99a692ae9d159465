#!/usr/bin/env bash
# experiment build of the library: xc_bca.hip compiled with extra defines, the other objects as built
#   tools/build_variant.sh NAME -DXC_SWEEP_WAVES_PER_EU=6   ->  tools/_build/libxc_NAME.so
# run a script against it with tools/run_with_lib.py
set -e
name=$1; shift
cd "$(dirname "$0")/../xcolumns_amd/csrc"
d=../../tools/_build/var_$name; mkdir -p $d
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -I../../include -I. -c xc_bca.hip -o $d/xc_bca.o
objs=$(ls _build/*.o | grep -v xc_bca.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/libxc_$name.so $d/xc_bca.o $objs
echo built tools/_build/libxc_$name.so
