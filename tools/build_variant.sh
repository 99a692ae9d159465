#!/usr/bin/env bash
# experiment build of the library: one source (SRC, default xc_bca) compiled with extra defines, the other objects as built
#   tools/build_variant.sh NAME -DXC_SWEEP_WAVES_PER_EU=6        ->  tools/_build/libxc_NAME.so
#   SRC=xc_bca_ord tools/build_variant.sh ord512 -DXC_ORD_BLOCK=512
# run a script against it with tools/run_with_lib.py
set -e
name=$1; shift
cd "$(dirname "$0")/../xcolumns_amd/csrc"
d=../../tools/_build/var_$name; mkdir -p $d
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -I../../include -I. -c ${SRC:-xc_bca}.hip -o $d/${SRC:-xc_bca}.o
objs=$(ls _build/*.o | grep -v "/${SRC:-xc_bca}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/libxc_$name.so $d/${SRC:-xc_bca}.o $objs
echo built tools/_build/libxc_$name.so
