#!/bin/bash
# Builds variants/libtd_<name>.so: csrc/<file>.hip recompiled with extra flags, the other objects
# taken from the regular build (run `python -c "import __graft_entry__ as g; g.build()"` first).
#   tools/build_variant.sh <name> <file.hip> <flags...>      select it with TD_HOTPATH_LIB
# The recompiled source is a DEVELOPMENT build (-DTD_DEV_SWITCHES): it reads the TD_* environment
# switches of its A/B runs; the shipped library (telluride_decoding_amd/build.py) reads none.
# `tools/build_variant.sh dev <file.hip>` = the regular kernels of that source with the switches on.
set -eu
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/telluride_decoding_amd/csrc/_obj
mkdir -p $root/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -Wno-unused-result \
  -I$root/include -I$root/telluride_decoding_amd/csrc -DTD_DEV_SWITCHES "$@" -c $root/telluride_decoding_amd/csrc/$src \
  -o $root/variants/$name.$src.o 2> $root/variants/$name.log
others=$(ls $obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/variants/libtd_$name.so $root/variants/$name.$src.o $others
echo built variants/libtd_$name.so
