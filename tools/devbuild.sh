#!/bin/bash
# Fast experiment build: only the 1024^2 fp32/fp64 tile shapes.  usage: tools/devbuild.sh name [extra hipcc flags]
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p cmblensing.jl_amd/_dev
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "-DCMBL_COL_LIST(X)=${COLS:-X(9,4,512) X(9,2,1024)}" "-DCMBL_ROW_LIST(X)=${ROWS:-X(10)}" "$@" \
  cmblensing.jl_amd/csrc/api.hip -o cmblensing.jl_amd/_dev/lib_$name.so
