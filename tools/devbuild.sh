#!/bin/bash
# Experiment build of the library into cmblensing.jl_amd/_dev/lib_<name>.so (select it with CMBL_LIB=), through the product's own
# parallel build (cmblensing.jl_amd/lib.py build: one object per translation unit, own object directory per name).
#   usage: tools/devbuild.sh name [extra hipcc flags]
#   COLS / ROWS restrict the compiled power-of-two tile shapes (default: the 1024^2 shapes only), FULL=1 compiles all of them.
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p cmblensing.jl_amd/_dev
lists=()
if [ -z "$FULL" ]; then lists=("-DCMBL_COL_LIST(X)=${COLS:-X(9,4,512) X(9,2,1024)}" "-DCMBL_ROW_LIST(X)=${ROWS:-X(10)}"); fi
exec python - "$name" "${lists[@]}" "$@" <<'PY'
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("l", "cmblensing.jl_amd/lib.py"); l = importlib.util.module_from_spec(spec); spec.loader.exec_module(l)
name, flags = sys.argv[1], sys.argv[2:]
print(l.build(extra_flags=flags, out=os.path.abspath(f"cmblensing.jl_amd/_dev/lib_{name}.so"), objdir=os.path.abspath(f"build/obj_{name}")))
PY
