export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_stag.so
ROUNDS=3 NT=20 python tools/gpu_opt_ab.py slice_stagger_ns 0,4000,8000,12000,16000,24000 1024 P f32 7 2>&1 | grep -E "MIN|round 0"
ROUNDS=2 NT=20 python tools/gpu_opt_ab.py slice_stagger_ns 0,6000,12000,20000 1024 IP f32 7 2>&1 | grep -E "MIN"
