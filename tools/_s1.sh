set -x
export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_occ.so
python tools/gpu_occ_ab.py 512 P > gpurun_out/r05_occ_512P.txt 2>&1
python tools/gpu_occ_ab.py 256 P > gpurun_out/r05_occ_256P.txt 2>&1
python tools/gpu_occ_ab.py 256 I > gpurun_out/r05_occ_256I.txt 2>&1
python tools/gpu_occ_ab.py 512 IP > gpurun_out/r05_occ_512IP.txt 2>&1
export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_stamps64.so
N=2048 DT=f64 NRK=10 NB=2048 CMBL_SLICE_STREAMS=1 timeout 600 python tools/gpu_stamps.py > gpurun_out/r05_stamps_2048_f64.txt 2>&1
tail -5 gpurun_out/r05_occ_512P.txt gpurun_out/r05_occ_256P.txt gpurun_out/r05_stamps_2048_f64.txt
