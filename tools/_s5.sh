export CMBL_PARITY_LOG=$PWD/gpurun_out/r05_parity.log
rm -f $CMBL_PARITY_LOG
python -m pytest tests -m gpu -q > gpurun_out/r05_gputest_1.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r05_gputest_1.log
grep "2048² fp32\|2048² QU fp32\|2048² fp64: \|1024² T+QU" $CMBL_PARITY_LOG | tail -30
