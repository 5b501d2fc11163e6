export CMBL_PARITY_LOG=$PWD/gpurun_out/r05_parity.log
rm -f $CMBL_PARITY_LOG
python -m pytest tests -m gpu -q > gpurun_out/r05_gputest_1.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r05_gputest_1.log
