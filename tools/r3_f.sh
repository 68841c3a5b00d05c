mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "longer_than or gerd_roseires_at" > gpurun_out/r3/gputest5.log 2>&1; echo "rc $?"; tail -30 gpurun_out/r3/gputest5.log
