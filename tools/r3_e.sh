mkdir -p gpurun_out/r3
python -m pytest tests/test_random_sweep.py tests/test_gpu_dropin.py -m gpu -q -x -k "heterogeneous or monte_carlo or restart or iterate" > gpurun_out/r3/gputest4.log 2>&1; echo "rc $?"; tail -30 gpurun_out/r3/gputest4.log
