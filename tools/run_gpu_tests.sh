mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/gputest.log 2>&1; echo "gpu tests rc $?"; tail -15 gpurun_out/r3/gputest.log
