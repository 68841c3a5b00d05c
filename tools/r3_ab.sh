mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/gputest13.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r3/gputest13.log | cut -c1-200
for rep in 1 2; do echo -n "irr: "; ARGS="--workload irr --reaches 8192 --steps 16 --warmup 2" bash tools/run_once.sh | cut -c1-60; echo -n "c4: "; ARGS="--workload c4 --reaches 32768 --steps 16 --warmup 2" bash tools/run_once.sh | cut -c1-60; done
