mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "irr or polyline or irregular or sweep" > gpurun_out/r3/gputest12.log 2>&1; echo "rc $?"; tail -2 gpurun_out/r3/gputest12.log | cut -c1-200
for rep in 1 2; do echo -n "table: "; ARGS="--workload irr --reaches 8192 --steps 16 --warmup 2" bash tools/run_once.sh; done
