mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/gputest13.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r3/gputest13.log | cut -c1-200
for rep in 1 2 3; do for lib in final finalpf1; do
  echo -n "$lib: "; LIB=flow-sim_amd/csrc/variants/lib_$lib.so ARGS="--reaches 65536 --steps 20 --warmup 5" bash tools/run_once.sh | cut -c1-60
done; done
