mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest2.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r3/gputest2.log
timeout -k 10 500 python tools/scan_supercritical.py 8000 > gpurun_out/r3/supercritical_scan.txt 2>&1; tail -5 gpurun_out/r3/supercritical_scan.txt
for rep in 1 2; do for lib in mon0 mon1; do
  echo -n "$lib c3: "; LIB=flow-sim_amd/csrc/variants/lib_$lib.so ARGS="--reaches 65536 --steps 20 --warmup 5" bash tools/run_once.sh
  echo -n "$lib c4: "; LIB=flow-sim_amd/csrc/variants/lib_$lib.so ARGS="--workload c4 --reaches 32768 --steps 16 --warmup 2" bash tools/run_once.sh
done; done 2>&1 | tee gpurun_out/r3/monitor_cost.txt
