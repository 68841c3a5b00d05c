# round 3, call 1: GPU test suite on the rebuilt library, then the flagship workload on the cross-wave variants
mkdir -p gpurun_out/r3
[ -n "$SKIPTESTS" ] || python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest1.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r3/gputest1.log
for rep in 1 2; do
for lib in ${LIBS:-xl0 xl2}; do
for shape in 16,4 8,8; do
  echo -n "$lib $shape: "
  LIB=flow-sim_amd/csrc/variants/lib_$lib.so SHAPE=$shape ARGS="--reaches 65536 --steps 20 --warmup 5" bash tools/run_once.sh
done; done; done 2>&1 | tee gpurun_out/r3/xlanes.txt
