mkdir -p gpurun_out/r3
for shape in 16,4 8,8; do echo "== stamps $shape"; FS_LIB=$PWD/flow-sim_amd/csrc/variants/lib_xl2stamp.so FS_KERNEL_SHAPE=$shape timeout -k 10 200 python tools/stamps.py 16384 4096; done 2>&1 | tee gpurun_out/r3/stamps_xl2.txt
for lib in x8pin x8nofence x8pf8 x8fence2; do
  echo -n "$lib 8,8: "
  LIB=flow-sim_amd/csrc/variants/lib_$lib.so SHAPE=8,8 ARGS="--reaches 65536 --steps 20 --warmup 5" bash tools/run_once.sh
done 2>&1 | tee gpurun_out/r3/x8_sched.txt
