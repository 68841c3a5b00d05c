L=$PWD/flow-sim_amd/csrc/variants/lib_w8.so
for shape in "16,4" "8,8"; do
  echo -n "shape $shape: "
  FS_KERNEL_SHAPE=$shape FS_LIB=$L timeout -k 10 200 python bench.py --reaches 16384 --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel']; print(f\"{d['value']:.4g} r-ts/s kernel_ms {d['roofline']['kernel_ms']:.2f} M={k['cells_per_thread']} W={k['waves_per_reach']} conv={d['config']['all_converged']}\")"
done
