# flagship workload with alternative kernel shapes (cells per lane, waves per reach)
for shape in "16,4" "8,8" "8,4"; do
  echo -n "shape $shape "
  FS_KERNEL_SHAPE=$shape FS_LIB=$PWD/flow-sim_amd/csrc/variants/lib_cur.so timeout -k 10 300 python bench.py --reaches 16384 --steps 32 --warmup 2 --no-cpu-baseline --nodes ${NODES:-4096} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} conv {d['config']['all_converged']}\")"
done
