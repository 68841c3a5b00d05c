# a workload size (NODES, REACHES) with alternative kernel shapes (cells per lane, waves per reach) of the shipped library
for shape in ${SHAPES:-"16,4" "8,8" "8,4"}; do
  echo -n "nodes ${NODES:-4096} shape $shape "
  FS_KERNEL_SHAPE=$shape timeout -k 10 300 python bench.py --reaches ${REACHES:-16384} --steps 32 --warmup 2 --no-cpu-baseline --nodes ${NODES:-4096} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} conv {d['config']['all_converged']}\")" || echo "n/a"
done
