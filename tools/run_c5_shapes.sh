# C5 (trapezoid + power rating curve, 512 nodes) on every variant library x kernel shape (cells per lane, waves per reach)
for dt in ${DTYPES:-f32 f64}; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
for shape in ${SHAPES:-"8,1" "4,2" "2,4"}; do
  echo -n "$dt $(basename $v) $shape "
  FS_KERNEL_SHAPE=$shape FS_LIB=$PWD/$v timeout -k 10 300 python bench.py --workload c5 --dtype $dt --nodes 512 --reaches 131072 --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} its {d['config']['mean_newton_iterations_per_step']:.3f} conv {d['config']['all_converged']}\")" || echo n/a
done
done
done
