# C5 (trapezoid + power rating curve, 512 nodes) in fp32 on every variant library
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "$(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 300 python bench.py --workload c5 --dtype f32 --nodes 512 --reaches 131072 --steps 32 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} its {d['config']['mean_newton_iterations_per_step']:.3f} conv {d['config']['all_converged']}\")" || echo n/a
done
done
