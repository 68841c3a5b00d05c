# C5 in fp64 on every variant library, two runs each
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "c5f64 $(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 300 python bench.py --workload c5 --dtype f64 --nodes 512 --reaches 131072 --steps 32 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel']; print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.3f} ({k['cells_per_thread']},{k['waves_per_reach']}) bc {k['boundary_class']} diag {k['diag']} vgprs {k['vgprs']} its {d['config']['mean_newton_iterations_per_step']:.4f} conv {d['config']['all_converged']}\")" || echo n/a
done
done
