# the one-wave workloads (C4, C5 fp32 / fp64, polyline ensemble) on every variant library, two runs each
run() { # name, bench args
  for rep in 1 2; do
  for v in flow-sim_amd/csrc/variants/lib_*.so; do
    echo -n "$1 $(basename $v) "
    FS_LIB=$PWD/$v timeout -k 10 300 python bench.py $2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel']; print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.3f} ({k['cells_per_thread']},{k['waves_per_reach']}) bc {k['boundary_class']} diag {k['diag']} tail {k.get('tail')} vgprs {k['vgprs']} its {d['config']['mean_newton_iterations_per_step']:.4f} conv {d['config']['all_converged']}\")" || echo n/a
  done
  done
}
run c4 "--workload c4 --reaches 32768 --steps 16 --warmup 2"
run c5f32 "--workload c5 --dtype f32 --nodes 512 --reaches 131072 --steps 32 --warmup 4"
run c5f64 "--workload c5 --dtype f64 --nodes 512 --reaches 131072 --steps 32 --warmup 4"
run irr "--workload irr --reaches 8192 --steps 16 --warmup 2"
