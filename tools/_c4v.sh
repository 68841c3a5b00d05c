for v in reg1 regall; do for gen in 0 1; do
  echo -n "$v general=$gen "
  FS_KERNEL_GENERAL=$gen FS_LIB=$PWD/flow-sim_amd/csrc/variants/lib_$v.so timeout -k 10 200 python bench.py --workload c4 --reaches 32768 --steps 32 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); its=d['config']['mean_newton_iterations_per_step']; print(f\"{d['value']:.4g} its {its:.2f} kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} conv {d['config']['all_converged']}\")"
done; done
