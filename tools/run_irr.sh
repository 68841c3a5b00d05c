# polyline-channel ensemble on every variant library
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "$(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 300 python bench.py --workload irr --reaches 8192 --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} conv {d['config']['all_converged']}\")" || echo n/a
done
