# C4 workload (32768-member gerd_roseires ensemble, 32 steps) on every variant library; two runs each
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "$(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 200 python bench.py --workload c4 --reaches 32768 --steps 32 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']['vgprs']} conv {d['config']['all_converged']}\")"
done
done
