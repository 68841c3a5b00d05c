# C4 (TABLE, 121 nodes) and C5 (TRAP_UNIFORM, 512 nodes, fp64 and fp32) on every variant library
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  for args in "--workload c4 --reaches 32768" "--workload c5 --nodes 512 --reaches 131072" "--workload c5 --dtype f32 --nodes 512 --reaches 131072"; do
    echo -n "$(basename $v) [$args] "
    FS_LIB=$PWD/$v timeout -k 10 300 python bench.py $args --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']} conv {d['config']['all_converged']}\")"
  done
done
