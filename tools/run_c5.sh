for v in flow-sim_amd/csrc/variants/lib_w*.so; do
  for dt in f32 f64; do
  echo -n "$(basename $v) $dt "
  FS_LIB=$PWD/$v timeout -k 10 200 python bench.py --workload c5 --dtype $dt --nodes 512 --reaches 131072 --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']}\")"
  done
done
