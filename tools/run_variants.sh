# flagship workload (16384 x 4096, 32 steps) on every variant library; two runs each
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "$(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 200 python bench.py --reaches 16384 --steps 32 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']['vgprs']} conv {d['config']['all_converged']}\")"
done
done
for v in flow-sim_amd/csrc/variants/lib_*.so; do echo -n "$(basename $v) digest "; FS_LIB=$PWD/$v timeout -k 10 100 python tools/variant_digest.py 2>&1 | tail -1; done
