# flagship workload at the driver's shape (65 536 x 4 096, 20 timed levels after 5) on every variant library; two runs each; then the state digest
ARGS=${ARGS:---reaches 65536 --steps 20 --warmup 5}
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  echo -n "$(basename $v) "
  FS_LIB=$PWD/$v timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.5g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']['vgprs']} its {d['config']['mean_newton_iterations_per_step']:.4f} conv {d['config']['all_converged']}\")"
done
done
for v in flow-sim_amd/csrc/variants/lib_*.so; do echo -n "$(basename $v) digest "; FS_LIB=$PWD/$v timeout -k 10 100 python tools/variant_digest.py 2>&1 | tail -1; done
