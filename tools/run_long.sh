# multi-pass kernel on every variant library: 8 192 reaches x 16 384 nodes and 16 384 x 8 192, 8 levels, two runs each; then the state digest of a small batch
for rep in 1 2; do
for v in flow-sim_amd/csrc/variants/lib_*.so; do
  for shape in "16384 8192" "8192 16384"; do
    set -- $shape
    echo -n "$(basename $v) nodes $1 reaches $2: "
    FS_LIB=$PWD/$v timeout -k 10 200 python bench.py --nodes $1 --reaches $2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.4g} r-ts/s  kernel_ms {d['roofline']['kernel_ms']:.2f} {d['config']['kernel']['vgprs']} conv {d['config']['all_converged']}\")"
  done
done
done
for v in flow-sim_amd/csrc/variants/lib_*.so; do echo -n "$(basename $v) digest "; FS_LIB=$PWD/$v FS_DIGEST_NODES=6000 timeout -k 10 100 python tools/variant_digest.py 2>&1 | tail -1; done
