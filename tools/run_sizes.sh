L=$PWD/flow-sim_amd/csrc/variants/lib_m.so
for cfg in "4096 16384" "2049 32768" "1025 65536"; do set -- $cfg
  echo -n "nodes $1 reaches $2: "
  FS_LIB=$L timeout -k 10 200 python bench.py --nodes $1 --reaches $2 --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel']; print(f\"{d['value']:.4g} r-ts/s kernel_ms {d['roofline']['kernel_ms']:.2f} M={k['cells_per_thread']} W={k['waves_per_reach']} its {d['config']['mean_newton_iterations_per_step']:.2f}  node-its/s {d['value']*$1*d['config']['mean_newton_iterations_per_step']:.4g}\")"
done
