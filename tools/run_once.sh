# one bench.py line, compact: LIB (variant library, optional), SHAPE (FS_KERNEL_SHAPE, optional), ARGS from the environment
[ -n "$LIB" ] && export FS_LIB=$PWD/$LIB
[ -n "$SHAPE" ] && export FS_KERNEL_SHAPE=$SHAPE
timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(f\"{d['value']:.4g} r-ts/s kernel_ms {d['roofline']['kernel_ms']:.2f} {c['kernel']} its {c['mean_newton_iterations_per_step']:.4f} conv {c['all_converged']} status {c.get('status_counts_rank0')}\")" || echo n/a
