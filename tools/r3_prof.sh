# round-3 profiles: traffic pair of the flagship (K = 20 and K = 8), polyline ensemble, C4, the (8, 8) shape, marker trace, extra bench lines
mkdir -p gpurun_out/r3 gpurun_out/prof
bash tools/profile.sh c3_f64 --reaches 65536 --steps 20 --warmup 5 && echo "c3_f64 done"
bash tools/profile.sh c3_f64_k8 --reaches 65536 --steps 8 --warmup 5 && echo "c3_f64_k8 done"
bash tools/profile.sh irr_f64 --workload irr --reaches 8192 --steps 16 --warmup 2 && echo "irr done"
bash tools/profile.sh c4_f64 --workload c4 --reaches 32768 --steps 16 --warmup 2 && echo "c4 done"
bash tools/profile.sh long_f64 --nodes 16384 --reaches 8192 --steps 8 --warmup 2 && echo "long done"
FS_KERNEL_SHAPE=8,8 bash tools/profile.sh c3_f64_8x8 --reaches 65536 --steps 20 --warmup 5 && echo "8x8 done"
mkdir -p gpurun_out/prof/markers
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof/markers -- python3 $OLDPWD/tools/bench_pcie.py 8192 > $OLDPWD/gpurun_out/prof/markers/bench.json 2> $OLDPWD/gpurun_out/prof/markers/err.txt ); echo "markers rc $?"
{
echo -n "c4 shared table:    "; ARGS="--workload c4 --reaches 32768 --steps 16 --warmup 2" bash tools/run_once.sh
echo -n "c4 per-reach table: "; ARGS="--workload c4 --reaches 32768 --steps 16 --warmup 2 --per-reach-geometry" bash tools/run_once.sh
echo -n "rect 16384 nodes x 8192 reaches (multi-pass): "; ARGS="--nodes 16384 --reaches 8192 --steps 8 --warmup 2" bash tools/run_once.sh
echo -n "rect 8192 nodes x 16384 reaches (multi-pass): "; ARGS="--nodes 8192 --reaches 16384 --steps 8 --warmup 2" bash tools/run_once.sh
echo -n "c5 f32: "; ARGS="--workload c5 --dtype f32 --nodes 512 --reaches 131072 --steps 32 --warmup 4" bash tools/run_once.sh
echo -n "c5 f64: "; ARGS="--workload c5 --dtype f64 --nodes 512 --reaches 131072 --steps 32 --warmup 4" bash tools/run_once.sh
echo -n "c3 default: "; ARGS="--steps 20 --warmup 5" bash tools/run_once.sh
} 2>&1 | tee gpurun_out/r3/extra_bench.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; tail -c 600 gpurun_out/r3/bench_default.json
