mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x -k "irr or polyline or irregular or sweep" > gpurun_out/r3/gputest7.log 2>&1; echo "irr tests rc $?"; tail -5 gpurun_out/r3/gputest7.log
for rep in 1 2; do
echo -n "table: "; ARGS="--workload irr --reaches 8192 --steps 16 --warmup 2" bash tools/run_once.sh
echo -n "walk:  "; FS_POLY_WALK=1 ARGS="--workload irr --reaches 8192 --steps 16 --warmup 2" bash tools/run_once.sh
done 2>&1 | tee gpurun_out/r3/irr_table_vs_walk.txt
