#!/bin/bash
# robustness evidence kept under profiles/: populations of the C5 / C3 draws far beyond the benchmark's (other seeds),
# every reach must end FS_OK; then the flow-regime grid against the pivoted LU of the C oracle
set -e
OUT=gpurun_out/scans; mkdir -p $OUT
: > $OUT/robustness_scan.txt
for seed in 1 2 3 4 5 6 7 8; do
  python tools/find_bad_reach.py --dtype f32 --nodes 512 --reaches 1048576 --seed $((20260214 + seed)) | grep "^kernel" >> $OUT/robustness_scan.txt
done
for seed in 1 2; do
  python tools/find_bad_reach.py --dtype f64 --nodes 512 --reaches 262144 --seed $((20260214 + seed)) | grep "^kernel" >> $OUT/robustness_scan.txt
done
python tools/find_bad_reach.py --workload c3 --dtype f64 --nodes 4096 --reaches 65536 --dx 250 --dt 600 --seed 20260299 | grep "^kernel" >> $OUT/robustness_scan.txt
python tools/find_bad_reach.py --workload c3 --dtype f32 --nodes 512 --reaches 262144 --dx 250 --dt 600 --seed 20260299 | grep "^kernel" >> $OUT/robustness_scan.txt
python tools/scan_regimes.py > $OUT/regime_scan.txt
tail -n 3 $OUT/regime_scan.txt
