#!/bin/bash
# vector-memory path of a workload (texture addresser, vector L1, L1 -> L2 requests and their latency): tools/profile_mem.sh TAG <bench.py args>
# one rocprofv3 --pmc pass per counter group (never combined with system traces); output under gpurun_out/prof/TAG_mem
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/${TAG}_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$* --no-cpu-baseline"
# (a counter group the hardware cannot collect in one pass makes rocprofv3 abort and then sit in its signal handler: two counters
# of one block per pass, and a time limit on each)
pass() { name=$1; shift; echo "pass $name"; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass lat1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass lat2 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
pass sq SQ_WAVES SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
python3 - <<PY
import csv, glob, collections
for name in ("ta1", "ta2", "tcp1", "tcp2", "tcp3", "lat1", "lat2", "sq"):
    fs = glob.glob("$OUT/%s/**/*_counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no output"); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows = [r for r in rows if "preissmann" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    print(name, {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last})
PY
