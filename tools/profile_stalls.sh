#!/bin/bash
# where a wave's cycles go: issue vs wait, instruction-cache behaviour, instruction mix
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="--reaches 16384 --steps 8 --warmup 1 --no-cpu-baseline ${BENCH_EXTRA}"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err || true; }
run wait  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
run mix   SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
run mix2  SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
cd $OUT; python3 - <<'PY'
import csv, glob
for d in ("wait","icache","mix","mix2"):
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        rows=[r for r in csv.DictReader(open(f)) if "preissmann" in r["Kernel_Name"]]
        last=max(int(r["Dispatch_Id"]) for r in rows)
        print(d, {r["Counter_Name"]: f'{float(r["Counter_Value"]):.4g}' for r in rows if int(r["Dispatch_Id"])==last})
PY
