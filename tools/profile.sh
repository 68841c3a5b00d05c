#!/bin/bash
# rocprofv3 passes for profiles/round4:  tools/profile.sh TAG <bench.py arguments ...>
#   kernel trace + stats, then the HBM traffic counters in separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
#   do not fit one pass; counters are never combined with system traces), the issue counters and the flop counters.
# Everything lands under gpurun_out/prof/TAG; tools/refresh_profiles.py TAG condenses it into profiles/round4/.
set -e; mkdir -p $(pwd)/gpurun_out/prof
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$* --no-cpu-baseline --no-extras"
echo "$ARGS" > $OUT/args.txt
sha256sum $ROOT/flow-sim_amd/csrc/libflowsim_hip.so | cut -d' ' -f1 > $OUT/library_sha256.txt
pass() { name=$1; shift; rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/$name.err || true; }
pass trace --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
pass pmc_wait --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
if [ "${FS_F32:-0}" = "1" ]; then
  pass pmc_flops --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32
else
  pass pmc_flops --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64
fi
find $OUT -name "*.csv" | wc -l
du -sh $OUT
