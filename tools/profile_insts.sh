#!/bin/bash
# Instruction mix of the bench line's kernel (rocprofv3 --pmc, SQ block only, 8 counters per pass).
# Usage (inside gpurun): bash tools/profile_insts.sh <tag> [bench args]   -> gpurun_out/<tag>/
set -e
TAG=${1:-insts}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-extra $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/i1 -o i -- $BENCH > $OUT/i1.json 2> $OUT/i1.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/i2 -o i -- $BENCH > $OUT/i2.json 2> $OUT/i2.err
