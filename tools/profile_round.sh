#!/bin/bash
# Profiles the bench line's kernel on the GPU box: kernel trace + stats, then the two HBM
# traffic counters in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Usage (inside gpurun): bash tools/profile_round.sh <tag>      outputs under gpurun_out/<tag>/
set -e
TAG=${1:-r01b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-extra"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $BENCH > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $BENCH > $OUT/write_bench.json 2> $OUT/write.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o s -- $BENCH > $OUT/sq_bench.json 2> $OUT/sq.err
find $OUT -name "*.csv" | head -20
