#!/bin/bash
# Profiles a command's kernels on the GPU box: kernel trace + stats, then the two HBM traffic
# counters and the SQ wave-state counters in passes of their own (MI355X_MICROARCH.md, rocprofv3
# PMC slots).  Default command = the driver's bench line.
# Usage (inside gpurun): bash tools/profile_round.sh <tag> [program args...]   -> gpurun_out/<tag>/
#   e.g. bash tools/profile_round.sh r02a
#        bash tools/profile_round.sh r02g8 bench.py --boards 8192 --steps 8 --warmup 2 --reps 1 --no-cpu-baseline --no-extra
#        bash tools/profile_round.sh r02p8 tools/ply_prof.py 8192 40
set -e
TAG=${1:-r02a}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- bench.py --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-extra; fi
PROG=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
RUN="python3 $PROG $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $RUN > $OUT/trace_out.json 2> $OUT/trace.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $RUN > $OUT/fetch_out.json 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $RUN > $OUT/write_out.json 2> $OUT/write.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o s -- $RUN > $OUT/sq_out.json 2> $OUT/sq.err
find $OUT -name "*.csv" | head -20
