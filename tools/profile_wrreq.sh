#!/bin/bash
# Write-request mix at the L2 -> memory interface (32-B against 64-B requests, request stalls) of the generator
# and of the pure-store probes, one PMC pass each.  Usage (inside gpurun): bash tools/profile_wrreq.sh <tag>
set -e
TAG=${1:-wrreq}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/gen -o g -- python3 $REPO/bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-extra > $OUT/gen.json 2> $OUT/gen.err
timeout -s KILL 60 rocprofv3 --pmc $PMC --output-format csv -d $OUT/fill -o f -- $REPO/tools/store_bench 65536 2 -1 > $OUT/fill.txt 2> $OUT/fill.err
timeout -s KILL 60 rocprofv3 --pmc $PMC --output-format csv -d $OUT/group -o p -- $REPO/tools/store_bench 1024 2 64 4 > $OUT/group.txt 2> $OUT/group.err
timeout -s KILL 60 rocprofv3 --pmc $PMC --output-format csv -d $OUT/split -o s -- $REPO/tools/store_bench 1024 2 64 3 > $OUT/split.txt 2> $OUT/split.err
find $OUT -name "*counter_collection.csv"
