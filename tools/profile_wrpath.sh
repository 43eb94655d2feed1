#!/bin/bash
# Where do the plane stores wait?  Stall counters of the vector memory path (TA -> TCP -> TCC) and of the
# address translation (UTCL1) for the generator and the pure-store probes.  Usage (inside gpurun):
#   bash tools/profile_wrpath.sh <tag>
set -e
TAG=${1:-wrpath}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum GRBM_GUI_ACTIVE"
P2="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OUT/gen$i -o g -- python3 $REPO/bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-extra > $OUT/gen$i.json 2> $OUT/gen$i.err
  timeout -s KILL 60 rocprofv3 --pmc $P --output-format csv -d $OUT/map$i -o m -- $REPO/tools/store_bench 1024 2 64 2 > $OUT/map$i.txt 2> $OUT/map$i.err
  timeout -s KILL 60 rocprofv3 --pmc $P --output-format csv -d $OUT/split$i -o s -- $REPO/tools/store_bench 1024 2 64 3 > $OUT/split$i.txt 2> $OUT/split$i.err
  timeout -s KILL 60 rocprofv3 --pmc $P --output-format csv -d $OUT/fill$i -o f -- $REPO/tools/store_bench 65536 2 -1 > $OUT/fill$i.txt 2> $OUT/fill$i.err
done
find $OUT -name "*counter_collection.csv"
