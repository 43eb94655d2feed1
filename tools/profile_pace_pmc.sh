#!/bin/bash
# L2 <-> memory request counters of k_generate_pipe at three store paces (free-running, overdriven, sustained): one PMC pass per pace and
# counter group.  Usage (inside gpurun): bash tools/profile_pace_pmc.sh <tag> [pace_ns ...]
set -e
TAG=${1:-pacepmc}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GOENV_GEN_PIPE=1
G1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum"
G2="TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"
G3="TCC_HIT_sum TCC_MISS_sum TCC_WRITE_REQ_sum TCC_TAG_STALL_sum"
for pace in ${@:-0 6000 6600}; do
  export GOENV_GEN_PACE_NS=$pace
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $G --output-format csv -d $OUT/p${pace}_g$i -o c -- python3 $REPO/tools/time_pipe_variants.py --one $REPO/elf_amd/libgoenv.so 1024 64 > $OUT/p${pace}_g$i.txt 2> $OUT/p${pace}_g$i.err
  done
done
find $OUT -name "*counter_collection.csv" | wc -l
