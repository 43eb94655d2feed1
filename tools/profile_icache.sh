#!/bin/bash
# Instruction-cache and scalar-cache counters of the launch-per-ply kernel (tools/ply_prof.py).
# Usage (inside gpurun): bash tools/profile_icache.sh <tag> [boards] [launches]
set -e
TAG=${1:-icache}; B=${2:-1024}; N=${3:-200}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/ply_prof.py $B $N"
timeout -s KILL 60 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/c1 -o i -- $P > $OUT/c1.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/c2 -o i -- $P > $OUT/c2.log 2>&1 || true
