#!/bin/bash
# SQ counters of the fused selfplay step against the on-device-policy ply (tools/api_step_prof.py).
# Usage (inside gpurun): bash tools/profile_api_step.sh <tag> [boards]   -> gpurun_out/<tag>/
set -e
TAG=${1:-apistep}; B=${2:-8192}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/api_step_prof.py $B 20"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o k -- $P > $OUT/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/i1 -o i -- $P > $OUT/i1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/i2 -o i -- $P > $OUT/i2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/i3 -o i -- $P > $OUT/i3.log 2>&1
