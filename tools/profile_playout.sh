#!/bin/bash
# rocprofv3 passes over the rules-only step kernel (k_playout) at one batch size: kernel trace, then three
# PMC passes of at most 8 SQ counters (+ GRBM_GUI_ACTIVE, a block of its own) each.
# Usage (inside gpurun): bash tools/profile_playout.sh <tag> <boards> [launches] [waves per board]
set -e
TAG=$1; B=$2; N=${3:-3}; NW=${4:-0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $REPO/tools/playout_prof.py $B $N $NW"
python3 $REPO/tools/playout_prof.py $B $N $NW > $OUT/plain.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $RUN > $OUT/trace_out.json 2> $OUT/trace.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o s -- $RUN > $OUT/sq_out.json 2> $OUT/sq.err
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/i1 -o i -- $RUN > $OUT/i1_out.json 2> $OUT/i1.err
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/i2 -o i -- $RUN > $OUT/i2_out.json 2> $OUT/i2.err
ls $OUT
