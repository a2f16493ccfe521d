#!/bin/bash
# SQ counter passes for bench.py (run on the GPU box through gpurun): three separate --pmc runs of at most
# 7 SQ + 1 GRBM counters each, no tracing domains beside them.  usage: tools/run_pmc_sq.sh <workload> <steps> <warmup>
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-1080p}; STEPS=${2:-2}; WARM=${3:-1}
OUT=$ROOT/gpurun_out/pmcsq_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
timeout -s KILL 240 rocprofv3 --pmc $A -d $OUT/a --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/a.log 2>&1
echo "pass A done"
timeout -s KILL 240 rocprofv3 --pmc $B -d $OUT/b --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/b.log 2>&1
echo "pass B done"
timeout -s KILL 240 rocprofv3 --pmc $C -d $OUT/c --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/c.log 2>&1
echo "pass C done"
python3 $ROOT/tools/pmc_sq.py $WL $OUT $OUT/pmc_sq.json
