#!/bin/bash
# SQ counter passes for the TQ kernel (run on the GPU box through gpurun): four separate --pmc runs of at most
# 8 SQ counters (+ 1 GRBM) each, no tracing domains beside them.  usage: tools/run_pmc_tq.sh <workload> [steps] [warmup]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-tq1080p}; STEPS=${2:-3}; WARM=${3:-1}
OUT=$ROOT/gpurun_out/pmctq_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
B="SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH"
C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
D="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
for P in a b c d; do
  case $P in a) CN=$A;; b) CN=$B;; c) CN=$C;; d) CN=$D;; esac
  timeout -s KILL 120 rocprofv3 --pmc $CN -d $OUT/$P --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --lean > $OUT/$P.log 2>&1
  echo "pass $P done"
done
python3 $ROOT/tools/pmc_sq.py $WL $OUT $OUT/pmc_sq.json
