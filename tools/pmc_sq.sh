#!/bin/bash
# ON THE GPU BOX: wave-level counters of ONE x3 GEMM case (tools/bench_x3.py T3D_ONLY), several --pmc passes, means per kernel.
#   bash tools/pmc_sq.sh fwd:512x256 [kernel-name substring]
case=${1:-fwd:512x256}; sub=${2:-PathX3}
root=$(pwd); cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  T3D_ONLY=$case rocprofv3 --pmc $set --output-format csv -d /tmp/sq_$i -o run -- python3 $root/tools/bench_x3.py > /tmp/sq_$i.log 2>&1
  python3 $root/tools/pmc_one.py /tmp/sq_$i "$sub"
done
