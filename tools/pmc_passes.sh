#!/bin/bash
# usage: pmc_passes.sh <outdir> <python script and args...>   — separate rocprofv3 --pmc passes (counters only, no tracing)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY" \
           "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 "$@" > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
