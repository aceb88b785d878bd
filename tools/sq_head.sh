# SQ counters of ONE head.conv1-like layer (256 -> 64 at 504x360, 16 pages) through the ring kernel, two --pmc passes -> gpurun_out/sqh/summary.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqh; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -- python3 $R/tools/ring_layer_probe.py ${1:-256} ${2:-64} 504 360 16 ${3:-0} > $O/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --output-format csv -d $O/p2 -- python3 $R/tools/ring_layer_probe.py ${1:-256} ${2:-64} 504 360 16 ${3:-0} > $O/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_VMEM --output-format csv -d $O/p3 -- python3 $R/tools/ring_layer_probe.py ${1:-256} ${2:-64} 504 360 16 ${3:-0} > $O/p3.log 2>&1
cd $R && (python tools/pmc_table.py $O/p1; python tools/pmc_table.py $O/p2; python tools/pmc_table.py $O/p3) > $O/summary.txt 2>&1; cat $O/summary.txt
