# SQ counters of ONE stage-0 layer (64 -> 64 at 504x360, 16 pages, residual) through the ring kernel -> profiles/r02_pmc_sq_stage0_layer.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq0; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/ring_layer_probe.py 64 64 504 360 16 1 > $O/p1.log 2>&1 &&
cd $R && python tools/pmc_table.py $O/p1 > $O/summary.txt && cat $O/summary.txt
