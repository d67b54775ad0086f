#!/bin/bash
# Round-3 counter collection for the config-5 kernel on the GPU box:  gpurun -- 'bash tools/collect_c5_pmc.sh [tag]'
# (separate --pmc passes, no trace domains combined with them; builds first, never from a profiled process)
set -e
TAG=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c 'import __graft_entry__ as g; g.build()' > /dev/null
export COPRA_NO_BUILD=1
O=gpurun_out
rm -rf $O/${TAG}_*
C5="python3 tools/try_config5.py 16384 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $C5 > $O/${TAG}_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $C5 >> $O/${TAG}_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $C5 >> $O/${TAG}_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_sq -- $C5 >> $O/${TAG}_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_sq2 -- $C5 >> $O/${TAG}_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/${TAG}_sq3 -- $C5 >> $O/${TAG}_run.log 2>&1 || echo "(third pass: some counter not available)"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/${TAG}_sq4 -- $C5 >> $O/${TAG}_run.log 2>&1 || echo "(fourth pass: some counter not available)"
python tools/pmc_summary.py $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq $O/${TAG}_sq2 $O/${TAG}_sq3 $O/${TAG}_sq4 > $O/${TAG}_rocprof_summary.json
find $O/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
tail -5 $O/${TAG}_run.log
