#!/bin/bash
# Round-2 profile collection on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh'
# Builds first and forbids rebuilding afterwards: rocprofv3 preloads a library that initialises the GPU in every child,
# so make -> hipcc must never be spawned from a profiled process (COPRA_NO_BUILD makes the loader raise instead).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c 'import __graft_entry__ as g; g.build()' > /dev/null
export COPRA_NO_BUILD=1
O=gpurun_out
rm -rf $O/hl_* $O/c5_*
BENCH="python3 bench.py --no-cpu-baseline --no-extra"
# ---- headline (BASELINE configs[2], batch 65536) ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl_stats -- $BENCH --steps 20 --warmup 2 > $O/hl_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hl_fetch -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hl_write -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/hl_sq -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/hl_sq2 -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/hl_sq3 -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1 || echo "(FP64 class counters not available)"
python tools/pmc_summary.py $O/hl_stats $O/hl_fetch $O/hl_write $O/hl_sq $O/hl_sq2 $O/hl_sq3 > profiles/r02/headline_rocprof_summary.json
find $O/hl_stats -name "*kernel_stats.csv" -exec cp {} profiles/r02/headline_kernel_stats.csv \;
# ---- config 5 (InitialStateLMPC 12/6/50, batch 16384, Riccati interior-point kernel) ----
C5="python3 tools/try_config5.py 16384 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- $C5 > $O/c5_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5_write -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/c5_sq -- $C5 >> $O/c5_run.log 2>&1
python tools/pmc_summary.py $O/c5_stats $O/c5_fetch $O/c5_write $O/c5_sq > profiles/r02/config5_rocprof_summary.json
find $O/c5_stats -name "*kernel_stats.csv" -exec cp {} profiles/r02/config5_kernel_stats.csv \;
cp profiles/r02/*.json profiles/r02/*.csv $O/ 2>/dev/null || true
# ---- the bench line itself (with cpu_baseline and extra) ----
python bench.py --steps 20 --warmup 3 > $O/bench_r02.json 2> $O/bench_r02.err
tail -c 3000 $O/bench_r02.json
head -5 profiles/r02/headline_kernel_stats.csv
head -5 profiles/r02/config5_kernel_stats.csv
