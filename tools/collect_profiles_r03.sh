#!/bin/bash
# Round-3 profile collection on the GPU box:  gpurun -- 'bash tools/collect_profiles_r03.sh'
# Builds first and forbids rebuilding afterwards: rocprofv3 preloads a library that initialises the GPU in every child,
# so make -> hipcc must never be spawned from a profiled process (COPRA_NO_BUILD makes the loader raise instead).
# Counters in their own --pmc passes (never combined with trace domains), summaries copied to profiles/r03/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c 'import __graft_entry__ as g; g.build()' > /dev/null
python -c 'import sys; sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); import test_cpp_api; test_cpp_api._build()' > /dev/null 2>&1 || true
export COPRA_NO_BUILD=1
O=gpurun_out
R=profiles/r03
mkdir -p $R
rm -rf $O/hl_* $O/c5_*
BENCH="python3 bench.py --no-cpu-baseline --no-extra"
# ---- headline (BASELINE configs[2], batch 65536) ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl_stats -- $BENCH --steps 20 --warmup 2 > $O/hl_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hl_fetch -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hl_write -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/hl_sq -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/hl_sq2 -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/hl_sq3 -- $BENCH --steps 5 --warmup 1 >> $O/hl_run.log 2>&1 || echo "(FP64 class counters not available)"
python tools/pmc_summary.py $O/hl_stats $O/hl_fetch $O/hl_write $O/hl_sq $O/hl_sq2 $O/hl_sq3 > $R/headline_rocprof_summary.json
find $O/hl_stats -name "*kernel_stats.csv" -exec cp {} $R/headline_kernel_stats.csv \;
# ---- config 5 (InitialStateLMPC 12/6/50, batch 16384): the LDS-resident Riccati interior-point kernel ----
C5="python3 tools/try_config5.py 16384 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- $C5 > $O/c5_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5_write -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/c5_sq -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c5_sq2 -- $C5 >> $O/c5_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/c5_sq3 -- $C5 >> $O/c5_run.log 2>&1 || echo "(FP64 class counters not available)"
python tools/pmc_summary.py $O/c5_stats $O/c5_fetch $O/c5_write $O/c5_sq $O/c5_sq2 $O/c5_sq3 > $R/config5_rocprof_summary.json
find $O/c5_stats -name "*kernel_stats.csv" -exec cp {} $R/config5_kernel_stats.csv \;
python tools/riccati_mfma_profile.py > $R/config5_phase_cycles.txt 2>&1
COPRA_OPTIONS=no_ric_fast=1 python tools/try_config5.py 16384 0 2>&1 | grep -E "solver|batch|status" > $R/config5_streaming_kernel.txt
# ---- dense-Hessian (MFMA 16x16x4) path: MFMA-busy share ----
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/hl_dense -- $BENCH --dense-hessian --steps 3 --warmup 1 >> $O/hl_run.log 2>&1
python tools/pmc_summary.py $O/hl_dense > $R/pmc_dense_mfma_path.json
# ---- the one-instance-per-lane pass (lmpc_lane.hpp): with / without it, its parts, the first tier behind it, the shared-model tick, batch sizes ----
python tools/exp/lane_pass_check.py 2>&1 | grep -v amdgpu.ids > $R/lane_pass_check.txt || true
VARIANTS=0,1,2,3 PHASE_DBG=8,9,10,11 python tools/exp/lane_pass_variants.py 2>&1 | grep -v amdgpu.ids > $R/lane_pass_parts.txt || true
python tools/exp/lane_tier1_phases.py 2>&1 | grep -v amdgpu.ids > $R/lane_tier1_phases.txt || true
python tools/exp/lane_shared_check.py 2>&1 | grep -v amdgpu.ids > $R/lane_shared_tick.txt || true
for b in 2048 8192 16384 24576 32768; do echo "batch $b"; COPRA_OPTIONS=lane_min_batch=1 python tools/exp/lane_pass_check.py $b 2>&1 | grep -v amdgpu.ids | head -1; done > $R/lane_batch_sweep.txt || true
python tools/exp/lane_threshold.py 2>&1 | grep -v amdgpu.ids > $R/lane_threshold.txt || true
python tools/exp/selection_rows.py 2>&1 | grep -v amdgpu.ids > $R/selection_rows.txt || true
python tools/exp/terminal_rows.py 2>&1 | grep -v amdgpu.ids > $R/terminal_rows.txt || true
python tools/exp/reference_trajectory.py 2>&1 | grep -v amdgpu.ids > $R/reference_trajectory.txt || true
rm -rf $O/rt_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rt_stats -- python3 tools/exp/reference_trajectory.py profile > $O/rt_run.log 2>&1 || true
find $O/rt_stats -name "*kernel_stats.csv" -exec cp {} $R/reference_trajectory_kernel_stats.csv \;
# ---- probes and side measurements ----
tools/exp/mfma_chain_probe > $R/mfma_chain_probe.txt 2>&1 || true
python tools/pcie_probe.py 2>&1 | grep -v amdgpu.ids > $R/pcie_probe.txt
python tools/host_pipeline.py 2>&1 | grep -v amdgpu.ids > $R/host_pipeline.txt
python tools/tight_ladder_rates.py 2>&1 | grep -v amdgpu.ids > $R/tight_ladder_rates.txt || true
python tools/latency_small_batches.py 2>&1 | grep -v amdgpu.ids > $R/latency_small_batches.txt || true
python tools/sweep_shapes.py 2>&1 | grep -v amdgpu.ids > $R/shape_sweep.txt || true
python tools/sweep_shapes.py --specialise 2>&1 | grep -v amdgpu.ids > $R/shape_sweep_specialised.txt || true
cp $R/*.json $R/*.csv $R/*.txt $O/ 2>/dev/null || true
# ---- the bench line itself (with cpu_baseline and extra) ----
python bench.py --steps 20 --warmup 3 > $O/bench_r03.json 2> $O/bench_r03.err
cp $O/bench_r03.json $R/bench_line_final.json
tail -c 1500 $O/bench_r03.json
head -4 $R/headline_kernel_stats.csv
head -4 $R/config5_kernel_stats.csv
