#!/bin/bash
# Round-5 profile collection on the GPU box:  gpurun -- 'bash tools/collect_profiles_r05.sh'
# Builds first and forbids rebuilding afterwards: rocprofv3 preloads a library that initialises the GPU in every child, so make -> hipcc
# must never be spawned from a profiled process (COPRA_NO_BUILD makes the loader raise instead).  Counters in their own --pmc passes (never
# combined with trace domains); every summary records the source hash of the library it was taken on (tools/pmc_summary.py), which
# bench.py compares with the loaded library (roofline.traffic_stale).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c 'import __graft_entry__ as g; g.build()' > /dev/null
python -c 'import sys; sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); import test_cpp_api; test_cpp_api._build()' > /dev/null 2>&1 || true
export COPRA_NO_BUILD=1
O=gpurun_out
R=profiles/r05
mkdir -p $R
rm -rf $O/hl5_* $O/c55_* $O/dn5_*
BENCH="python3 bench.py --no-cpu-baseline --no-extra"
# ---- headline (BASELINE configs[2], batch 65536): the pair copra_lmpc_lane_kernel + copra_lmpc_fused_ric_kernel ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl5_stats -- $BENCH --steps 20 --warmup 2 > $O/hl5_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hl5_fetch -- $BENCH --steps 5 --warmup 1 >> $O/hl5_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hl5_write -- $BENCH --steps 5 --warmup 1 >> $O/hl5_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/hl5_sq -- $BENCH --steps 5 --warmup 1 >> $O/hl5_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/hl5_sq2 -- $BENCH --steps 5 --warmup 1 >> $O/hl5_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/hl5_sq3 -- $BENCH --steps 5 --warmup 1 >> $O/hl5_run.log 2>&1 || echo "(FP64 class counters not available)"
python tools/pmc_summary.py $O/hl5_stats $O/hl5_fetch $O/hl5_write $O/hl5_sq $O/hl5_sq2 $O/hl5_sq3 > $R/headline_rocprof_summary.json
find $O/hl5_stats -name "*kernel_stats.csv" -exec cp {} $R/headline_kernel_stats.csv \;
# ---- config 5 (InitialStateLMPC 12/6/50, batch 16384): the LDS-resident Riccati interior-point kernel ----
C5="python3 tools/try_config5.py 16384 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c55_stats -- $C5 > $O/c55_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c55_fetch -- $C5 >> $O/c55_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c55_write -- $C5 >> $O/c55_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/c55_sq -- $C5 >> $O/c55_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c55_sq2 -- $C5 >> $O/c55_run.log 2>&1
python tools/pmc_summary.py $O/c55_stats $O/c55_fetch $O/c55_write $O/c55_sq $O/c55_sq2 > $R/config5_rocprof_summary.json
find $O/c55_stats -name "*kernel_stats.csv" -exec cp {} $R/config5_kernel_stats.csv \;
# ---- the dense Psi' W Psi path (what configs[2] names): kernel stats, MFMA counters, HBM traffic, phase split ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dn5_stats -- $BENCH --dense-hessian --steps 5 --warmup 1 > $O/dn5_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/dn5_sq -- $BENCH --dense-hessian --steps 3 --warmup 1 >> $O/dn5_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dn5_fetch -- $BENCH --dense-hessian --steps 3 --warmup 1 >> $O/dn5_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dn5_write -- $BENCH --dense-hessian --steps 3 --warmup 1 >> $O/dn5_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/dn5_sq2 -- $BENCH --dense-hessian --steps 3 --warmup 1 >> $O/dn5_run.log 2>&1
python tools/pmc_summary.py $O/dn5_stats $O/dn5_sq $O/dn5_fetch $O/dn5_write $O/dn5_sq2 > $R/dense_path_rocprof_summary.json
python tools/dense_phase_profile.py 2>&1 | grep -v amdgpu.ids > $R/dense_path_phase_cycles.txt || true
# ---- side measurements ----
python tools/exp/lane_tier1_phases.py 2>&1 | grep -v amdgpu.ids > $R/lane_tier1_phases.txt || true
python tools/tight_ladder_rates.py 2>&1 | grep -v amdgpu.ids > $R/tight_ladder_rates.txt || true
python tools/exp/truth_distances.py 2>&1 | grep -v amdgpu.ids > $R/truth_distances.txt || true
python tests/run_config4_single_gpu.py 2>&1 | grep -v amdgpu.ids | tail -30 > $R/config4_single_gpu.txt || true
for b in 8192 16384 24576 32768; do echo "batch $b"; COPRA_OPTIONS=lane_min_batch=1 python tools/exp/lane_pass_check.py $b 2>&1 | grep -v amdgpu.ids | head -1; done > $R/lane_batch_sweep.txt || true
python tools/sweep_shapes.py 2>&1 | grep -v amdgpu.ids > $R/shape_sweep.txt || true
python tools/sweep_shapes.py --specialise 2>&1 | grep -v amdgpu.ids > $R/shape_sweep_specialised.txt || true
# ---- the random differential tests at their wide settings (the GPU suite runs the first few hundred seeds of each) ----
(python tests/fuzz/fuzz_vs_oracle.py 0 3000 48 2>&1 | grep -v amdgpu.ids | tail -40) > $R/fuzz_random_controllers.txt || true
(python tests/fuzz/fuzz_integrators.py 0 150 2>&1 | grep -v amdgpu.ids | grep "<<<<\|mismatching" | cut -c1-500) > $R/fuzz_integrator_shapes.txt || true
(for sh in "12 6" "5 3" "7 2" "3 3" "6 1" "4 2"; do echo "== shape $sh"; python tests/fuzz/fuzz_interior_point.py 0 60 $sh 2>&1 | grep -v amdgpu.ids | grep "certified\|<<<<\|mismatching" | cut -c1-420; done) > $R/fuzz_interior_point_kernels.txt || true
(python tests/fuzz/fuzz_shared_general_rows.py 0 300 2>&1 | grep -v amdgpu.ids) > $R/fuzz_shared_general_rows.txt || true
python tools/exp/shared_general_rows.py 2>&1 | grep -v amdgpu.ids > $R/shared_general_rows.txt || true
python tools/exp/shared_tracking.py 2>&1 | grep -v amdgpu.ids > $R/shared_tracking.txt || true
python tools/exp/shared_goals.py 2>&1 | grep -v amdgpu.ids > $R/shared_goals.txt || true
python tools/exp/shared_goals_batches.py 2>&1 | grep -v amdgpu.ids > $R/shared_goals_batches.txt || true
python tools/exp/tier_choice_map.py 2>&1 | grep -v amdgpu.ids > $R/tier_choice_map.txt || true
python tools/exp/shared_tick_shapes.py 2>&1 | grep -v amdgpu.ids > $R/shared_tick_shapes.txt || true
(python tests/fuzz/fuzz_shared_general_rows.py 0 300 1024 integrators 2>&1 | grep -v amdgpu.ids) > $R/fuzz_shared_integrators.txt || true
(python tests/fuzz/fuzz_shared_general_rows.py 300 60 32768 integrators 2>&1 | grep -v amdgpu.ids) > $R/fuzz_shared_integrators_32768.txt || true
(python tests/fuzz/fuzz_shared_general_rows.py 0 240 24576 integrators-refs 2>&1 | grep -v amdgpu.ids) > $R/fuzz_shared_integrators_refs.txt || true
(python tests/fuzz/fuzz_dense_qp.py 1000 800 2>&1 | grep -v amdgpu.ids) > $R/fuzz_dense_qp.txt || true
(python tests/fuzz/fuzz_modes.py 0 400 2>&1 | grep -v amdgpu.ids | grep " <\|ERROR\|mismatching" | cut -c1-400) > $R/fuzz_engine_modes.txt || true
# ---- the bench line itself (with cpu_baseline and extra): AFTER the summaries, so that its roofline.traffic is the one just measured ----
python bench.py --steps 20 --warmup 3 > $O/bench_r05.json 2> $O/bench_r05.err
cp $O/bench_r05.json $R/bench_line_final.json
# (only gpurun_out/ travels back from the box: the summaries go there as well, under their own directory)
mkdir -p $O/profiles_r05 && cp $R/* $O/profiles_r05/
tail -c 1200 $O/bench_r05.json
head -5 $R/headline_kernel_stats.csv
head -4 $R/config5_kernel_stats.csv
