#!/bin/bash
# Round-6 profile collection on the GPU box:  gpurun -- 'bash tools/collect_profiles_r06.sh'
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
R=profiles/r06
mkdir -p $R
rm -rf $O/hl6_* $O/tl6_* $O/rf6_* $O/jk6_*
BENCH="python3 bench.py --no-cpu-baseline --no-extra"
# ---- headline (BASELINE configs[2], batch 65536): copra_lmpc_axis_kernel (+ copra_lmpc_fused_ric_kernel for the handful it leaves) ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl6_stats -- $BENCH --steps 50 --warmup 2 > $O/hl6_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hl6_fetch -- $BENCH --steps 5 --warmup 1 >> $O/hl6_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hl6_write -- $BENCH --steps 5 --warmup 1 >> $O/hl6_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/hl6_sq -- $BENCH --steps 5 --warmup 1 >> $O/hl6_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/hl6_sq2 -- $BENCH --steps 5 --warmup 1 >> $O/hl6_run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/hl6_sq3 -- $BENCH --steps 5 --warmup 1 >> $O/hl6_run.log 2>&1 || echo "(FP64 class counters not available)"
python tools/pmc_summary.py $O/hl6_stats $O/hl6_fetch $O/hl6_write $O/hl6_sq $O/hl6_sq2 $O/hl6_sq3 > $R/headline_rocprof_summary.json
find $O/hl6_stats -name "*kernel_stats.csv" -exec cp {} $R/headline_kernel_stats.csv \;
# ---- the same command on the round-5 pair (copra_options_t::no_axis_solver) ----
COPRA_OPTIONS=no_axis_solver=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl6_stats_r05pair -- $BENCH --steps 50 --warmup 2 >> $O/hl6_run.log 2>&1
find $O/hl6_stats_r05pair -name "*kernel_stats.csv" -exec cp {} $R/headline_round5_pair_kernel_stats.csv \;
# ---- the tight workload (v_max 0.25 / u_max 1.2): kernel trace ----
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tl6_stats -- python3 tools/exp/axis_phases.py 0.25 1.2 > $O/tl6_run.log 2>&1
find $O/tl6_stats -name "*kernel_stats.csv" -exec cp {} $R/tight_kernel_stats.csv \;
# ---- the widened solver: kernel traces of the reference / limits workloads and of the jerk-controlled model ----
export PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rf6_stats -- python3 tools/exp/axis_tracking_rates.py > $O/rf6_run.log 2>&1 || true
find $O/rf6_stats -name "*kernel_stats.csv" -exec cp {} $R/references_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/jk6_stats -- python3 tools/exp/jerk_model_rates.py > $O/jk6_run.log 2>&1 || true
find $O/jk6_stats -name "*kernel_stats.csv" -exec cp {} $R/jerk_model_kernel_stats.csv \;
# ---- side measurements ----
python tools/exp/axis_phases.py 0.6 3.0 2>&1 | grep -v amdgpu.ids > $R/axis_phases_headline.txt || true
python tools/exp/axis_phases.py 0.4 2.0 2>&1 | grep -v amdgpu.ids > $R/axis_phases_vmax040.txt || true
python tools/exp/axis_phases.py 0.25 1.2 2>&1 | grep -v amdgpu.ids > $R/axis_phases_vmax025.txt || true
python tools/exp/axis_batches.py 2>&1 | grep -v amdgpu.ids > $R/axis_load_sweep.txt || true
python tools/exp/axis_gpu_check.py 2>&1 | grep -v amdgpu.ids > $R/tight_ladder_rates.txt || true
PYTHONPATH=. python tools/exp/axis_tracking_rates.py 2>&1 | grep -v amdgpu.ids > $R/references_rates.txt || true
PYTHONPATH=. python tools/exp/axis_limits_rates.py 2>&1 | grep -v amdgpu.ids >> $R/references_rates.txt || true
PYTHONPATH=. python tools/exp/axis_major_rates.py 2>&1 | grep -v amdgpu.ids > $R/axis_major_rates.txt || true
PYTHONPATH=. python tools/exp/jerk_model_rates.py 2>&1 | grep -v amdgpu.ids > $R/jerk_model_rates.txt || true
PYTHONPATH=. python tools/exp/shared_vs_batch.py 2>&1 | grep -v amdgpu.ids > $R/shared_model_against_instance_by_instance.txt || true
PYTHONPATH=. python tools/exp/config2_rates.py 2>&1 | grep -v amdgpu.ids > $R/config2_rates.txt || true
(hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/exp/store_probe.hip 2> /dev/null && /tmp/store_probe) > $R/store_probe.txt 2>&1 || true
python tests/run_config4_single_gpu.py 2>&1 | grep -v amdgpu.ids | tail -30 > $R/config4_single_gpu.txt || true
python tools/sweep_shapes.py 2>&1 | grep -v amdgpu.ids > $R/shape_sweep.txt || true
python tools/exp/truth_distances.py 2>&1 | grep -v amdgpu.ids > $R/truth_distances.txt || true
# ---- the random differential tests that reach the new kernel, at wide settings ----
(python tests/fuzz/fuzz_integrators.py 0 150 2>&1 | grep -v amdgpu.ids | grep "<<<<\|mismatching" | cut -c1-500) > $R/fuzz_integrator_shapes.txt || true
(python tests/fuzz/fuzz_integrators.py 1000 90 chain3 2>&1 | grep -v amdgpu.ids | grep "<<<<\|mismatching" | cut -c1-500) > $R/fuzz_chains_of_three_states.txt || true
(python tests/fuzz/fuzz_integrators.py 3000 90 chain1 2>&1 | grep -v amdgpu.ids | grep "<<<<\|mismatching" | cut -c1-500) > $R/fuzz_one_state_per_control.txt || true
(python tests/fuzz/fuzz_modes.py 0 420 2>&1 | grep -v amdgpu.ids | grep " <\|ERROR\|mismatching" | cut -c1-400) > $R/fuzz_engine_modes.txt || true
(python tests/fuzz/fuzz_vs_oracle.py 0 1500 48 2>&1 | grep -v amdgpu.ids | tail -40) > $R/fuzz_random_controllers.txt || true
# ---- the bench line itself (with cpu_baseline and extra): AFTER the summaries, so that its roofline.traffic is the one just measured ----
python bench.py --warmup 3 > $O/bench_r06.json 2> $O/bench_r06.err
cp $O/bench_r06.json $R/bench_line_final.json
mkdir -p $O/profiles_r06 && cp $R/* $O/profiles_r06/
tail -c 1500 $O/bench_r06.json
head -6 $R/headline_kernel_stats.csv
