cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/hl_* gpurun_out/final_*
python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; tail -3 gpurun_out/final_pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hl_stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/hl_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/hl_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/hl_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/hl_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/hl_sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/hl_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/hl_sq2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/hl_run.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d gpurun_out/hl_ic -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> gpurun_out/hl_run.log 2>&1
python tools/pmc_summary.py gpurun_out/hl_stats gpurun_out/hl_fetch gpurun_out/hl_write gpurun_out/hl_sq gpurun_out/hl_sq2 gpurun_out/hl_ic > gpurun_out/hl_summary.json
find gpurun_out/hl_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/hl_kernel_stats.csv \;
cp gpurun_out/hl_summary.json profiles/r01/headline_rocprof_summary_final.json
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 1700 gpurun_out/bench_final.json
head -4 gpurun_out/hl_kernel_stats.csv
python tools/phase_profile.py > gpurun_out/final_phase.txt 2>&1; tail -16 gpurun_out/final_phase.txt
python tools/sweep_shapes.py > gpurun_out/final_sweep.txt 2>&1; echo "== factor-only tier off" >> gpurun_out/final_sweep.txt; COPRA_NO_TRI=1 python tools/sweep_shapes.py 2>&1 | tail -11 >> gpurun_out/final_sweep.txt; echo "== copra_batch_specialise" >> gpurun_out/final_sweep.txt; python tools/sweep_shapes.py --specialise 2>&1 | tail -11 >> gpurun_out/final_sweep.txt; cat gpurun_out/final_sweep.txt | grep -v amdgpu.ids
python tools/bench_shared.py 65536 20 2>&1 | tail -1 > gpurun_out/final_shared.json; cat gpurun_out/final_shared.json
python tools/try_small.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final_small.txt; cat gpurun_out/final_small.txt
