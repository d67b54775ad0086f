cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_gputest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_e.log
python tools/sweep_shapes.py > gpurun_out/r04_sweep_library.txt 2>&1
python tools/sweep_shapes.py --specialise > gpurun_out/r04_sweep_specialised.txt 2>&1
python bench.py --no-cpu-baseline --no-extra > gpurun_out/r04_bench_e.json 2> gpurun_out/r04_bench_e.err
grep -n "FAILED\|passed\|failed" gpurun_out/r04_gputest_e.log | tail -20; paste -d'|' gpurun_out/r04_sweep_library.txt gpurun_out/r04_sweep_specialised.txt | cut -c1-200; head -c 400 gpurun_out/r04_bench_e.json
