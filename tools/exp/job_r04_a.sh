cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputest_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_a.log
python tools/exp/truth_distances.py > gpurun_out/r04_truth_distances.txt 2>&1
python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -5 gpurun_out/r04_gputest_a.log; cat gpurun_out/r04_truth_distances.txt; head -c 1500 gpurun_out/r04_bench_a.json
