cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputest_g.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_g.log
python bench.py --no-cpu-baseline --no-extra > gpurun_out/r04_bench_g.json 2> gpurun_out/r04_bench_g.err
python tools/sweep_shapes.py > gpurun_out/r04_sweep_library_g.txt 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/r04_gputest_g.log | tail; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_g.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')})"; cat gpurun_out/r04_sweep_library_g.txt
