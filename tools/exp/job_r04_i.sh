cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputest_i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_i.log
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_i.json 2> gpurun_out/r04_bench_i.err
grep -n "FAILED\|passed\|failed\|relaxed workload\|first solve" gpurun_out/r04_gputest_i.log | tail; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_i.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}); print({k:(v.get('solves_per_s'),v.get('error')) for k,v in d['extra'].items()})"
