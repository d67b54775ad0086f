cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputest_k.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_k.log
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_k.json 2> gpurun_out/r04_bench_k.err
grep -n "FAILED\|passed\|failed\|relaxed workload\|first solve\|equal iteration" gpurun_out/r04_gputest_k.log | tail; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_k.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}); print({k:(v.get('solves_per_s'),v.get('error')) for k,v in d['extra'].items()})"
