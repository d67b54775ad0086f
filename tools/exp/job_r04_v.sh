cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/exp/fuzz_vs_oracle.py 0 3000 48 > gpurun_out/r04_fuzz_v.log 2>&1
tail -14 gpurun_out/r04_fuzz_v.log | cut -c1-330
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputest_v.log 2>&1; grep -n "FAILED\|passed\|failed\|random controllers" gpurun_out/r04_gputest_v.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_v.json 2> gpurun_out/r04_bench_v.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_v.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}); print({k:(v.get('solves_per_s'),v.get('error')) for k,v in d['extra'].items()}); print(d['extra']['config5_initial_state_12_6_50_riccati_ipm'])"
