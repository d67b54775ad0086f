cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/exp/fuzz_vs_oracle.py 0 3000 48 > gpurun_out/r04_fuzz_n.log 2>&1
tail -25 gpurun_out/r04_fuzz_n.log
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "random_controllers or config5 or riccati or long_horizon or beyond or nine_classes or tracking" > gpurun_out/r04_gputest_n.log 2>&1; tail -8 gpurun_out/r04_gputest_n.log
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_n.json 2> gpurun_out/r04_bench_n.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_n.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}); print({k:(v.get('solves_per_s'),v.get('error')) for k,v in d['extra'].items()}); print(d['extra']['config5_initial_state_12_6_50_riccati_ipm'])"
