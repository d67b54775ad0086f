cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/exp/ric_variants.py > gpurun_out/r04_ricvariants_t.log 2>&1
grep -c "BAD" gpurun_out/r04_ricvariants_t.log; grep "BAD" gpurun_out/r04_ricvariants_t.log | cut -c1-500 | head
timeout 900 python tools/exp/fuzz_integrators.py 0 150 > gpurun_out/r04_fuzzint_t.log 2>&1
grep -c "<<<<" gpurun_out/r04_fuzzint_t.log; grep "<<<<\|mismatching" gpurun_out/r04_fuzzint_t.log | cut -c1-600 | head -20
timeout 600 python tools/exp/config5_fallbacks.py > gpurun_out/r04_c5fallbacks_t.log 2>&1; cat gpurun_out/r04_c5fallbacks_t.log
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputest_t.log 2>&1; grep -n "FAILED\|passed\|failed\|random controllers" gpurun_out/r04_gputest_t.log | tail -12
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_t.json 2> gpurun_out/r04_bench_t.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_t.json')); print({k:d[k] for k in ('value','ms_per_step','kernel_ms')}); print({k:(v.get('solves_per_s'),v.get('error')) for k,v in d['extra'].items()}); print(d['extra']['config5_initial_state_12_6_50_riccati_ipm'])"
