cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_gputest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_d.log
python tools/dense_phase_profile.py > gpurun_out/r04_dense_phase_new.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err
grep -n "FAILED\|passed\|failed\|first solve" gpurun_out/r04_gputest_d.log | tail -20; cat gpurun_out/r04_dense_phase_new.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_d.json'))
print({k:d[k] for k in ('value','ms_per_step','kernel_ms') if k in d})
for k,v in d.get('extra',{}).items():
    print(k, {a:b for a,b in v.items() if a in ('solves_per_s','kernel_ms','first_solve_ms','first_solve_over_steady','first_solve_active_capacity','steady_active_capacity','error','median_us')})
PY
