cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_gputest_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_b.log
python bench.py > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
grep -n "FAILED\|passed\|failed\|distance from\|first solve" gpurun_out/r04_gputest_b.log | tail -40; head -c 600 gpurun_out/r04_bench_b.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_b.json'))
print({k:d[k] for k in ('value','ms_per_step','kernel_ms','max_rel_u_err','max_rel_x_err','iterations_agree','status_agree') if k in d})
print(d['roofline'].get('traffic_stale'), d.get('cpu_baseline',{}).get('value'))
for k,v in d.get('extra',{}).items():
    print(k, {a:b for a,b in v.items() if a in ('solves_per_s','kernel_ms','first_solve_ms','first_solve_over_steady','first_solve_active_capacity','steady_active_capacity','error','median_us')})
PY
