cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/exp/fuzz_integrators.py 0 90 > gpurun_out/r04_fuzzint_p.log 2>&1
grep -c "<<<<" gpurun_out/r04_fuzzint_p.log; grep "<<<<\|mismatching" gpurun_out/r04_fuzzint_p.log | cut -c1-600 | head -40
timeout 600 python tools/exp/config5_fallbacks.py > gpurun_out/r04_c5fallbacks_p.log 2>&1; cat gpurun_out/r04_c5fallbacks_p.log
