cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/exp/config5_fallbacks.py > gpurun_out/r04_c5fallbacks_u.log 2>&1; cat gpurun_out/r04_c5fallbacks_u.log
timeout 1200 python tools/exp/fuzz_vs_oracle.py 0 3000 48 > gpurun_out/r04_fuzz_u.log 2>&1
tail -16 gpurun_out/r04_fuzz_u.log | cut -c1-330
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "config5 or riccati or long_horizon or beyond or nine_classes or random" > gpurun_out/r04_gputest_u.log 2>&1; tail -4 gpurun_out/r04_gputest_u.log
