cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/exp/fuzz_ipm_12_6.py 0 120 > gpurun_out/r04_fuzzipm_w.log 2>&1
grep -c "<<<<" gpurun_out/r04_fuzzipm_w.log; grep "<<<<\|mismatching" gpurun_out/r04_fuzzipm_w.log | cut -c1-400 | head -30; tail -3 gpurun_out/r04_fuzzipm_w.log | cut -c1-300
