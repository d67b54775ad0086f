cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/exp/fuzz_modes.py 0 400 > gpurun_out/r04_fuzzmodes_y.log 2>&1
grep -c "<<<<\|ERROR" gpurun_out/r04_fuzzmodes_y.log; grep "<<<<\|ERROR\|mismatching" gpurun_out/r04_fuzzmodes_y.log | cut -c1-420 | head -40
