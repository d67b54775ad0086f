cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/exp/fuzz_modes.py 0 160 > gpurun_out/r04_fuzzmodes_x.log 2>&1
grep -c "<<<<\|ERROR" gpurun_out/r04_fuzzmodes_x.log; grep "<<<<\|ERROR\|mismatching" gpurun_out/r04_fuzzmodes_x.log | cut -c1-420 | head -40
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "interior_point_kernel" > gpurun_out/r04_gputest_x.log 2>&1; tail -4 gpurun_out/r04_gputest_x.log
