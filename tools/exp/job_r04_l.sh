cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/exp/fuzz_vs_oracle.py 600 1400 48 > gpurun_out/r04_fuzz_l2.log 2>&1
tail -30 gpurun_out/r04_fuzz_l2.log
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "random_controllers" > gpurun_out/r04_gputest_l.log 2>&1; tail -5 gpurun_out/r04_gputest_l.log
