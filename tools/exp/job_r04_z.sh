cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputest_z.log 2>&1; grep -n "FAILED\|passed\|failed\|random\|controllers over" gpurun_out/r04_gputest_z.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke_z.log 2>&1; tail -3 gpurun_out/r04_smoke_z.log
