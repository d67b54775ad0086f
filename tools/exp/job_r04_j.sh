cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "relax or hundred_iteration" > gpurun_out/r04_gputest_j.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_j.log
grep -n "FAILED\|passed\|failed\|relaxed workload\|Error\|equal iteration" gpurun_out/r04_gputest_j.log | tail
