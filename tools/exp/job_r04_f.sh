cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputest_f.log
python tools/dense_phase_profile.py > gpurun_out/r04_dense_phase_regs.txt 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/r04_gputest_f.log | tail -20; cat gpurun_out/r04_dense_phase_regs.txt
