cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/exp/debug_seed.py 1920 > gpurun_out/r04_debug_m.log 2>&1
python tools/exp/debug_seed.py 1781 >> gpurun_out/r04_debug_m.log 2>&1
python tools/exp/debug_seed.py 1640 >> gpurun_out/r04_debug_m.log 2>&1
cat gpurun_out/r04_debug_m.log | cut -c1-400
