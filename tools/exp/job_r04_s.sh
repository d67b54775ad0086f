cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export COPRA_NO_BUILD=1
timeout 600 python tools/exp/ric_debug_case.py 2 8 > gpurun_out/r04_ricdebug_s.log 2>&1
cat gpurun_out/r04_ricdebug_s.log | cut -c1-400
