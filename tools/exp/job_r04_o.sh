cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/exp/debug_integrator_seed.py 6 24576 > gpurun_out/r04_debug_o.log 2>&1
cat gpurun_out/r04_debug_o.log | cut -c1-420
