cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/exp/ric_variants.py > gpurun_out/r04_ricvariants_q.log 2>&1
cat gpurun_out/r04_ricvariants_q.log | cut -c1-900
