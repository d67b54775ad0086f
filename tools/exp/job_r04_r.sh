cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export COPRA_NO_BUILD=1
timeout 600 python tools/exp/ric_variants.py few > gpurun_out/r04_ricvariants_r.log 2>&1
cat gpurun_out/r04_ricvariants_r.log | cut -c1-900
