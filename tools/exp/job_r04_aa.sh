cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python tools/exp/fuzz_specialise.py 0 30 > gpurun_out/r04_fuzzjit_aa.log 2>&1
cat gpurun_out/r04_fuzzjit_aa.log | cut -c1-420 | tail -34
timeout 600 python tools/exp/fuzz_vs_oracle.py 3000 2000 2 > gpurun_out/r04_fuzz_b2_aa.log 2>&1; tail -8 gpurun_out/r04_fuzz_b2_aa.log | cut -c1-300
