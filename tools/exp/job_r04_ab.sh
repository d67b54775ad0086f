cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sh in "5 3" "7 2" "3 3" "6 1" "4 2"; do
  echo "== shape $sh"
  timeout 900 python tools/exp/fuzz_ipm_12_6.py 0 60 $sh > gpurun_out/r04_fuzzipm_ab.log 2>&1
  grep "<<<<\|mismatching\|Error\|error" gpurun_out/r04_fuzzipm_ab.log | cut -c1-400 | head -12
done
