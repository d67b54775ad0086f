cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r04_fuzzipm_ac.log
for sh in "12 6" "5 3" "7 2" "3 3" "6 1" "4 2"; do
  echo "== shape $sh" >> gpurun_out/r04_fuzzipm_ac.log
  timeout 1500 python tools/exp/fuzz_ipm_12_6.py 0 60 $sh 2>&1 | grep -v amdgpu.ids | grep "certified\|<<<<\|mismatching\|Error\|error" | cut -c1-420 >> gpurun_out/r04_fuzzipm_ac.log
done
cat gpurun_out/r04_fuzzipm_ac.log
python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "python_surface" 2>&1 | tail -4
