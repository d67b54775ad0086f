#!/bin/bash
# A second, disjoint seed range of every random differential generator on the final build of round 5 (GPU box):
#   gpurun --timeout 3000 -- 'bash tools/exp/soak_r05.sh'        output: gpurun_out/soak_r05/*.txt  (copied to profiles/r05/wide_*.txt)
export COPRA_NO_BUILD=1
O=gpurun_out/soak_r05
mkdir -p $O
f() { grep -v amdgpu.ids; }
(timeout 900 python tests/fuzz/fuzz_vs_oracle.py 3000 6000 48 2>&1 | f | tail -60) > $O/wide_random_controllers_3000.txt
(timeout 900 python tests/fuzz/fuzz_integrators.py 150 450 2>&1 | f | grep "<<<<\|mismatching" | cut -c1-500) > $O/wide_integrator_shapes_150.txt
(timeout 600 python tests/fuzz/fuzz_modes.py 400 800 2>&1 | f | grep " <\|ERROR\|mismatching" | cut -c1-400) > $O/wide_modes_400.txt
(for sh in "12 6" "5 3" "7 2" "3 3" "6 1" "4 2"; do echo "== shape $sh"; timeout 600 python tests/fuzz/fuzz_interior_point.py 60 60 $sh 2>&1 | f | grep "certified\|<<<<\|mismatching" | cut -c1-420; done) > $O/wide_interior_point_60.txt
(timeout 600 python tests/fuzz/fuzz_shared_general_rows.py 300 300 2>&1 | f | tail -12) > $O/wide_shared_general_rows_300.txt
(timeout 600 python tests/fuzz/fuzz_shared_general_rows.py 360 120 32768 integrators 2>&1 | f | tail -8) > $O/wide_shared_integrators_32768_360.txt
(timeout 600 python tests/fuzz/fuzz_shared_general_rows.py 240 240 24576 integrators-refs 2>&1 | f | tail -8) > $O/wide_shared_integrators_refs_240.txt
(timeout 600 python tests/fuzz/fuzz_dense_qp.py 3000 1600 2>&1 | f | tail -14) > $O/wide_dense_qp_3000.txt
tail -n 2 $O/*.txt | cut -c1-300
