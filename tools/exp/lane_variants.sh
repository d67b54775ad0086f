#!/bin/bash
# What a phase of the one-instance-per-lane pass costs: one library per value of COPRA_LANE_EXP (lmpc_lane.hpp: experiment switches compiled
# out of the product build), the headline's kernel times of each.  Results of the variants are WRONG by construction: timing only.
#   build (anywhere hipcc is):  bash tools/exp/lane_variants.sh build "0 1 2 4 8 16 31"
#   measure (GPU box):          bash tools/exp/lane_variants.sh run   "0 1 2 4 8 16 31"
set -e
cd "$(dirname "$0")/../../copra_amd/csrc"
mkdir -p variants
SRCHASH=$(cat libcopra_hip.so.srchash)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed -DCOPRA_SRC_HASH=\"$SRCHASH\" $EXTRA" # (EXTRA: more -D switches, e.g. -DCOPRA_LANE_KB=1; TAG names the outputs)
if [ "$1" = build ]; then
  for v in $2; do
    ( /opt/rocm/bin/hipcc $FLAGS -DCOPRA_LANE_EXP=$v -c -o variants/copra_hip_exp$TAG$v.o copra_hip.hip && \
      /opt/rocm/bin/hipcc $FLAGS -shared -o variants/libcopra_hip_exp$TAG$v.so variants/copra_hip_exp$TAG$v.o build/copra_hip_ric.o build/copra_hip_setters.o \
         build/copra_hip_jit.o build/copra_hip_qp.o build/copra_hip_packed16.o build/copra_hip_packed32.o ) &
  done
  wait
  ls -la variants/*.so
else
  cd ../..
  for v in $2; do
    echo "== COPRA_LANE_EXP=$v $TAG"
    python - "$TAG$v" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from copra_amd import _capi
_capi.LIB_PATH = os.path.abspath("copra_amd/csrc/variants/libcopra_hip_exp%s.so" % sys.argv[1])
_capi.build_library = lambda force=False: False
from copra_amd import BatchLMPC, workloads
b = 65536
wl = workloads.com_preview(b)
eng = BatchLMPC(6, 3, wl["N"], b, wl["costs"], wl["cstrs"])
eng.set_system(wl["A"], wl["B"], wl["d"], wl["x0"])
for _ in range(4):
    eng.solve()
eng.enable_phase_profile(True)
ts = []
for _ in range(6):
    eng.solve(); eng.synchronize(); ts.append(eng.last_solve_seconds())
pr = eng.phase_profile()[: b // 64]
print("solve %.4f ms | pass per wave: staging %.0f sweep %.0f roll-out %.0f verdict %.0f total %.0f cycles | finished %s"
      % (1e3 * float(np.mean(ts[2:])), pr[:, 0].mean(), pr[:, 1].mean(), pr[:, 2].mean(), pr[:, 3].mean(), pr[:, 7].mean(), eng.lane_pass_info()))
if pr[:, 4].mean() > 0:  # (bit 64: the roll-out's stages in three parts)
    print("   roll-out: gains+controls %.0f | rows+bounds %.0f | dynamics %.0f | rest (store phase) %.0f"
          % (pr[:, 4].mean(), pr[:, 5].mean(), pr[:, 6].mean(), pr[:, 2].mean() - pr[:, 4:7].sum(axis=1).mean()))
PY
  done
fi
