#!/bin/bash
# Same-box A/B of two builds (box-to-box spread of the pool is +-2 %, more than most of what is worth measuring): the working tree against an
# older commit checked out AND BUILT next to it --  git worktree add ab_old <commit>; (cd ab_old && python -c 'import __graft_entry__ as g; g.build()')
# -- three alternating runs of the headline under rocprofv3 --kernel-trace --stats; the kernel_stats.csv files land in gpurun_out/ab2_<i>_{n,ab_old}.
# (How the 3 % that the stage-varying affine term cost the headline's tier in a shared build was found: 281 -> 290 us, profiles/r03/README.md.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export COPRA_NO_BUILD=1
for i in 1 2 3; do
  for d in . ab_old; do
    (cd $d && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab2_${i}_$(basename $d | tr . n) -- python3 bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 2 2>/dev/null | cut -c90-130)
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/ab2_${i}_$(basename $d | tr . n) -name "*kernel_stats.csv")
    echo "$d: $(sed -n 2,3p $f | cut -d, -f1,4 | tr '\n' ' ')"
  done
done
