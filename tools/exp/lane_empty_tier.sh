# durations of the three launches when EVERY instance ends in the lane pass (the first tier is 65536 workgroups that leave at once)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export COPRA_NO_BUILD=1
O=gpurun_out
rm -rf $O/lane_empty
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lane_empty -- python3 tools/exp/lane_pass_variants.py > $O/lane_empty.log 2>&1
find $O/lane_empty -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-150 | head -8
