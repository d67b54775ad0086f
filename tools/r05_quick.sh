#!/bin/bash
# Round-5 quick look on the GPU box: headline kernel stats + tier phases + the parity tests of the headline pair.
#   gpurun -- 'bash tools/r05_quick.sh <tag>'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-q}
O=gpurun_out/r05_$T
mkdir -p $O
export COPRA_NO_BUILD=1
BENCH="python3 bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH --steps 20 --warmup 3 > $O/run.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
$BENCH --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python tools/exp/lane_tier1_phases.py 2>&1 | grep -v amdgpu.ids > $O/lane_tier1_phases.txt
head -4 $O/kernel_stats.csv | cut -c1-200
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench', d['value'], d['ms_per_step'], d.get('extra',{}).get('verification', d.get('verification')))"
cat $O/lane_tier1_phases.txt
if [ "$2" != "notest" ]; then
python -m pytest tests/test_gpu_parity.py -x -q -k "lane or ric or headline or handover or horizon or config3 or config4 or shared or reference" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
fi
