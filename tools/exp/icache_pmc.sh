set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export COPRA_NO_BUILD=1
O=gpurun_out
rm -rf $O/c5_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/c5_ic -- python3 tools/try_config5.py 16384 0 > $O/c5_ic.log 2>&1 || (tail -5 $O/c5_ic.log; rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch" | head -20)
python tools/pmc_summary.py $O/c5_ic > $O/c5_ic_summary.json
