# kernel durations and SQ counters of the one-instance-per-lane pass (GPU box):  gpurun -- 'bash tools/exp/lane_pmc.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export COPRA_NO_BUILD=1
O=gpurun_out
rm -rf $O/lane_*
BENCH="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lane_stats -- $BENCH > $O/lane_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/lane_sq -- $BENCH >> $O/lane_run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/lane_sq2 -- $BENCH >> $O/lane_run.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/lane_sq3 -- $BENCH >> $O/lane_run.log 2>&1 || echo "(third counter set not available)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lane_fetch -- $BENCH >> $O/lane_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lane_write -- $BENCH >> $O/lane_run.log 2>&1
python tools/pmc_summary.py $O/lane_stats $O/lane_sq $O/lane_sq2 $O/lane_sq3 $O/lane_fetch $O/lane_write > $O/lane_pmc_summary.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/lane_pmc_summary.json"))
for k in d.get("kernel_stats", []):
    print(k["Name"][:60], k["Calls"], k["AverageNs"])
for kern, c in d.get("counters", {}).items():
    if "lane" in kern or "fused_ric" in kern:
        print(kern[:50], {n: round(v["mean_per_launch"]) for n, v in c.items()})
PY
