#!/bin/bash
# usage (on the GPU box): tools/profile_run.sh <tag>
# Writes gpurun_out/profile_<tag>/: kernel-trace stats of the headline bench command, HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes), SQ instruction mix.
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 100 --steps-per-launch 100 --no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
lines = []
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    lines += [l.rstrip() for l in open(f)]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "okStep" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 200000:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
lines.append("== PMC, okStep* kernel, average per dispatch (each dispatch = 100 Environment steps x 4096 agents) ==")
for k in sorted(agg):
    lines.append("%-26s %18.1f   (%d dispatches)" % (k, agg[k] / n[k], n[k]))
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = agg["FETCH_SIZE"] / n["FETCH_SIZE"]; w = agg["WRITE_SIZE"] / n["WRITE_SIZE"]
    lines.append("HBM traffic per dispatch: FETCH_SIZE %.1f KB (raw; x2 = %.1f KB if the gfx950 half-count applies), WRITE_SIZE %.1f KB" % (f, 2 * f, w))
    lines.append("  -> per agent-step: fetch %.1f B (raw) / %.1f B (x2), write %.1f B ; algorithmic 354 B" % (f * 1024 / 409600, 2 * f * 1024 / 409600, w * 1024 / 409600))
try:
    lines.append("== bench line of the stats pass ==")
    lines.append(open(out + "/stats_bench.json").read().strip())
except Exception as e:
    lines.append(str(e))
open(out + "/SUMMARY.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
