#!/bin/bash
# usage (on the GPU box): tools/profile_policy.sh <tag> c3|c5
# Writes gpurun_out/profile_<tag>_<cfg>/: kernel-trace stats of `bench.py --config <cfg>` (whole generations / episodes through the
# product's drivers) and PMC passes of the fused policy step kernel: HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes), L2 hits / misses, SQ instruction mix and wait buckets.  One --pmc pass per counter set, never
# combined with tracing.
set -e
TAG=$1
CFG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "c3" ]; then
  BENCH="python3 $GRAFT_REPO_ROOT/bench.py --config c3 --generations 3"
else
  BENCH="python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 2000"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
out, cfg = "$OUT", "$CFG"
lines = []
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats) of bench.py --config %s ==" % cfg)
    lines += [l.rstrip() for l in open(f)]
bench = None
try:
    bench = json.loads([l for l in open(out + "/stats_bench.json") if l.startswith("{")][0])
    lines.append("== bench line of the stats pass ==")
    lines.append(json.dumps(bench))
except Exception as e:
    lines.append("bench line unreadable: %s" % e)
# the fused policy step kernels: the cooperative kernel (full population, long lists) and the tail kernel (short lists)
def is_policy_kernel(name):
    return any(k in name for k in ("okStepCoopKernel<1", "okStepCoopKernel<2", "okStepTailKernel<1", "okStepTailKernel<2"))
# counters: sums over ALL dispatches of the policy step kernels in the run (launches differ in length and in listed agents)
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if is_policy_kernel(r["Kernel_Name"]):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
lines.append("== PMC, fused policy step kernels (cooperative + tail), SUM over the run's dispatches (warm-up generation / episode included) ==")
for k in sorted(tot):
    lines.append("%-26s %20.0f   (%d dispatches)" % (k, tot[k], n[k]))
live = float(bench["rank0_live_agent_steps_with_warmup"]) if bench else None
kern_ns = None
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if is_policy_kernel(r["Name"]):
            kern_ns = (kern_ns or 0.0) + float(r["TotalDurationNs"])
if live and kern_ns:
    lines.append("the run (warm-up included): %.4e live agent-steps, policy step kernel busy %.3f ms -> %.3e live agent-steps/s kernel-only" % (live, kern_ns * 1e-6, live / (kern_ns * 1e-9)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    lines.append("HBM-side traffic of the run's policy launches: FETCH_SIZE %.1f MB raw (x2 = %.1f MB with the gfx950 half-count correction), WRITE_SIZE %.1f MB" % (tot["FETCH_SIZE"] / 1024, 2 * tot["FETCH_SIZE"] / 1024, tot["WRITE_SIZE"] / 1024))
    if live and kern_ns:
        b = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
        lines.append("  -> %.1f B per live agent-step beyond L2 (fetch x2 + write); %.1f GB/s over the kernel's busy time = %.4f of the 8 TB/s HBM peak" % (b / live, b / (kern_ns * 1e-9) / 1e9, b / (kern_ns * 1e-9) / 8e12))
if "TCC_HIT_sum" in tot:
    lines.append("L2: hits %.0f misses %.0f -> hit rate %.3f" % (tot["TCC_HIT_sum"], tot["TCC_MISS_sum"], tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])))
if "SQ_INSTS_VALU" in tot:
    wc = tot["SQ_WAVE_CYCLES"]
    lines.append("wave time: active %.3f, waiting on memory counters %.3f, issue stall %.3f ; VALU lane utilisation %.3f ; LDS bank-conflict share of LDS cycles %.3f" % (
        tot["SQ_ACTIVE_INST_ANY"] / wc, tot["SQ_WAIT_ANY"] / wc, tot["SQ_WAIT_INST_ANY"] / wc,
        tot["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot["SQ_ACTIVE_INST_VALU"]), tot["SQ_LDS_BANK_CONFLICT"] / max(tot.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
    lines.append("instructions: VALU %.3e SALU %.3e LDS %.3e VMEM %.3e (fp64 VALU: %.3e)" % (tot["SQ_INSTS_VALU"], tot["SQ_INSTS_SALU"], tot["SQ_INSTS_LDS"], tot["SQ_INSTS_VMEM"],
                 tot.get("SQ_INSTS_VALU_FMA_F64", 0) + tot.get("SQ_INSTS_VALU_MUL_F64", 0) + tot.get("SQ_INSTS_VALU_ADD_F64", 0)))
open(out + "/SUMMARY.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
