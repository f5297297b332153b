#!/bin/bash
# usage (on the GPU box): tools/profile_run.sh <tag>
# Writes gpurun_out/profile_<tag>/: kernel-trace stats of the headline bench command, HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes), SQ instruction mix.
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 100 --steps-per-launch 100 --repeats 3 --no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.log
# the command the driver runs (one 20-step launch per timed region, 30 regions)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_k20 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only > $OUT/stats_k20_bench.json 2> $OUT/stats_k20.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
lines = []
avg_ns = None
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    lines += [l.rstrip() for l in open(f)]
    for r in csv.DictReader(open(f)):
        if "okStepCoopKernel" in r["Name"]:
            avg_ns = float(r["AverageNs"])
for f in glob.glob(out + "/stats_k20/*/*kernel_stats.csv"):
    lines.append("== kernel stats of bench.py --steps 20 --warmup 5 (the command the driver runs; 20 steps per launch) ==")
    lines += [l.rstrip() for l in open(f)]
try:
    lines.append("== bench line of that pass ==")
    lines.append(open(out + "/stats_k20_bench.json").read().strip())
except Exception as e:
    lines.append(str(e))
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "okStep" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 200000:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
lines.append("== PMC, okStep* kernel, average per dispatch (each dispatch = 100 Environment steps x 4096 agents) ==")
for k in sorted(agg):
    lines.append("%-26s %18.1f   (%d dispatches)" % (k, agg[k] / n[k], n[k]))
ws = 4096 * 100.0  # wave-steps per dispatch (one wave per agent at 64 rays, 100 steps per launch)
q = lambda k: agg[k] / max(n[k], 1)
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = q("FETCH_SIZE"); w = q("WRITE_SIZE")
    lines.append("HBM traffic per dispatch: FETCH_SIZE %.1f KB (raw; x2 = %.1f KB with the gfx950 half-count correction), WRITE_SIZE %.1f KB" % (f, 2 * f, w))
    lines.append("  -> per agent-step: fetch %.2f B (raw) / %.2f B (x2), write %.2f B ; algorithmic 354 B" % (f * 1024 / ws, 2 * f * 1024 / ws, w * 1024 / ws))
summary = {}
if "SQ_INSTS_VALU" in agg:
    wave_cyc = 4 * q("SQ_WAVE_CYCLES") / ws
    lines.append("per wave-step: VALU %.0f  SALU %.0f  LDS %.0f  VMEM %.1f  branches %.0f instructions" % tuple(q(k) / ws for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH")))
    lines.append("per wave-step cycles (quad-cycle counters x4): wave %.0f = active %.0f + waiting on memory counters %.0f + issue stall %.0f" % (wave_cyc, 4 * q("SQ_ACTIVE_INST_ANY") / ws, 4 * q("SQ_WAIT_ANY") / ws, 4 * q("SQ_WAIT_INST_ANY") / ws))
    lane_util = q("SQ_THREAD_CYCLES_VALU") / (64.0 * q("SQ_ACTIVE_INST_VALU")) if agg.get("SQ_ACTIVE_INST_VALU") else None
    lines.append("VALU lane utilisation (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU): %s" % lane_util)
    kern_cyc = q("GRBM_GUI_ACTIVE") / 8.0 if "GRBM_GUI_ACTIVE" in agg else None
    if kern_cyc:
        lines.append("GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles per dispatch; SIMD VALU issue share at 2 cycles per wave64 instruction, 4 waves per SIMD: %.3f" % (kern_cyc, q("SQ_INSTS_VALU") / 1024.0 * 2.0 / kern_cyc))
    summary = {"valu": q("SQ_INSTS_VALU") / ws, "salu": q("SQ_INSTS_SALU") / ws, "lds": q("SQ_INSTS_LDS") / ws,
               "active_frac": q("SQ_ACTIVE_INST_ANY") / q("SQ_WAVE_CYCLES"), "wait_frac": q("SQ_WAIT_ANY") / q("SQ_WAVE_CYCLES"),
               "issue_stall_frac": q("SQ_WAIT_INST_ANY") / q("SQ_WAVE_CYCLES"), "lane_util": lane_util}
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg and summary:
    import hashlib
    hh = hashlib.sha256()
    import sys
    sys.path.insert(0, "$GRAFT_REPO_ROOT")
    from bench import KERNEL_SOURCES  # the one list of files the hash covers
    for rel in KERNEL_SOURCES:
        hh.update(open("$GRAFT_REPO_ROOT/" + rel, "rb").read())
    tj = {"kernel_source_sha256": hh.hexdigest(),  # bench.py reports these counters only while the kernel's sources still hash to this
          "source": "profiles/%s_SUMMARY.txt (rocprofv3 --pmc passes of bench.py --steps 1000 --steps-per-launch 100 --headline-only)" % "$TAG",
          "kernel": "okStepCoopKernel", "agents": 4096, "rays": 64, "track": "Silverstone",
          "hbm_bytes_per_agent_step": (2 * q("FETCH_SIZE") + q("WRITE_SIZE")) * 1024 / ws,
          "fetch_size_kb_raw_per_100_step_launch": q("FETCH_SIZE"), "write_size_kb_per_100_step_launch": q("WRITE_SIZE"),
          "clock_ghz": (q("GRBM_GUI_ACTIVE") / 8.0) / avg_ns if ("GRBM_GUI_ACTIVE" in agg and avg_ns) else 2.4,
          "per_wave_step": summary,
          "lds_bank_conflict_frac": (q("SQ_LDS_BANK_CONFLICT") / q("SQ_LDS_IDX_ACTIVE")) if agg.get("SQ_LDS_IDX_ACTIVE") else None,
          "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per 100-step launch / 409600 agent-steps: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B). Far below the algorithmic 354 B because the per-ray outputs are overwritten every step and stay in L2: they reach HBM once per launch."}
    json.dump(tj, open(out + "/hbm_traffic.json", "w"), indent=1)
try:
    lines.append("== bench line of the stats pass ==")
    lines.append(open(out + "/stats_bench.json").read().strip())
except Exception as e:
    lines.append(str(e))
open(out + "/SUMMARY.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
