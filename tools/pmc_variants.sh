#!/bin/bash
# usage (GPU box): tools/pmc_variants.sh <variant> [<variant> ...]   -- SQ counters of the step kernel per library variant
# (tools/build_variant.sh), 100-step launches of the C2 workload; prints per wave-step averages.
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$V
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
    i=$((i+1))
    OKENV_VARIANT=$V rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/variant_run.py > $OUT/pass$i.log 2>&1 || echo "pass $i of $V failed"
  done
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'okStepCoop' in r['Kernel_Name'] and int(r['Grid_Size'])>=200000:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
ws = 4096*100.0
q = lambda k: agg[k]/max(n[k],1)
print("== $V (per wave-step; *_CYCLES/ACTIVE/WAIT in cycles = quad-cycles x4)")
print("  insts: VALU %.0f SALU %.0f LDS %.0f VMEM %.1f SMEM %.2f BRANCH %.0f" % tuple(q(k)/ws for k in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM","SQ_INSTS_SMEM","SQ_INSTS_BRANCH")))
print("  wave cycles %.0f = active %.0f + wait_any %.0f + wait_inst %.0f ; active VALU %.0f SCA %.0f LDS %.0f ; wait_inst_lds %.0f" % tuple(4*q(k)/ws for k in ("SQ_WAVE_CYCLES","SQ_ACTIVE_INST_ANY","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS")))
print("  dispatches:", dict(n))
PY
done
