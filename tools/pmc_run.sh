#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench args...]   -- collects SQ counter passes for the step kernel into gpurun_out/pmc_<tag>/
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 100 --steps-per-launch 100 --no-cpu-baseline --headline-only "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'okStep' in r['Kernel_Name'] and int(r['Grid_Size'])>=200000:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(agg): print("%-28s %16.0f  (avg per dispatch over %d dispatches)"%(k,agg[k]/n[k],n[k]))
PY
