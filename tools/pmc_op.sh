#!/bin/bash
# PMC counters of one micro-benchmarked stage: tools/pmc_op.sh <tag> <bench_op args...>   (run on the GPU box)
# counters are collected in their own passes (no tracing domains), summed per kernel name
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $root/tools/bench_op.py "$@" --iters 3 > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "elementwise" in k or "distribution" in k: continue
        print(k)
        for c, v in d.items(): print(f"   {c:32s} {v / n[(k, c)]:16.0f}  (per dispatch, {n[(k, c)]} dispatches)")
PY
