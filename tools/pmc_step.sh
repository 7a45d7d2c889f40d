#!/bin/bash
# SQ instruction mix of every kernel of one U-Net training step (run on the GPU box): tools/pmc_step.sh <tag>
tag=$1
root=$(pwd)
out=$root/gpurun_out/pmcstep_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  -d $out/p -o p --output-format csv -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $out/run.log 2>&1 || tail -5 $out/run.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void s2k::", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -(kv[1]["SQ_INSTS_VALU"] * 4 + kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]))
print("%-70s %6s %10s %10s %8s %8s %8s" % ("kernel (summed over the run)", "n", "VALU*4 Mcy", "MFMA Mcy", "VALU/MFMA", "wait%", "active%"))
for k, d in rows[:40]:
    wc = d["SQ_WAVE_CYCLES"] or 1
    print("%-70s %6d %10.1f %10.1f %8.1f %8.1f %8.1f" % (k, n[k], d["SQ_INSTS_VALU"] * 4 / 1e6, d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1e6,
          d["SQ_INSTS_VALU"] / max(d["SQ_INSTS_MFMA"], 1), 100 * d["SQ_WAIT_INST_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc))
PY
