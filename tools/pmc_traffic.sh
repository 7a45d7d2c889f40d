#!/bin/bash
# HBM traffic of the bench's kernels from the L2 memory-side counters (run on the GPU box):
#   tools/pmc_traffic.sh <tag>
# FETCH_SIZE and WRITE_SIZE are collected in separate --pmc passes (TCC slots), no tracing domains.
tag=$1
root=$(pwd)
out=$root/gpurun_out/traffic_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr -d $out/$ctr -o p --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $out/$ctr.log 2>&1 || tail -5 $out/$ctr.log
  # calibration on known byte counts: a float4 stream (torch copy) and the dword-per-lane conv / depthwise kernels
  rocprofv3 --pmc $ctr -d $out/cal_$ctr -o p --output-format csv -- python3 $root/tools/bench_op.py copy --C 144 --H 128 --iters 2 > $out/cal_$ctr.log 2>&1
  rocprofv3 --pmc $ctr -d $out/cal2_$ctr -o p --output-format csv -- python3 $root/tools/bench_op.py dwfwd --C 240 --H 64 --S 1 --pro 0 --nostats --iters 2 > $out/cal2_$ctr.log 2>&1
  rocprofv3 --pmc $ctr -d $out/cal3_$ctr -o p --output-format csv -- python3 $root/tools/bench_op.py conv3 --M 128 --C 128 --H 64 --iters 2 > $out/cal3_$ctr.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for kind, pat in (("bench", "$out/%s" % ctr), ("cal_copy", "$out/cal_%s" % ctr), ("cal_dw", "$out/cal2_%s" % ctr), ("cal_conv3", "$out/cal3_%s" % ctr)):
        acc = collections.defaultdict(float); n = collections.Counter()
        for f in glob.glob(pat + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != ctr: continue
                k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
                acc[k] += float(r["Counter_Value"]); n[k] += 1
        res.setdefault(kind, {})[ctr] = {k: {"sum": v, "dispatches": n[k], "avg": v / n[k]} for k, v in acc.items() if n[k]}
json.dump(res, open("$root/gpurun_out/traffic_$tag.json", "w"), indent=1)
for kind, d in res.items():
    for ctr, kk in d.items():
        top = sorted(kk.items(), key=lambda kv: -kv[1]["sum"])[:6]
        for k, v in top: print(kind, ctr, k, "avg %.1f  n %d" % (v["avg"], v["dispatches"]))
PY
