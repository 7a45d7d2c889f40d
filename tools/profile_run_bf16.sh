#!/bin/bash
# bf16 parts of tools/profile_run.sh only
tag=r03
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  out=$root/gpurun_out/prof_${tag}_$name
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- "$@" > $out/run.log 2>&1 || tail -5 $out/run.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/summarize_rocprof.py $f $root/gpurun_out/${tag}_${name}_kernel_stats.md "$tag $name: $*"
  grep -h '^{' $out/run.log | tail -1 > $root/gpurun_out/${tag}_${name}_bench.json.log
}
run unet_bf16 python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile --precision bf16-mixed
f=$(find $root/gpurun_out/prof_${tag}_unet_bf16 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $root/tools/stream_split.py $f --steps 8 --per-step 2 > $root/gpurun_out/${tag}_unet_bf16_streams.txt 2>&1
run mae_bf16 python3 $root/tools/bench_prithvi.py mae --batch 64 --steps 5 --warmup 2 --precision bf16-mixed
find $root/gpurun_out -path "*prof_${tag}_*" -name "*kernel_trace.csv" -delete
