#!/bin/bash
# rocprofv3 kernel stats of the bench commands (run on the GPU box): tools/profile_run.sh <tag> [bf16]
# writes gpurun_out/prof_<tag>_*/ + condensed summaries gpurun_out/<tag>_*.md; "bf16": only the bf16-mixed runs
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {  # name, then the program and its args
  name=$1; shift
  out=$root/gpurun_out/prof_${tag}_$name
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- "$@" > $out/run.log 2>&1 || tail -5 $out/run.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/summarize_rocprof.py $f $root/gpurun_out/${tag}_${name}_kernel_stats.md "$tag $name: $*"
  grep -h '^{' $out/run.log | tail -1 > $root/gpurun_out/${tag}_${name}_bench.json.log
}
streams() {  # per-queue view of a run's kernel trace (main / side stream overlap, the main queue's waits)
  f=$(find $root/gpurun_out/prof_${tag}_$1 -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/stream_split.py $f --steps 8 --per-step 2 --skip-last $2 > $root/gpurun_out/${tag}_$1_streams.txt 2>&1
}
only=$2
if [ "$only" != "bf16" ]; then
# the headline step, f32 (the bench's own profile leg runs after the timed steps: skipped by --skip-last)
run unet python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi --no-bf16
f=$(find $root/gpurun_out/prof_${tag}_unet -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $root/tools/busy_union.py $f --steps 10 --skip-last 11 --per-step 2 > $root/gpurun_out/${tag}_unet_busy.txt
fi
run unet_bf16 python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile --precision bf16-mixed
streams unet_bf16 0
[ "$only" != "bf16" ] && run mae python3 $root/tools/bench_prithvi.py mae --batch 64 --steps 5 --warmup 2
run mae_bf16 python3 $root/tools/bench_prithvi.py mae --batch 64 --steps 5 --warmup 2 --precision bf16-mixed
if [ "$only" != "bf16" ]; then
run seg python3 $root/tools/bench_prithvi.py seg --batch 16 --steps 5 --warmup 2
run seg_unfrozen python3 $root/tools/bench_prithvi.py seg --batch 16 --steps 5 --warmup 2 --unfrozen
fi
# keep the merge small: the raw traces stay on the box
find $root/gpurun_out -path "*prof_${tag}_*" -name "*kernel_trace.csv" -delete
