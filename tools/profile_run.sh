#!/bin/bash
# rocprofv3 kernel stats of the bench commands (run on the GPU box): tools/profile_run.sh <tag>
# writes gpurun_out/prof_<tag>_{unet,mae,seg}/ + condensed summaries gpurun_out/<tag>_*.md
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
run unet python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi
f=$(find $root/gpurun_out/prof_${tag}_unet -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $root/tools/busy_union.py $f --steps 10 --skip-last 11 --per-step 2 > $root/gpurun_out/${tag}_unet_busy.txt
run mae python3 $root/tools/bench_prithvi.py mae --batch 64 --steps 5 --warmup 2
run seg python3 $root/tools/bench_prithvi.py seg --batch 16 --steps 5 --warmup 2
