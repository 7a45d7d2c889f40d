#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of one conv stage: bash tools/exp_prof_op.sh "<bench_op args>" tag [env...]
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
args="$1"; tag="$2"; shift 2
for kv in "$@"; do export "$kv"; done
rm -rf $ROOT/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -o p -- python3 $ROOT/tools/bench_op.py $args > $ROOT/gpurun_out/prof_$tag.log 2>&1
f=$(find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag: $args"; grep "^conv1\|^wgrad" $ROOT/gpurun_out/prof_$tag.log
if [ -z "$f" ]; then echo "   (no kernel_stats.csv)"; find $ROOT/gpurun_out/prof_$tag | head -5; rm -rf $ROOT/gpurun_out/prof_$tag; exit 0; fi
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
rm -rf $ROOT/gpurun_out/prof_$tag
