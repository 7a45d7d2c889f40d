#!/bin/bash
# How much of a short-K 1x1 conv is its BatchNorm-statistics epilogue (f64 atomics per row and wave)?  with / without STATS
for shape in "240 40 64" "144 24 128" "1056 176 16" "768 128 16" "384 64 32" "40 240 64" "176 1056 16" "304 1824 8" "24 144 128"; do
  set -- $shape
  for ns in "" "--nostats"; do
    echo -n "stats=${ns:-yes} "; python tools/bench_op.py conv1 --M $1 --C $2 --H $3 $ns --scratch --iters 50 2>/dev/null | grep "^conv1"
  done
done
