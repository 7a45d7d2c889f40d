#!/bin/bash
# Prithvi MAE step with fc2's data gradient multiplied by gelu'(fc1 output) in the CONV epilogue (FLAG_RES_GELU_GRAD) vs a separate
# ACT_BWD pass: alternating runs on one box
for i in 1 2 3; do
  for f in 1 0; do
    echo -n "fuse=$f: "
    S2LC_FUSE_GELU_GRAD=$f timeout -k 10 200 python tools/bench_prithvi.py ${WHAT:-mae --batch 64} --steps 10 --warmup 3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f samples/s  %.3f ms' % (d['samples_per_s'], d['ms_per_step']))"
  done
done
