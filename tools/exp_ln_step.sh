#!/bin/bash
# the MAE step with the row LayerNorm kernels (S2K_LN_ROWS=1: forward, where the launcher's rule sends it) vs the tile kernels only (0): alternating
# runs on one box, tuning build on both sides
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
for i in 1 2 3; do
  for f in ${MODES:-1 0}; do
    echo -n "S2K_LN_ROWS=$f: "
    S2K_LIB=$T S2K_TUNING=1 S2K_LN_ROWS=$f timeout -k 10 200 python tools/bench_prithvi.py mae --batch 64 --steps 10 --warmup 3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f samples/s  %.3f ms' % (d['samples_per_s'], d['ms_per_step']))"
  done
done
