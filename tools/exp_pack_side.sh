#!/bin/bash
# WEIGHT_PACK on the side stream during the forward: S2K_PACK_SIDE = 0 in front of the backward, 1 (default) during the forward:
# alternating runs on one box, shipped library (planner switch: S2K_TUNING=1 makes the planner honour it)
for i in 1 2 3; do
  for m in ${MODES:-1 0}; do
    echo -n "S2K_PACK_SIDE=$m: unet "
    S2K_TUNING=1 S2K_PACK_SIDE=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f tiles/s  %.3f ms' % (d['value'], d['ms_per_step']), end='   ')"
    echo -n "mae "
    S2K_TUNING=1 S2K_PACK_SIDE=$m timeout -k 10 200 python tools/bench_prithvi.py mae --batch 64 --steps 10 --warmup 3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f samples/s  %.3f ms' % (d['samples_per_s'], d['ms_per_step']))"
  done
done
