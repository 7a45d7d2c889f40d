#!/bin/bash
# the headline step with the quad weight-gradient kernels: S2K_WG_Q4 = 0 none (wgrad_pc_kernel), 1 = 1x1 only,
# 5 = 1x1 and the SiLU + SE-gate operand (the shipped setting) - alternating runs on one box, tuning build
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
for i in 1 2 3; do
  for m in ${MODES:-0 1 5}; do
    echo -n "S2K_WG_Q4=$m: "
    S2K_LIB=$T S2K_TUNING=1 S2K_WG_Q4=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f tiles/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
