#!/bin/bash
# tile configurations of the 1x1 conv on the HBM-bound MBConv shapes: S2K_PIX_FORCE 0 = dispatcher, 1 = 64x64, 2 = 128x128 K64,
# 3 = 64x256, 4 = 32x256, 5 = 128x128 K16
for cfg in "240 40 64" "40 240 64" "144 24 128" "24 144 128" "384 64 32" "64 384 32" "768 128 16" "128 768 16" "1056 176 16" "176 1056 16"; do set -- $cfg
  for f in 0 1 2 3 4 5; do
    echo -n "force=$f  "; S2K_PIX_FORCE=$f timeout -k 10 60 python tools/bench_op.py conv1 --M $1 --C $2 --H $3 --nostats --pro 2 2>&1 | tail -1
  done
done
