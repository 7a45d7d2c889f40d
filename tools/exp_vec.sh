#!/bin/bash
# vector (HW % 4 == 0) vs scalar pixel staging for the ViT linear shapes
run() { timeout -k 10 120 python tools/bench_op.py "$@" --nostats 2>&1 | tail -1; }
for cfg in "3072 768" "768 3072" "2304 768" "768 768"; do set -- $cfg
  echo "enc $1 $2: N=50 / N=52 / N=52 novec"; run conv1 --B 64 --M $1 --C $2 --N 50; run conv1 --B 64 --M $1 --C $2 --N 52
  run wgrad1 --B 64 --M $1 --C $2 --N 50; run wgrad1 --B 64 --M $1 --C $2 --N 52
done
for cfg in "2048 512" "512 2048" "1536 512" "512 512"; do set -- $cfg
  echo "dec $1 $2: N=197 / N=200"; run conv1 --B 64 --M $1 --C $2 --N 197; run conv1 --B 64 --M $1 --C $2 --N 200
  run wgrad1 --B 64 --M $1 --C $2 --N 197; run wgrad1 --B 64 --M $1 --C $2 --N 200
done
