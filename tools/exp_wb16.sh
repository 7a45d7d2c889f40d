#!/bin/bash
# thin 1x1 bf16 weight gradients: pixel splits and the atomic combine (tuning build)
export S2K_LIB=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so S2K_TUNING=1
for shape in "24 24 128" "32 32 128" "64 64 128"; do
  set -- $shape
  for slots in 512 256 128 64; do
    for e in 0 2; do
      echo -n "slots=$slots exp=$e "; S2K_WB16_SLOTS=$slots S2K_WG_EXP=$e python tools/bench_op.py wgrad1 --bf16 --M $1 --C $2 --H $3 --iters 20 2>/dev/null | grep "^wgrad1"
    done
  done
done
