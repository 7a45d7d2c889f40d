#!/bin/bash
# deep 1x1 bf16 convs: 64- vs 128-channel chunks (tuning build: S2K_B16_DEEP_MIN = smallest Ctot that takes 128)
export S2K_LIB=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so S2K_TUNING=1
for shape in "176 1056 16 2" "304 1824 8 2" "512 3072 8 2" "1056 176 16 0" "3072 512 8 0" "2048 512 8 0" "128 768 16 0" "512 512 16 0" "64 384 32 2" "768 3072 56 0" "3072 768 56 0"; do
  set -- $shape
  for dm in 100000 512; do
    echo -n "deep_min=$dm "; S2K_B16_DEEP_MIN=$dm python tools/bench_op.py conv1 --bf16 --M $1 --C $2 --H $3 --pro $4 --scratch --iters 30 2>/dev/null | grep "^conv1"
  done
done
