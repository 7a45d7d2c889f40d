#!/bin/bash
# write-heavy bf16 1x1 convs: where does the time go (tuning build: S2K_CV_EXP 1 = no epilogue, 2 = no MFMA loop)
export S2K_LIB=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so S2K_TUNING=1
for shape in "240 40 64" "144 24 128" "40 240 64"; do
  set -- $shape
  for e in 0 1 2 3; do
    echo -n "exp=$e "; S2K_CV_EXP=$e python tools/bench_op.py conv1 --bf16 --M $1 --C $2 --H $3 --nostats --iters 20 2>/dev/null | grep "^conv1"
  done
done
