#!/bin/bash
# where does the LDS-DMA ring kernel's time go: S2K_CV_EXP 1 = no epilogue, 2 = no MFMAs, 4 = no DMA, 8 = no global stores (tuning build; results are garbage)
export S2K_LIB=${S2K_LIB:-$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so} S2K_TUNING=1
SHAPES=${SHAPES:-"240 40 64;1056 176 16;304 1824 8"}
IFS=';' read -ra LIST <<< "$SHAPES"
for shape in "${LIST[@]}"; do
  set -- $shape
  for e in ${EXPS:-0 1 2 4 8 3 5 6 7}; do
    echo -n "exp=$e "; S2K_CV_EXP=$e python tools/bench_op.py conv1 --M $1 --C $2 --H $3 --nostats --scratch --iters 10 $EXTRA 2>/dev/null | grep "^conv1\|per wave"
  done
done
