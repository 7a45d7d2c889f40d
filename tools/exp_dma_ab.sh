#!/bin/bash
# A/B of the LDS-DMA ring kernel (csrc/conv_dma.hip) against the kernels it replaces, shape by shape (tuning build: S2K_CONV_DMA)
export S2K_LIB=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so S2K_TUNING=1
SHAPES=${SHAPES:-"240 40 64;144 24 128;1056 176 16;768 128 16;384 64 32;40 240 64;128 768 16;304 1824 8;512 3072 8;64 384 32"}
IFS=';' read -ra LIST <<< "$SHAPES"
for shape in "${LIST[@]}"; do
  set -- $shape
  for dma in 1 0; do
    for ns in "" "--nostats"; do
      echo -n "dma=$dma stats=${ns:-yes} "; S2K_CONV_DMA=$dma python tools/bench_op.py conv1 --M $1 --C $2 --H $3 $ns --scratch --iters 10 $EXTRA 2>/dev/null | grep "^conv1\|per wave"
    done
  done
done
