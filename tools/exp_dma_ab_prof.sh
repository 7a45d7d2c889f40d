#!/bin/bash
# rocprofv3 kernel durations, LDS-DMA ring kernel vs the kernels it replaces (tuning build: S2K_CONV_DMA=1 / 0), one line per shape
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
IFS=';' read -ra LIST <<< "${SHAPES:-512 3072 8;512 2048 8;1056 304 8;384 128 16;768 176 16;176 1056 16;64 384 32;240 64 32}"
for sh in "${LIST[@]}"; do
  set -- $sh
  for dma in 1 0; do
    timeout -k 5 100 bash tools/exp_prof_op.sh "conv1 --M $1 --C $2 --H $3 --scratch --iters 5 $EXTRA" ab${dma}_$1_$2 S2K_LIB=$T S2K_TUNING=1 S2K_CONV_DMA=$dma < /dev/null | grep -v "at::native\|rocclr\|^conv1" | tr '\n' ' '; echo
  done
done
