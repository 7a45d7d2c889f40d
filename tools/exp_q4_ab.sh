#!/bin/bash
# rocprofv3 kernel durations: the quad-layout 1x1 kernel (csrc/conv_q4.hip, bench_op --q4) against what the launcher picks without the
# quad weight copy (generic / producer-consumer / LDS-DMA ring), shape by shape
L=$PWD/sentinel2-landcover-classification_amd/libs2k.so
IFS=';' read -ra LIST <<< "${SHAPES:-1056 176 16;240 40 64;304 1824 8;768 128 16;384 64 32;128 768 16;40 240 64;144 24 128;512 3072 8;64 384 32;176 1056 16}"
for sh in "${LIST[@]}"; do
  set -- $sh
  for q in "--q4" ""; do
    timeout -k 5 100 bash tools/exp_prof_op.sh "conv1 --M $1 --C $2 --H $3 --scratch --iters 5 $EXTRA $q" q${q}_$1_$2 S2K_LIB=$L < /dev/null | grep -v "at::native\|rocclr\|^conv1" | tr '\n' ' '; echo
  done
done
