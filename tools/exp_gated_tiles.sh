#!/bin/bash
# SiLU + SE-gated project convs on the generic kernel: which tile (tuning build: S2K_PIX_FORCE 0 = the launcher's choice, 1 = 64x64 / 64-ch
# chunks, 2 = 128x128 / 64, 3 = 64x256 / 16, 5 = 128x128 / 16); rocprofv3 kernel durations
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
IFS=';' read -ra LIST <<< "${SHAPES:-304 1824 8;176 1056 16;128 768 16;512 3072 8;40 240 64;64 384 32}"
for sh in "${LIST[@]}"; do
  set -- $sh
  for f in ${FORCES:-0 1 2 3 5}; do
    timeout -k 5 100 bash tools/exp_prof_op.sh "conv1 --M $1 --C $2 --H $3 --pro 2 --gate --scratch --iters 5" g${f}_$1_$2 S2K_LIB=$T S2K_TUNING=1 S2K_PIX_FORCE=$f < /dev/null | grep -v "at::native\|rocclr\|^conv1" | tr '\n' ' '; echo
  done
done
