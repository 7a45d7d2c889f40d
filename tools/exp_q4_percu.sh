#!/bin/bash
# write-heavy 1x1 layers (short reductions on large maps) on the quad kernel: one workgroup per CU vs two (128 x 128 tile, 16-channel
# stages in a ring of 4: 67 KB of LDS, 108 registers - two fit), rocprofv3 kernel durations; tuning build
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
IFS=';' read -ra LIST <<< "${SHAPES:-240 40 64;144 24 128;384 64 32;144 40 64;64 128 128}"
for sh in "${LIST[@]}"; do
  set -- $sh
  for env in "S2K_Q4_PER_CU=1" "S2K_Q4_TILE=3 S2K_Q4_KCH=16 S2K_Q4_PER_CU=1" "S2K_Q4_TILE=3 S2K_Q4_KCH=16 S2K_Q4_PER_CU=2" "S2K_Q4_TILE=4 S2K_Q4_KCH=16 S2K_Q4_PER_CU=2"; do
    echo -n "$env: "
    timeout -k 5 100 bash tools/exp_prof_op.sh "conv1 --M $1 --C $2 --H $3 --iters 5 --q4" qp_$1_$2 S2K_LIB=$T S2K_TUNING=1 $env < /dev/null | grep -v "at::native\|rocclr\|^conv1" | tr '\n' ' '; echo
  done
done
