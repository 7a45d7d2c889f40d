#!/bin/bash
# wgrad_q4_kernel on the MAE / U-Net 1x1 shapes: whole kernel, consumers alone (S2K_WG_EXP=1: the first tile staged only), producers alone
# (4: no MFMA loop), and wgrad_pc_kernel (S2K_WG_Q4=0) - tuning build, HIP-event times of back-to-back launches
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
IFS=';' read -ra LIST <<< "${SHAPES:-64 3072 768 52;64 768 3072 52;64 2048 512 200;64 768 768 52;32 1056 176 256;32 240 40 4096}"
for sh in "${LIST[@]}"; do
  set -- $sh
  for env in "S2K_WG_Q4=1" "S2K_WG_Q4=1 S2K_WG_EXP=1" "S2K_WG_Q4=1 S2K_WG_EXP=4" "S2K_WG_Q4=0"; do
    echo -n "B=$1 M=$2 C=$3 N=$4 $env: "
    env S2K_LIB=$T S2K_TUNING=1 $env timeout -k 5 60 python tools/bench_op.py wgrad1 --B $1 --M $2 --C $3 --N $4 --rep 10 --iters 10 2>&1 | grep "TF/s" | cut -c1-120
  done
done
