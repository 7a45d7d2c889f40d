#!/bin/bash
# channel LayerNorm forward at the MAE sizes: tile kernel (S2K_LN_ROWS=0) vs row kernel (4 = wherever it supports the shape), and the row
# kernel with one part switched off (tuning build, S2K_LN_DBG: 1 = no all-channel sums, 2 = no normalising pass).  lnbwd: the tile kernel
# (the row form of the backward was removed: see csrc/vit.hip ln_rows_geometry)
T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
for sh in "768 52" "512 200" "768 196"; do
  set -- $sh
  for what in lnfwd "lnbwd --beta --bias"; do
    for env in ${ENVS:-"S2K_LN_ROWS=0" "S2K_LN_ROWS=4" "S2K_LN_ROWS=4 S2K_LN_DBG=3"}; do
      echo -n "$what C=$1 N=$2 $env: "
      env S2K_LIB=$T S2K_TUNING=1 $env timeout -k 5 60 python tools/bench_op.py $what --B 64 --C $1 --N $2 --rep 20 --iters 20 2>&1 | grep "TF/s" | sed "s/ TF.s/ TB.s/"
    done
  done
done
