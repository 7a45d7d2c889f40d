#!/bin/bash
# A/B of the thin (M <= 32) 3x3 producer / consumer conv against the generic kernel (tuning build)
export S2K_LIB=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so S2K_TUNING=1
for shape in "32 32 256 0" "32 32 256 3" "24 64 128 0" "32 48 256 0"; do
  set -- $shape
  for cfg in "0 4" "1 4" "1 8"; do
    set -- $shape; t=${cfg% *}; k=${cfg#* }
    echo -n "THIN=$t KCH=$k "; S2K_CONV_PC_THIN=$t S2K_CONV_PC_THIN_KCH=$k python tools/bench_op.py conv3 --M $1 --C $2 --H $3 --pro $4 --iters 20 2>/dev/null | grep "^conv3"
  done
done
