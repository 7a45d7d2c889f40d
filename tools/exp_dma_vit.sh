#!/bin/bash
# ViT Linears (feature-major tokens, H = 1): LDS-DMA ring kernel (FLAG_DMA) vs the producer / consumer kernel, rocprofv3 kernel durations
L=$PWD/sentinel2-landcover-classification_amd/libs2k.so
for sh in "2304 768 52" "768 768 52" "3072 768 52" "768 3072 52" "1536 512 200" "512 512 200" "2048 512 200" "512 2048 200"; do
  set -- $sh
  for dma in ${MODES:-"--dma" ""}; do
    timeout -k 5 100 bash tools/exp_prof_op.sh "conv1 --B 64 --M $1 --C $2 --N $3 --nostats --bias --iters 5 $dma" vit${dma}_$1_$2 S2K_LIB=$L < /dev/null | grep -v "at::native\|rocclr\|^conv1" | tr '\n' ' '; echo
  done
done
