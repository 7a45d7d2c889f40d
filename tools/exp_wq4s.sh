T=$PWD/sentinel2-landcover-classification_amd/libs2k_tuning.so
for sh in "256 256 32" "128 128 64" "512 512 16"; do
  set -- $sh
  for env in "S2K_WG_Q4=3" "S2K_WG_Q4=3 S2K_WG_EXP=1" "S2K_WG_Q4=3 S2K_WG_EXP=4" "S2K_WG_Q4=3 S2K_WG_EXP=2" "S2K_WG_Q4=1"; do
    echo -n "M=$1 C=$2 H=$3 $env: "
    env S2K_LIB=$T S2K_TUNING=1 $env timeout -k 5 60 python tools/bench_op.py wgrad3 --B 32 --M $1 --C $2 --H $3 --rep 10 --iters 10 2>&1 | grep "TF/s" | cut -c1-110
  done
done
