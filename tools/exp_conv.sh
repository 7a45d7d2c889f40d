set -e
cd $GRAFT_REPO_ROOT
export S2K_LIB=$GRAFT_REPO_ROOT/sentinel2-landcover-classification_amd/libs2k_tuning.so
for k3 in 4 8; do
echo "KCH3=$k3"
for shape in "--M 128 --C 128 --H 64" "--M 512 --C 512 --H 16" "--M 64 --C 64 --H 128"; do
S2K_CONV_PC_KCH3=$k3 python tools/bench_op.py conv3 $shape --pro 3 --iters 50
done
done
echo old; S2K_CONV_PC=0 python tools/bench_op.py conv3 --M 128 --C 128 --H 64 --pro 3 --iters 50
for k1 in 32 64 0; do
echo "KCH1=$k1 (0 = generic)"
for shape in "--M 512 --C 512 --H 32" "--M 3072 --C 768 --H 1 --N 200 --B 64" "--M 2048 --C 512 --H 1 --N 200 --B 64" "--M 512 --C 2048 --H 1 --N 200 --B 64"; do
if [ $k1 = 0 ]; then S2K_CONV_PC=2 python tools/bench_op.py conv1 $shape --iters 50; else S2K_CONV_PC_KCH1=$k1 python tools/bench_op.py conv1 $shape --iters 50; fi
done
done
