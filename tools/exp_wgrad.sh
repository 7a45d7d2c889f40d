set -e
cd $GRAFT_REPO_ROOT
for shape in "--M 128 --C 128 --H 64" "--M 512 --C 512 --H 16" "--M 64 --C 64 --H 128" "--M 256 --C 256 --H 32"; do
python tools/bench_op.py wgrad3 $shape --pro 3 --iters 100
python tools/bench_op.py wgrad3 $shape --pro 0 --iters 100
done
for shape in "--M 2048 --C 2048 --H 8" "--M 512 --C 3072 --H 8" "--M 3072 --C 768 --H 1 --N 200 --B 64" "--M 768 --C 3072 --H 1 --N 52 --B 64" "--M 768 --C 768 --H 1 --N 52 --B 64" "--M 128 --C 256 --H 64" "--M 240 --C 40 --H 64"; do
python tools/bench_op.py wgrad1 $shape --iters 50
done
export S2K_LIB=$GRAFT_REPO_ROOT/sentinel2-landcover-classification_amd/libs2k_tuning.so
S2K_WG_EXP=0 python tools/bench_op.py wgrad3 --M 128 --C 128 --H 64 --pro 3 --iters 100
S2K_WG_EXP=0 python tools/bench_op.py wgrad1 --M 2048 --C 2048 --H 8 --iters 100
S2K_WG_EXP=0 python tools/bench_op.py wgrad1 --M 3072 --C 768 --H 1 --N 200 --B 64 --iters 100
