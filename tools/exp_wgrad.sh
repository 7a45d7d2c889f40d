set -e
cd $GRAFT_REPO_ROOT
python tools/bench_op.py wgrad3 --M 32 --C 32 --H 256 --pro 3 --iters 50
python tools/bench_op.py wgrad3 --M 32 --C 32 --H 256 --pro 0 --iters 50
python tools/bench_op.py wgrad3 --M 32 --C 13 --H 256 --pro 0 --iters 50
python tools/bench_op.py wgrad3 --M 64 --C 24 --H 128 --pro 0 --iters 50
