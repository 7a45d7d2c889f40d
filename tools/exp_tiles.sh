for shape in "768 3072 50" "3072 768 50" "2304 768 50" "768 768 50" "512 2048 197" "2048 512 197" "304 1824 64" "512 3072 64" "176 1056 256"; do
  set -- $shape
  for cfg in 1 2; do for sp in 1 2 3 4 6 8; do
    r=$(S2K_PIX_FORCE=$cfg S2K_SPLITS=$sp python tools/bench_op.py conv1 --B 64 --M $1 --C $2 --N $3 --nostats --scratch --iters 10 2>&1 | grep conv1 | awk '{print $(NF-3), $(NF-1)}')
    echo "M=$1 C=$2 N=$3 cfg=$cfg splits=$sp : $r"
  done; done
done
