# A-B of the decoder weight-gradient deferral on one box, alternating runs (noise between runs of one box is ~0.3 %):
#   tools/exp_defer_ab.sh            (S2K_DEFER_WGRAD / S2K_DEFER_MIN_GFLOP are planner switches: S2K_TUNING=1)
b="python bench.py --steps 40 --warmup 5 --no-prithvi --no-cpu-baseline --no-bf16 --no-profile"
run() { echo -n "$* -> "; e=$1; shift; env S2K_TUNING=1 $e $b "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
run A=$i
run S2K_DEFER_MIN_GFLOP=8
run S2K_DEFER_MIN_GFLOP=30
run S2K_DEFER_WGRAD=0
done
for i in 1 2 3; do
run A=$i --precision bf16-mixed
run S2K_DEFER_MIN_GFLOP=8 --precision bf16-mixed
run S2K_DEFER_MIN_GFLOP=30 --precision bf16-mixed
run S2K_DEFER_WGRAD=0 --precision bf16-mixed
done
