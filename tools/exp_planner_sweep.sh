b="python bench.py --steps 30 --warmup 5 --no-prithvi --no-cpu-baseline --no-bf16 --no-profile"
run() { echo -n "$* -> "; env S2K_TUNING=1 "$@" $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=1
run S2K_DEFER_MIN_GFLOP=1
run S2K_DEFER_MIN_GFLOP=2
run S2K_DEFER_MIN_GFLOP=8
run S2K_DEFER_MIN_GFLOP=16
run S2K_DEFER_WGRAD=0
run S2K_SIDE_MAX_GFLOP=30
run S2K_SIDE_MAX_GFLOP=15
run S2K_SE_COMBINE_IN_APPLY=0
run S2K_FOLD_BN=0
run A=2
