# A-B of WGRAD_FINALIZE on the side stream (planner switch S2K_FINALIZE_SIDE), alternating runs on one box
b="python bench.py --steps 40 --warmup 5 --no-prithvi --no-cpu-baseline --no-bf16 --no-profile"
run() { echo -n "$* -> "; e=$1; shift; env S2K_TUNING=1 $e $b "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss'])"; }
for i in 1 2 3; do
run A=$i
run S2K_FINALIZE_SIDE=0
done
for i in 1 2 3; do
run A=$i --precision bf16-mixed
run S2K_FINALIZE_SIDE=0 --precision bf16-mixed
done
for w in mae seg; do for v in 1 0; do echo -n "$w FINALIZE_SIDE=$v -> "; S2K_TUNING=1 S2K_FINALIZE_SIDE=$v python tools/bench_prithvi.py $w --steps 8 --warmup 3 2>/dev/null | tail -n 1 | cut -c1-160; done; done
