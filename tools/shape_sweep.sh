#!/bin/bash
# The headline step on other EfficientNet-UNet shapes (run on the GPU box): tools/shape_sweep.sh <tag>
# Same bench.py, same timed region (forward + focal loss + backward + Adam), f32 and bf16-mixed; one line per shape in
# gpurun_out/<tag>_shape_sweep.md.  These are robustness / planner-coverage numbers, not BASELINE.json configurations.
tag=$1
out=gpurun_out/${tag}_shape_sweep.md
{
echo "# $tag: bench.py on other shapes (python bench.py --version V --bands C --size H --batch B --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi)"
echo
echo "| variant | bands | size | batch | f32 tiles/s | f32 ms/step | bf16-mixed tiles/s | bf16-mixed ms/step | loss f32 | loss bf16-mixed |"
echo "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"
} > $out
for cfg in "b5 13 256 32" "b7 13 256 8" "b5 6 224 16" "b0 4 512 8" "b3 13 128 64" "b0 13 256 64" "b2 10 192 24"; do
  set -- $cfg
  python3 bench.py --version $1 --bands $2 --size $3 --batch $4 --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi 2> gpurun_out/${tag}_sweep.err | tail -n 1 > gpurun_out/${tag}_sweep_line.json || { echo "| $1 | $2 | $3 | $4 | FAILED |" >> $out; continue; }
  python3 - "$@" gpurun_out/${tag}_sweep_line.json >> $out <<'PY'
import json, sys
d = json.load(open(sys.argv[5]))
b = d.get("bf16_mixed", {})
print("| %s | %s | %s | %s | %.1f | %.2f | %.1f | %.2f | %.6f | %s |" % (*sys.argv[1:5], d["value"], d["ms_per_step"], b.get("value", float("nan")), b.get("ms_per_step", float("nan")), d["loss"], b.get("parity", {}).get("loss_bf16_mixed", b.get("loss", "-"))))
PY
done
