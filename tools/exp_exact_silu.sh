#!/bin/bash
# What the fast SiLU (v_exp_f32 + v_rcp_f32, csrc/common.h) costs in parity terms and buys in time: the training-size parity test of
# BASELINE.json configs[1]'s network (logits / class mask / gradient table against the reference fixture and the fp32 oracle) and the
# headline step, once with the shipped library and once with libs2k_exact.so (expf + true division; __graft_entry__.build_tuning(exact_silu=True))
E=$PWD/sentinel2-landcover-classification_amd/libs2k_exact.so
for lib in "" "$E"; do
  echo "==== ${lib:-shipped libs2k.so}"
  env ${lib:+S2K_LIB=$lib} timeout -k 10 500 python -m pytest tests/test_training_size_gpu.py -q -s -k "train_bs8_matches or evalgrad" 2>&1 | grep -E "rel err|class mask|gradient error|passed|failed"
  env ${lib:+S2K_LIB=$lib} timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('bench: %.1f tiles/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
