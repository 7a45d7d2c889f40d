"""Round-3 fixtures from the *imported reference* (build container only — /root/reference never travels): the BASELINE.json
configurations at TRAINING size, gradients included.

    python tests/golden/make_golden_r3.py [unet|evalgrad|mae|seg|all]

unet      unet_b5_256x13_train_bs8.npz: configs[1]'s network and tile shape (efficientnet-unet-b5, 13 bands, 256x256) in TRAIN
          mode at batch 8 with injected drop-connect noise: logits, focal / CE loss, gradient subsamples, BatchNorm running
          statistics (the reference's own sub-modules wired per SURVEY §8 a7-G, make_golden.build_ref_unet).
evalgrad  unet_b5_256x13_evalgrad_bs4.npz: the same network differentiated in EVAL mode (BatchNorm on running statistics): the
          well-conditioned gradient fixture (bar 1e-3 on every weight tensor) at the benchmark's tile shape.
mae       prithvi_mae_full_bs2_grads.npz: Prithvi-100M MAE (configs[4]'s model), bs 2, mask 0.75, WITH gradients.
seg       prithvi_seg_full_train_unfrozen_bs1.npz: PrithviSegmentationNet on Prithvi-100M (configs[3]'s model), bs 1, train mode,
          unfrozen backbone, CrossEntropyLoss(ignore_index=0), WITH gradients and the head's BatchNorm buffers.
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

import ref_harness  # noqa: E402
import make_golden  # noqa: E402
import make_golden_prithvi as mgp  # noqa: E402
import make_golden_r2  # noqa: E402


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(8)
    ns = ref_harness.load()
    if what in ("unet", "all"):
        make_golden.unet_case(ns, "b5_256x13_train_bs8", "b5", 13, 256, 8, 4, False, True, 9)
    if what in ("evalgrad", "all"):
        make_golden_r2.evalgrad_case(ns, "b5_256x13_evalgrad_bs4", "b5", 13, 256, 4, 4, False, 33)
    if what in ("mae", "all"):
        mgp.mae_case(ns, "full_bs2_grads", mgp.FULL, 2, 0.75, 15, True)
    if what in ("seg", "all"):
        mgp.seg_case(ns, "full_train_unfrozen_bs1", mgp.FULL, 1, 4, 256, False, True, 25, True)
